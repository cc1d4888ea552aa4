"""Device-resident vector environments over libtma_hip.so.

`HipVecEnv` keeps the Stable-Baselines3 `VecEnv` surface the reference builds with `make_vector_env`
(/root/reference/backend/mlagents/training.py:71-89 -> DummyVecEnv(Monitor(env))): `num_envs`, spaces, `reset()`,
`step()`, `step_async/step_wait`, `seed()`, `close()`, `get_attr/set_attr/env_method/env_is_wrapped`, and per-env
`infos` carrying `terminal_observation`, `TimeLimit.truncated` and the Monitor `episode` dict.
`HipVectorEnv` exposes the same engine through the Gymnasium `VectorEnv` signatures
(`reset(seed=...) -> (obs, infos)`, `step -> (obs, rewards, terminations, truncations, infos)`).

All state lives in HBM; `reset_device/step_device` return torch tensors (views owned by the env, valid until the
next step) and are what the PPO engine uses.  The NumPy-returning methods copy for drop-in callers.
Episode k of global env i is seeded with `seed + i + k * 2**20` (k = 0 is the reference's `seed + rank`).
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Any

import numpy as np
import torch

from . import _lib
from .spaces import task_spaces

TASK_ALIASES: dict[str, str] = {}  # (round 3: "ant" is a task of its own -- the reference's Ant-v5 shapes; "crawler" is BASELINE's 172 / 20 shape)


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError(
            "three-mlagents_amd needs an AMD GPU (gfx950): the HIP kernels are the only compute path and no CPU "
            "fallback exists.  torch.cuda.is_available() is False."
        )
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise ValueError(f"device must be a cuda/hip device, got {dev}")
    return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())


class HipEnvEngine:
    """Thin owner of one `tma_env` handle plus its output tensors."""

    def __init__(self, task: str, num_envs: int, *, seed: int = 1, device=None, env_offset: int = 0, ring_depth: int | None = None):
        task = TASK_ALIASES.get(task, task)
        self.task_name = task
        if ring_depth is None:
            # A refill launch is bound by the serial MT19937 seeding chain of its few thousand lanes, not by their number: a deeper
            # ring means proportionally fewer refill (and rollout-chunk) launches at the same cost each.  Memory = N * depth records.
            # (a handful of envs -- the reference's own 1 .. 8: every chunk boundary is a few launches for almost no work, so the window is wider)
            # Measured (round 3, rollout of 1024 steps): GridWorld 4096 envs 3.86 / 3.71 / 3.56 / 3.53 ms at depth 128 / 256 / 512 / 1024,
            # Push 2048 x 2048 12.96 / 12.33 / 12.10 / 11.90 ms: every refill is five small launches between two rollout chunks.
            ring_depth = 512 if num_envs <= 4096 else (256 if num_envs <= 16384 else (64 if num_envs <= 131072 else 32))
            if os.environ.get("TMA_RING_DEPTH"):  # measurement switch
                ring_depth = int(os.environ["TMA_RING_DEPTH"])
        self.task = _lib.task_id(task)  # KeyError for unknown tasks
        L = _lib.lib()
        self.num_envs = int(num_envs)
        self.obs_dim = L.tma_task_obs_dim(self.task)
        self.num_actions = L.tma_task_num_actions(self.task)
        self.act_dim = L.tma_task_act_dim(self.task)
        self.state_dim = L.tma_task_state_dim(self.task)
        self.max_episode_steps = L.tma_task_max_episode_steps(self.task)
        self.observation_space, self.action_space = task_spaces(task)
        self.device = _require_gpu(device)
        self.ring_depth = int(ring_depth)
        self._h = C.c_void_p()
        _lib.check(L.tma_env_create(self.task, self.num_envs, self.device.index, int(seed) & 0xFFFFFFFF, int(env_offset) & 0xFFFFFFFF,
                                    self.ring_depth, C.byref(self._h)))
        self.seed_base = int(seed) & 0xFFFFFFFF
        self.env_offset = int(env_offset)
        self._bufs: dict[int, dict[str, torch.Tensor]] = {}

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().tma_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _out(self, n_steps: int):
        b = self._bufs.get(n_steps)
        if b is None:
            N, D, dev = self.num_envs, self.obs_dim, self.device
            b = dict(
                obs=torch.empty((n_steps, N, D), dtype=torch.float32, device=dev),
                rew=torch.empty((n_steps, N), dtype=torch.float32, device=dev),
                term=torch.empty((n_steps, N), dtype=torch.uint8, device=dev),
                trunc=torch.empty((n_steps, N), dtype=torch.uint8, device=dev),
                term_obs=torch.zeros((n_steps, N, D), dtype=torch.float32, device=dev),
                ep_ret=torch.empty((n_steps, N), dtype=torch.float64, device=dev),
                ep_len=torch.empty((n_steps, N), dtype=torch.int32, device=dev),
            )
            self._bufs[n_steps] = b
        return b

    # -- engine calls ---------------------------------------------------------------------
    def set_option(self, key: str, value: int) -> None:
        _lib.check(_lib.lib().tma_env_set_option(self._h, key.encode(), int(value)))

    def seed(self, seed: int):
        self.seed_base = int(seed) & 0xFFFFFFFF
        _lib.check(_lib.lib().tma_env_seed(self._h, self.seed_base))

    def reset(self, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.num_envs, self.obs_dim), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().tma_env_reset(self._h, _lib.ptr(out), self._stream()))
        return out

    def steps_until_refill(self) -> int:
        out = C.c_int(0)
        _lib.check(_lib.lib().tma_env_steps_until_refill(self._h, C.byref(out)))
        return out.value

    def step(self, actions: torch.Tensor | None, *, n_steps: int = 1, tape_seed: int = 0, tape_t0: int = 0, outputs: dict | None = None,
             want_terminal_obs: bool = True, want_episode: bool = True) -> dict[str, torch.Tensor]:
        """One launch of `n_steps` vector steps.  `actions` None -> device-generated tape."""
        b = outputs if outputs is not None else self._out(n_steps)
        dtype = 0
        aptr = None
        if actions is not None:
            if actions.device != self.device:
                actions = actions.to(self.device, non_blocking=True)
            if self.num_actions > 0:
                if actions.dtype == torch.int32:
                    dtype = _lib.ACT_I32
                elif actions.dtype == torch.int64:
                    dtype = _lib.ACT_I64
                else:
                    raise ValueError(f"task '{self.task_name}' has Discrete({self.num_actions}) actions: pass int32 or int64, got {actions.dtype}")
                need = n_steps * self.num_envs
            else:
                if actions.dtype != torch.float32:
                    actions = actions.float()
                dtype = _lib.ACT_F32
                need = n_steps * self.num_envs * self.act_dim
            if actions.numel() != need:
                raise ValueError(f"actions has {actions.numel()} elements, expected {need}")
            actions = actions.contiguous()
            aptr = _lib.ptr(actions)
        _lib.check(
            _lib.lib().tma_env_step(
                self._h, aptr, dtype, int(tape_seed) & 0xFFFFFFFF, int(tape_t0) & 0xFFFFFFFF, int(n_steps), _lib.ptr(b["obs"]),
                _lib.ptr(b.get("rew")), _lib.ptr(b.get("term")), _lib.ptr(b.get("trunc")),
                _lib.ptr(b.get("term_obs")) if want_terminal_obs else None,
                _lib.ptr(b.get("ep_ret")) if want_episode else None, _lib.ptr(b.get("ep_len")) if want_episode else None, self._stream(),
            )
        )
        return b

    def get_state(self) -> torch.Tensor:
        out = torch.empty((self.num_envs, self.state_dim), dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().tma_env_get_state(self._h, _lib.ptr(out), self._stream()))
        return out

    def set_state(self, state) -> None:
        st = torch.as_tensor(np.asarray(state, np.float64) if not torch.is_tensor(state) else state, dtype=torch.float64).to(self.device).contiguous()
        if st.numel() != self.num_envs * self.state_dim:
            raise ValueError(f"state has {st.numel()} elements, expected {self.num_envs * self.state_dim}")
        _lib.check(_lib.lib().tma_env_set_state(self._h, _lib.ptr(st), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def episode_index(self) -> torch.Tensor:
        out = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().tma_env_episode_index(self._h, _lib.ptr(out), self._stream()))
        return out

    def episode_log(self, capacity: int) -> None:
        """Turn the per-episode Monitor log on (capacity records kept between pops) or off (0)."""
        _lib.check(_lib.lib().tma_env_episode_log(self._h, int(capacity)))
        self._log_cap = int(capacity)

    def pop_episode_log(self) -> tuple[np.ndarray, np.ndarray, np.ndarray, int]:
        """(returns f64[n], lengths i32[n], env index i32[n], episodes seen) since the last pop, in the order the kernels logged them
        (per env: the order the episodes finished)."""
        cap = getattr(self, "_log_cap", 0)
        if cap <= 0:
            raise RuntimeError("episode log is off: call episode_log(capacity) first")
        host = getattr(self, "_log_host", None)
        if host is None or len(host[0]) != cap:  # (staging buffers are kept: a fresh 16 MB of pages per pop costs a millisecond)
            host = self._log_host = (np.empty(cap, np.float64), np.empty(cap, np.int32), np.empty(cap, np.int32))
        r, l, e = host
        n, seen = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().tma_env_pop_episode_log(self._h, r.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                                      cap, C.byref(n), C.byref(seen), self._stream()))
        return r[:n.value].copy(), l[:n.value].copy(), e[:n.value].copy(), int(seen.value)

    def detach_episode_log(self) -> None:
        """Host-side swap of the Monitor aggregate / episode-log buffers (include/tma.h tma_env_detach_episode_log): what the kernels launched
        so far wrote can then be read with pop_detached_episode_log on a side stream while later launches fill the other set."""
        _lib.check(_lib.lib().tma_env_detach_episode_log(self._h))

    def pop_detached_episode_log(self, stream_ptr=None):
        """((sum of returns, sum of lengths, episodes), returns f64[n], lengths i32[n], env index i32[n], episodes seen) of the detached set;
        synchronises only the given stream (default: the current one), which must be ordered behind the kernels that wrote the set."""
        cap = max(int(getattr(self, "_log_cap", 0)), 0)
        host = getattr(self, "_log_host", None)
        if cap and (host is None or len(host[0]) != cap):
            host = self._log_host = (np.empty(cap, np.float64), np.empty(cap, np.int32), np.empty(cap, np.int32))
        r, l, e = host if cap else (np.empty(0, np.float64), np.empty(0, np.int32), np.empty(0, np.int32))
        n, seen, st = C.c_int64(0), C.c_int64(0), (C.c_double * 3)()
        _lib.check(_lib.lib().tma_env_pop_detached_episode_log(self._h, r.ctypes.data_as(C.c_void_p) if cap else None, l.ctypes.data_as(C.c_void_p) if cap else None,
                                                               e.ctypes.data_as(C.c_void_p) if cap else None, cap, C.byref(n), C.byref(seen), st,
                                                               stream_ptr if stream_ptr is not None else self._stream()))
        k = n.value
        return (float(st[0]), float(st[1]), int(st[2])), r[:k].copy(), l[:k].copy(), e[:k].copy(), int(seen.value)

    def clear_episode_log(self) -> None:
        """Empty the episode log and the Monitor aggregate, ordered on the current stream; no read-back, no synchronisation."""
        _lib.check(_lib.lib().tma_env_clear_episode_log(self._h, self._stream()))

    def pop_episode_stats(self) -> tuple[float, float, int]:
        out = (C.c_double * 3)()
        _lib.check(_lib.lib().tma_env_pop_episode_stats(self._h, out, self._stream()))
        return float(out[0]), float(out[1]), int(out[2])


class HipVecEnv:
    """SB3 `VecEnv`-shaped view of a HipEnvEngine (drop-in for DummyVecEnv(Monitor(...)))."""

    def __init__(self, task: str, num_envs: int, *, seed: int = 1, device=None, env_offset: int = 0, ring_depth: int | None = None):
        self.engine = HipEnvEngine(task, num_envs, seed=seed, device=device, env_offset=env_offset, ring_depth=ring_depth)
        self.task_id = self.engine.task_name
        self.num_envs = self.engine.num_envs
        self.observation_space = self.engine.observation_space
        self.action_space = self.engine.action_space
        self.device = self.engine.device
        self.render_mode = None
        self._actions = None
        self._t_start = time.time()
        self.reset_infos: list[dict[str, Any]] = [{} for _ in range(self.num_envs)]

    # -- device API (used by the PPO engine) ---------------------------------------------
    def reset_device(self) -> torch.Tensor:
        return self.engine.reset()

    def step_device(self, actions: torch.Tensor, **kw) -> dict[str, torch.Tensor]:
        return self.engine.step(actions, **kw)

    # -- SB3 VecEnv API -------------------------------------------------------------------
    def seed(self, seed: int | None = None):
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1))
        self.engine.seed(seed)
        return [seed + i for i in range(self.num_envs)]

    def reset(self) -> np.ndarray:
        obs = self.engine.reset().cpu().numpy()
        self.reset_infos = [{} for _ in range(self.num_envs)]
        return obs

    def step_async(self, actions) -> None:
        self._actions = actions

    def _actions_to_device(self, actions) -> torch.Tensor:
        if torch.is_tensor(actions):
            return actions
        a = np.asarray(actions)
        if self.engine.num_actions > 0:
            a = a.astype(np.int64, copy=False).reshape(self.num_envs)
        else:
            a = a.astype(np.float32, copy=False).reshape(self.num_envs, self.engine.act_dim)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def step_wait(self):
        out = self.engine.step(self._actions_to_device(self._actions))
        obs = out["obs"][0].cpu().numpy()
        rew = out["rew"][0].cpu().numpy()
        term = out["term"][0].cpu().numpy().astype(bool)
        trunc = out["trunc"][0].cpu().numpy().astype(bool)
        dones = term | trunc
        infos: list[dict[str, Any]] = [{"TimeLimit.truncated": False} for _ in range(self.num_envs)]
        idx = np.nonzero(dones)[0]
        if idx.size:
            tobs = out["term_obs"][0].cpu().numpy()
            ep_ret = out["ep_ret"][0].cpu().numpy()
            ep_len = out["ep_len"][0].cpu().numpy()
            now = round(time.time() - self._t_start, 6)
            for i in idx:
                infos[i] = {
                    "steps": int(ep_len[i]),
                    "TimeLimit.truncated": bool(trunc[i] and not term[i]),
                    "terminal_observation": tobs[i].copy(),
                    "episode": {"r": round(float(ep_ret[i]), 6), "l": int(ep_len[i]), "t": now},  # Monitor
                }
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self) -> None:
        self.engine.close()

    def get_attr(self, attr_name: str, indices=None):
        n = self.num_envs if indices is None else len(list(np.atleast_1d(indices)))
        return [getattr(self.engine, attr_name) for _ in range(n)]

    def set_attr(self, attr_name: str, value, indices=None) -> None:
        raise AttributeError(f"HipVecEnv attributes are engine-owned; cannot set '{attr_name}'")

    def env_method(self, method_name: str, *args, indices=None, **kwargs):
        raise AttributeError(f"HipVecEnv has no per-env python objects; cannot call '{method_name}'")

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(list(np.atleast_1d(indices)))
        return [getattr(wrapper_class, "__name__", "") == "Monitor"] * n  # Monitor bookkeeping is built into the kernel

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None

    @property
    def unwrapped(self):
        return self

    def __len__(self):
        return self.num_envs


class HipVectorEnv:
    """Gymnasium `VectorEnv`-shaped view (reset(seed=) -> (obs, infos); step -> 5-tuple)."""

    def __init__(self, task: str, num_envs: int, *, seed: int = 1, device=None, env_offset: int = 0, ring_depth: int | None = None):
        self._vec = HipVecEnv(task, num_envs, seed=seed, device=device, env_offset=env_offset, ring_depth=ring_depth)
        self.engine = self._vec.engine
        self.num_envs = self._vec.num_envs
        self.single_observation_space = self._vec.observation_space
        self.single_action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self.action_space = self._vec.action_space
        self.closed = False

    def reset(self, *, seed: int | None = None, options: dict | None = None):
        if seed is not None:
            self._vec.seed(int(seed))
        obs = self._vec.reset()
        if options and "position" in options and self.engine.task_name == "basic":
            st = np.tile(np.array([float(np.clip(int(options["position"]), 0, 20)), 0.0]), (self.num_envs, 1))
            self.engine.set_state(st)
            obs = np.zeros_like(obs)
            obs[:, int(st[0, 0])] = 1.0
        return obs, {}

    def step(self, actions):
        out = self.engine.step(self._vec._actions_to_device(actions))
        obs = out["obs"][0].cpu().numpy()
        rew = out["rew"][0].cpu().numpy()
        term = out["term"][0].cpu().numpy().astype(bool)
        trunc = out["trunc"][0].cpu().numpy().astype(bool)
        infos: dict[str, Any] = {}
        done = term | trunc
        if done.any():  # gymnasium >= 1.0 vector autoreset info layout (final_obs + mask)
            infos["final_obs"] = out["term_obs"][0].cpu().numpy()
            infos["_final_obs"] = done
            infos["episode"] = {"r": out["ep_ret"][0].cpu().numpy(), "l": out["ep_len"][0].cpu().numpy(), "_r": done, "_l": done}
            infos["_episode"] = done
        return obs, rew, term, trunc, infos

    def close(self, **kwargs):
        self.closed = True
        self._vec.close()
