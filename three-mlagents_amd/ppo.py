"""PPO engine with the Stable-Baselines3 `PPO` surface the reference drives.

Drop-in for `ALGORITHMS["ppo"](policy, env, seed=seed, **kwargs)` + `.learn()/.predict()/.save()/.load()` as used at
/root/reference/backend/mlagents/training.py:150,166-170,175,269,278.  Every arithmetic step of the hot path -- policy
forward and sampling, env step, timeout bootstrap, GAE, minibatch forward/backward of the clipped-surrogate loss,
global-norm clipping and Adam -- is a HIP kernel reached through the C ABI (include/tma.h); this file is host plumbing
(buffers, loop order, callbacks, artefacts).  Semantics follow SB3 2.9.0; deviations, all
forced by running thousands of envs on a GPU, are listed in DESIGN.md §"Deviations":
  * action sampling and the minibatch permutation use counter-based device RNGs instead of torch/numpy global RNGs;
  * per-(env, episode) reset seeds.
"""
from __future__ import annotations

import ctypes as C
import io
import json
import math
import os
import platform
import time
import zipfile
from typing import Any

import numpy as np
import torch

from . import _lib
from .vec_env import HipVecEnv

SB3_KEYS = [
    "mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias",
    "mlp_extractor.policy_net.2.weight", "mlp_extractor.policy_net.2.bias",
    "action_net.weight", "action_net.bias",
    "mlp_extractor.value_net.0.weight", "mlp_extractor.value_net.0.bias",
    "mlp_extractor.value_net.2.weight", "mlp_extractor.value_net.2.bias",
    "value_net.weight", "value_net.bias",
]


def _hidden_from_net_arch(net_arch) -> int:
    """SB3 net_arch -> H.  Supported: dict(pi=[H,H], vf=[H,H]) or [H,H] (SB3 default is [64,64]; the reference passes
    dict(pi=[256,256], vf=[256,256]), backend/mlagents/training.py:363-365)."""
    if net_arch is None:
        return 64
    if isinstance(net_arch, dict):
        pi, vf = list(net_arch.get("pi", [])), list(net_arch.get("vf", []))
    else:
        pi = vf = list(net_arch)
    if len(pi) != 2 or len(vf) != 2 or len(set(pi + vf)) != 1:
        raise ValueError(f"three-mlagents_amd supports net_arch with two equal hidden layers for pi and vf, got {net_arch}")
    H = int(pi[0])
    if H % 64 or not 64 <= H <= 1024:
        raise ValueError(f"hidden width must be a multiple of 64 in [64, 1024], got {H}")
    return H


def _dist_backend() -> str | None:
    import torch.distributed as td

    return td.get_backend() if td.is_available() and td.is_initialized() else None


class HipActorCriticPolicy:
    """Parameters of SB3's ActorCriticPolicy(MlpPolicy) in one flat HBM buffer + the forward kernels."""

    def __init__(self, obs_dim: int, act_dim: int, continuous: bool, hidden: int, device, seed: int = 0, mfma_dtype: str = "f32"):
        if mfma_dtype not in ("f32", "bf16", "bf16x3"):  # bf16x3 (round 5): the f32 update on the bf16 MFMA, every operand as three bf16 terms (csrc/tma_split3.h)
            raise ValueError(f"mfma_dtype must be 'f32', 'bf16' or 'bf16x3', got {mfma_dtype!r}")
        self.mfma_dtype = mfma_dtype
        self.obs_dim, self.act_dim, self.continuous, self.hidden = int(obs_dim), int(act_dim), bool(continuous), int(hidden)
        self.device = torch.device(device)
        # dims.device: every tma_policy_* / tma_ppo_* call makes it the calling thread's HIP device (learn() may run in a worker thread)
        self.dims = _lib.PolicyDims(int(obs_dim), int(hidden), int(act_dim), 1 if continuous else 0, {"f32": 0, "bf16": 1, "bf16x3": 2}[mfma_dtype],
                                    self.device.index if self.device.index is not None else -1)
        nt, ntot = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().tma_policy_param_count(C.byref(self.dims), C.byref(nt), C.byref(ntot)))
        self.n_trainable, self.n_total = nt.value, ntot.value
        offs = (C.c_int32 * 13)()
        _lib.check(_lib.lib().tma_policy_param_offsets(C.byref(self.dims), offs))
        self.offsets = list(offs)
        self.params = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        self.load_state_dict(self._orthogonal_init(seed))

    # -- init / (de)serialisation in SB3's state_dict naming -------------------------------
    def _orthogonal_init(self, seed: int) -> dict[str, torch.Tensor]:
        gen = torch.Generator().manual_seed(int(seed))
        D, H, A = self.obs_dim, self.hidden, self.act_dim

        def ortho(o, i, gain):
            w = torch.empty(o, i)
            torch.nn.init.orthogonal_(w, gain=gain, generator=gen)
            return w

        g2 = math.sqrt(2.0)
        sd = {
            SB3_KEYS[0]: ortho(H, D, g2), SB3_KEYS[1]: torch.zeros(H), SB3_KEYS[2]: ortho(H, H, g2), SB3_KEYS[3]: torch.zeros(H),
            SB3_KEYS[4]: ortho(A, H, 0.01), SB3_KEYS[5]: torch.zeros(A),
            SB3_KEYS[6]: ortho(H, D, g2), SB3_KEYS[7]: torch.zeros(H), SB3_KEYS[8]: ortho(H, H, g2), SB3_KEYS[9]: torch.zeros(H),
            SB3_KEYS[10]: ortho(1, H, 1.0), SB3_KEYS[11]: torch.zeros(1),
        }
        if self.continuous:
            sd["log_std"] = torch.zeros(A)
        return sd

    def _segments(self):
        D, H, A = self.obs_dim, self.hidden, self.act_dim
        shapes = [(H, D), (H,), (H, H), (H,), (A, H), (A,), (H, D), (H,), (H, H), (H,), (1, H), (1,)]
        return list(zip(SB3_KEYS, self.offsets[:12], shapes))

    def load_state_dict(self, sd: dict[str, torch.Tensor]) -> None:
        self.params[: self.n_trainable].copy_(self.flat_from_named(sd).to(self.device))
        _lib.check(_lib.lib().tma_policy_sync(_lib.ptr(self.params), C.byref(self.dims), _lib.stream_ptr(self.device)))

    def state_dict(self) -> dict[str, torch.Tensor]:
        return self.named_from_flat(self.params[: self.n_trainable])

    def flat_from_named(self, sd: dict[str, torch.Tensor]) -> torch.Tensor:
        """SB3-named tensors ([out][in] matrices) -> one vector in the engine's trainable layout (CPU)."""
        flat = torch.zeros(self.n_trainable, dtype=torch.float32)
        for key, off, shape in self._segments():
            w = sd[key].detach().to(torch.float32).cpu().reshape(shape)
            w = w.t().contiguous() if len(shape) == 2 else w
            flat[off:off + w.numel()] = w.reshape(-1)
        if self.continuous:
            flat[self.offsets[12]:self.offsets[12] + self.act_dim] = sd["log_std"].detach().float().cpu()
        return flat

    def named_from_flat(self, vec: torch.Tensor) -> dict[str, torch.Tensor]:
        """A vector in the engine's trainable layout (parameters, or an optimizer moment) -> SB3-named tensors."""
        flat = vec.detach().cpu()
        sd = {}
        for key, off, shape in self._segments():
            n = int(np.prod(shape))
            w = flat[off:off + n]
            sd[key] = (w.reshape(shape[1], shape[0]).t().contiguous() if len(shape) == 2 else w.clone())
        if self.continuous:
            sd["log_std"] = flat[self.offsets[12]:self.offsets[12] + self.act_dim].clone()
        return sd

    # -- kernels ---------------------------------------------------------------------------
    def _rows(self, obs: torch.Tensor) -> torch.Tensor:
        """[n, obs_dim] float32 on the policy's device, or ValueError (the kernels read n * obs_dim floats whatever the tensor holds;
        SB3 raises the same for a mis-shaped observation)."""
        if obs.dim() != 2 or obs.shape[1] != self.obs_dim or obs.shape[0] < 1:
            raise ValueError(f"observations must have shape [n, {self.obs_dim}], got {tuple(obs.shape)}")
        return obs.to(self.device, torch.float32).contiguous()

    def act(self, obs: torch.Tensor, *, rng_seed: int = 0, rng_step: int = 0, env_offset: int = 0, deterministic: bool = False):
        obs = self._rows(obs)
        n = obs.shape[0]
        if self.continuous:
            actions = torch.empty((n, self.act_dim), dtype=torch.float32, device=self.device)
        else:
            actions = torch.empty((n,), dtype=torch.int32, device=self.device)
        values = torch.empty((n,), dtype=torch.float32, device=self.device)
        logp = torch.empty((n,), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().tma_policy_act(_lib.ptr(self.params), C.byref(self.dims), _lib.ptr(obs), n, int(rng_seed) & 0xFFFFFFFF,
                                             int(rng_step) & 0xFFFFFFFF, int(env_offset) & 0xFFFFFFFF, 1 if deterministic else 0,
                                             _lib.ptr(actions), _lib.ptr(values), _lib.ptr(logp), _lib.stream_ptr(self.device)))
        return actions, values, logp

    def predict_values(self, obs: torch.Tensor) -> torch.Tensor:
        obs = self._rows(obs)
        values = torch.empty((obs.shape[0],), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().tma_policy_values(_lib.ptr(self.params), C.byref(self.dims), _lib.ptr(obs), obs.shape[0], _lib.ptr(values),
                                                _lib.stream_ptr(self.device)))
        return values


class PPO:
    """SB3-shaped PPO whose compute is libtma_hip.so.  Constructor/learn/predict/save/load keep SB3's names and defaults."""

    def __init__(self, policy: str = "MlpPolicy", env: HipVecEnv | None = None, learning_rate: float = 3e-4, n_steps: int = 2048,
                 batch_size: int = 64, n_epochs: int = 10, gamma: float = 0.99, gae_lambda: float = 0.95, clip_range: float = 0.2,
                 clip_range_vf=None, normalize_advantage: bool = True, ent_coef: float = 0.0, vf_coef: float = 0.5,
                 max_grad_norm: float = 0.5, policy_kwargs: dict | None = None, tensorboard_log: str | None = None, verbose: int = 0,
                 seed: int | None = None, device="auto", _init_setup_model: bool = True, stats_window_size: int | None = None, **unused):
        if policy not in ("MlpPolicy", "MultiInputPolicy"):
            raise ValueError(f"three-mlagents_amd PPO supports MlpPolicy (vector observations), got '{policy}'")
        if clip_range_vf is not None:
            raise ValueError("clip_range_vf is not supported (the reference leaves it None)")
        if callable(learning_rate) or callable(clip_range):
            raise ValueError("schedules are not supported: pass constant learning_rate / clip_range (the reference does)")
        self.policy_class = policy
        self.env = env
        # SB3's ep_info_buffer (OnPolicyAlgorithm: deque(maxlen=stats_window_size), default 100): rollout/ep_rew_mean / ep_len_mean over the LAST
        # stats_window_size finished episodes, carried across iterations.  Here that is OPT-IN (stats_window_size=100): the window is filled from the
        # device episode log (on with a monitor_dir), whose order WITHIN a vector step is the order the kernels' atomics landed -- with thousands of
        # episodes per interval the last hundred are then a different sample from run to run, while the default, the mean over ALL episodes of the
        # interval (DESIGN.md deviation 6), is reproducible to the last bit of a sum of doubles
        import collections

        self.stats_window_size = int(stats_window_size) if stats_window_size else 0
        self._ep_info_r: collections.deque = collections.deque(maxlen=max(self.stats_window_size, 1))
        self._ep_info_l: collections.deque = collections.deque(maxlen=max(self.stats_window_size, 1))
        self.learning_rate, self.n_steps, self.batch_size, self.n_epochs = float(learning_rate), int(n_steps), int(batch_size), int(n_epochs)
        self.gamma, self.gae_lambda, self.clip_range = float(gamma), float(gae_lambda), float(clip_range)
        self.normalize_advantage, self.ent_coef, self.vf_coef = bool(normalize_advantage), float(ent_coef), float(vf_coef)
        self.max_grad_norm = float(max_grad_norm)
        self.policy_kwargs = dict(policy_kwargs or {})
        self.tensorboard_log, self.verbose = tensorboard_log, int(verbose)
        self.seed = 0 if seed is None else int(seed)
        self.num_timesteps = 0
        self._n_updates = 0
        self._adam_step = 0
        self._epoch_counter = 0
        self._rollout_counter = 0
        self.logger_values: dict[str, float] = {}
        self.ep_info: list[tuple[float, float]] = []
        self.policy: HipActorCriticPolicy | None = None
        self._last_obs_valid = False
        from . import dist as _dist

        self.world_size, self.rank = _dist.world_size(), _dist.rank()
        # data-parallel diagnostics (bench.py `dp_timing`): set to a dict and train() records HIP events around every collective of its
        # first `dp_timing_samples` minibatches / epochs -- {"grad_allreduce_us": [...], "adv_allreduce_us": [...]} after dp_timing_collect()
        self.dp_timing: dict | None = None
        self.dp_timing_samples = 64
        self._dp_events: dict[str, list] = {"grad_allreduce_us": [], "adv_allreduce_us": []}
        self._native_comm = None  # dist.NativeComm (set by _setup_model when the job runs on the RCCL backend)
        if env is not None and _init_setup_model:
            self._setup_model()

    # ------------------------------------------------------------------------------------
    def _setup_model(self) -> None:
        env = self.env
        if not isinstance(env, HipVecEnv):
            raise ValueError("env must be a three_mlagents_amd HipVecEnv (use training.make_vector_env)")
        eng = env.engine
        self.device = eng.device
        self.n_envs = env.num_envs
        self.observation_space, self.action_space = env.observation_space, env.action_space
        H = _hidden_from_net_arch(self.policy_kwargs.get("net_arch"))
        cont = eng.num_actions == 0
        A = eng.act_dim if cont else eng.num_actions
        # policy_kwargs["mfma_dtype"] = "bf16" (engine extension; BASELINE.json configs[2]): bf16 MFMA operands for 128..256-wide nets
        self.policy = HipActorCriticPolicy(eng.obs_dim, A, cont, H, self.device, seed=self.seed,
                                           mfma_dtype=self.policy_kwargs.get("mfma_dtype", "f32"))
        env.seed(self.seed)  # BaseAlgorithm.set_random_seed -> env.seed(seed): env i gets seed + i
        T, N, D, dev = self.n_steps, self.n_envs, eng.obs_dim, self.device
        f32 = torch.float32
        # terminal-observation slots: the rollout paths that run the value net in batches (per-step launches; the fused f32 256-wide chunk,
        # which carries the policy net only) bootstrap this many steps per launch (include/tma.h) -- up to one reset-ring window
        self._tobs_slots = max(1, min(max(128, eng.ring_depth), T, (256 << 20) // max(1, N * D * 4)))
        self.buf = dict(
            obs=torch.zeros((T + 1, N, D), dtype=f32, device=dev),
            actions=torch.zeros((T, N, A), dtype=f32, device=dev) if cont else torch.zeros((T, N), dtype=torch.int32, device=dev),
            rewards=torch.zeros((T, N), dtype=f32, device=dev), values=torch.zeros((T, N), dtype=f32, device=dev),
            log_probs=torch.zeros((T, N), dtype=f32, device=dev), terminated=torch.zeros((T, N), dtype=torch.uint8, device=dev),
            truncated=torch.zeros((T, N), dtype=torch.uint8, device=dev), terminal_obs=torch.zeros((self._tobs_slots, N, D), dtype=f32, device=dev),
            last_values=torch.zeros((N,), dtype=f32, device=dev), advantages=torch.zeros((T, N), dtype=f32, device=dev),
            returns=torch.zeros((T, N), dtype=f32, device=dev),
        )
        P = self.policy.n_trainable
        self.grad = torch.zeros(P, dtype=f32, device=dev)
        self.exp_avg = torch.zeros(P, dtype=f32, device=dev)
        self.exp_avg_sq = torch.zeros(P, dtype=f32, device=dev)
        self.workspace = torch.zeros(int(_lib.lib().tma_ppo_workspace_bytes(C.byref(self.policy.dims))), dtype=torch.uint8, device=dev)
        b = self.buf
        self._rb = _lib.RolloutBuffers(_lib.ptr(b["obs"]), _lib.ptr(b["actions"]), _lib.ptr(b["rewards"]), _lib.ptr(b["values"]),
                                       _lib.ptr(b["log_probs"]), _lib.ptr(b["terminated"]), _lib.ptr(b["truncated"]), _lib.ptr(b["terminal_obs"]),
                                       _lib.ptr(b["last_values"]), N, self._tobs_slots)
        # sample records for the update (include/tma.h tma_rollout.packed): filled once per rollout at the end of collect_rollouts()
        # on by default where the shape has them (TMA_NO_PACKED=1: gather from the planes, the A/B switch of DESIGN.md section 10-3)
        use_records = not os.environ.get("TMA_NO_PACKED")
        n_packed = _lib.lib().tma_ppo_packed_floats(C.byref(self.policy.dims), T, N) if use_records else 0
        self._packed = torch.empty(n_packed, dtype=torch.float32, device=dev) if n_packed > 0 else None
        self._rollout_view = _lib.Rollout(_lib.ptr(b["obs"]), _lib.ptr(b["actions"]), _lib.ptr(b["log_probs"]), _lib.ptr(b["advantages"]),
                                          _lib.ptr(b["returns"]), T, N, _lib.ptr(self._packed) if self._packed is not None else None)
        self._hp = _lib.PPOHParams(self.clip_range, self.ent_coef, self.vf_coef, 1 if self.normalize_advantage else 0)
        # data parallel: global advantage statistics come from tma_ppo_epoch_prepare's partials -- its limits are checked HERE, before any
        # rank has collected a rollout, not inside train()
        if self.world_size > 1 and self.normalize_advantage and not (T * N <= (1 << 22) and self.batch_size >= 256):
            raise ValueError("data-parallel PPO needs batch_size >= 256 per rank and n_steps * n_envs <= 2**22 per rank (global advantage statistics)")
        # The collectives of the data-parallel update.  With the RCCL backend the library owns a communicator of its own (dist.NativeComm ->
        # include/tma.h tma_comm_*) and the native epoch loop issues ncclAllReduce itself on the compute stream: no Python, no torch.distributed
        # and no stream hand-off per minibatch.  Any other backend (gloo: the CPU / one-GPU tests), TMA_NO_NATIVE_RCCL=1 or a failed self-check keep
        # the callback into torch.distributed.  TMA_NATIVE_RCCL=1 builds the communicator at world size 1 too (exercises real RCCL on one GPU).
        self._native_comm = None
        # ("nccl" IN the backend string: a group initialised without a backend name reports "cpu:gloo,cuda:nccl" and serves CUDA tensors over RCCL)
        # On top of it, the PEER EXCHANGE (tma_comm_p2p_*: the gradient sum as direct xGMI stores into the peers' inboxes, fused into the slab
        # reduction and the sum-of-squares pass on the H = 64 path -- no collective launch in the minibatch chain): set up, checked and TIMED against
        # RCCL at construction, kept when it is the faster one.  TMA_P2P=0: never; TMA_P2P=1: always, and with a non-RCCL backend (gloo: several
        # ranks on one GPU, the tests) a communicator that has ONLY the exchange.
        want_native = (os.environ.get("TMA_NATIVE_RCCL") == "1" or (self.world_size > 1 and "nccl" in (_dist_backend() or ""))
                       or (os.environ.get("TMA_P2P") in ("1", "auto") and torch.cuda.is_available()))
        if want_native and not os.environ.get("TMA_NO_NATIVE_RCCL"):
            self._native_comm = self._make_native_comm()
        if self.world_size > 1 and self._native_comm is None and self.rank == 0:
            import sys

            print(f"three-mlagents_amd: data-parallel collectives go through the torch.distributed callback (backend {_dist_backend()!r}"
                  f"{', TMA_NO_NATIVE_RCCL set' if os.environ.get('TMA_NO_NATIVE_RCCL') else ''}), not the library's own RCCL communicator",
                  file=sys.stderr, flush=True)

    def _make_native_comm(self):
        """dist.NativeComm, checked against torch.distributed on a known vector before it is trusted with gradients; None (and one stderr
        line) if RCCL cannot be bound or the check fails -- the callback path then carries the collectives.  Every rank takes part in the
        agreement (a MIN all-reduce of the local verdict over torch.distributed) whatever happened locally, so the ranks cannot end up on
        different paths or leave each other waiting."""
        import sys

        from . import dist as _dist

        comm, ok, why = None, False, ""

        def agree(mine: bool) -> bool:  # MIN over the ranks of a local verdict (torch.distributed, the group that already works)
            if self.world_size == 1:
                return mine
            import torch.distributed as tdist

            flag = torch.tensor([1.0 if mine else 0.0], device=self.device if "nccl" in str(tdist.get_backend()) else "cpu")
            tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
            return flag.item() == 1.0

        def probe_ok(c) -> bool:  # exact-integer sums: f32 over 1024 elements, f64 over 8 (the advantage-sum message)
            tri = float(self.world_size * (self.world_size + 1) // 2)
            p32 = torch.arange(1, 1025, dtype=torch.float32, device=self.device) * float(self.rank + 1)
            p64 = (torch.arange(1, 9, dtype=torch.float64, device=self.device) + 2.0 ** 40) * float(self.rank + 1)
            c.all_reduce_(p32, self._stream())
            c.all_reduce_(p64, self._stream())
            torch.cuda.current_stream(self.device).synchronize()
            return bool(torch.equal(p32, torch.arange(1, 1025, dtype=torch.float32, device=self.device) * tri)
                        and torch.equal(p64, (torch.arange(1, 9, dtype=torch.float64, device=self.device) + 2.0 ** 40) * tri))

        use_rccl = self.world_size == 1 or "nccl" in (_dist_backend() or "")  # (gloo group + TMA_P2P=1: the exchange alone)
        if use_rccl:
            # first agreement BEFORE any native collective step: a rank that cannot bind RCCL must not leave the others inside the unique-id
            # broadcast or ncclCommInitRank
            if not agree(bool(_lib.lib().tma_comm_available())):
                print("three-mlagents_amd: native RCCL communicator not used (librccl.so.1 could not be bound on every rank); collectives go through "
                      "torch.distributed", file=sys.stderr, flush=True)
                return None
        try:
            comm = _dist.NativeComm(self.device, rccl=use_rccl)
            ok = probe_ok(comm) if use_rccl else True
            why = "" if ok else "self-check of the native all-reduce against the expected sum failed"
        except Exception as exc:  # noqa: BLE001
            why = str(exc)
        if not agree(ok) and ok:
            ok, why = False, "another rank could not use its native communicator"
        # (TMA_P2P=auto: the whole start-up procedure -- contest included -- at any world size: how the one-GPU tests run the code a node runs)
        # OPT-IN (round 6, ADVICE): the exchange has run between processes sharing one GPU only -- never across xGMI -- and a bad mapping there
        # is a memory fault, not an error code; a timing contest also lets the SAME job land on RCCL (ring order) or the exchange (rank order)
        # from one start to the next, i.e. different gradient bits.  Until a multi-GPU node has passed tests/test_dist_gpu.py the default is
        # RCCL, every time; TMA_P2P=1 forces the exchange, TMA_P2P=auto runs the checked contest.  The path taken is recorded
        # (self.allreduce_path -> the saved model's data["tma"], metadata.json).
        if ok and os.environ.get("TMA_P2P") in ("1", "auto"):
            self._setup_peer_exchange(comm, agree, probe_ok, use_rccl)
        if ok and (use_rccl or comm.p2p_enabled):
            return comm
        if ok:
            why = f"no RCCL side for this backend and no peer exchange either ({comm.p2p_note})"
        if comm is not None:
            comm.close()
        print(f"three-mlagents_amd: native communicator not used ({why}); collectives go through torch.distributed", file=sys.stderr, flush=True)
        return None

    def _setup_peer_exchange(self, comm, agree, probe_ok, has_rccl: bool) -> None:
        """Peer exchange on top of `comm` (dist.NativeComm.p2p_*): inbox + handle gather + attach, an exact-sum self-check, and -- when RCCL is
        there to compare with -- a timing contest on a gradient-sized message (stand-alone launches both ways, max over ranks); the exchange
        stays on when it wins or TMA_P2P=1 forces it.  Every step is followed by an agreement over torch.distributed, so the ranks leave with the
        same answer; the outcome is one stderr line on rank 0 and comm.p2p_note."""
        import sys
        import time

        from . import dist as _dist

        P = int(self.grad.numel())
        words = max(65536, (P + 4095) // 4096 * 4096)  # slot size: the flat gradient (an inbox is 2 x world x 8 bytes per word)
        note, on = "", False

        def step(fn) -> tuple[bool, object]:
            try:
                return True, fn()
            except Exception as exc:  # noqa: BLE001
                return False, exc

        if self.world_size > 8 or P > (1 << 20):
            note = f"not applicable (world {self.world_size}, {P} gradient words; it serves <= 8 ranks and <= {1 << 20} words)"
        else:
            good, mine = step(lambda: comm.p2p_prepare(words))
            if not agree(good):
                note = f"inbox allocation / IPC export failed{'' if good else f': {mine}'}"
            else:
                if self.world_size > 1:
                    import torch.distributed as tdist

                    t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
                    t = t.to(self.device) if "nccl" in str(tdist.get_backend()) else t
                    gathered = [torch.zeros_like(t) for _ in range(self.world_size)]
                    tdist.all_gather(gathered, t)
                    handles = [g.cpu().numpy().tobytes() for g in gathered]
                else:
                    handles = [mine]
                good, exc = step(lambda: comm.p2p_attach(handles))
                if not agree(good):
                    note = f"mapping a peer's inbox failed{'' if good else f': {exc}'}"
                else:
                    comm.p2p_enable(True)
                    comm.p2p_set_timeout(10.0)  # the ranks are in step here (the agreement above): a peer's words are microseconds away or never come
                    good, res = step(lambda: probe_ok(comm))
                    good = good and bool(res) and not comm.p2p_status()["timed_out"]
                    try:  # the running timeout: the user's TMA_P2P_TIMEOUT_S, else 120 s (the 10 s above served the self-check only)
                        comm.p2p_set_timeout(float(os.environ.get("TMA_P2P_TIMEOUT_S") or 120.0))
                    except ValueError:
                        comm.p2p_set_timeout(120.0)
                    if not agree(good):
                        note = "self-check of the exchange against the expected sums failed"
                    elif not has_rccl or os.environ.get("TMA_P2P") == "1":
                        on, note = True, "on (TMA_P2P=1)" if has_rccl else "on (the communicator's only path)"
                    else:
                        def per_call_us(n: int = 40) -> float:
                            g = torch.zeros(P, dtype=torch.float32, device=self.device)
                            for _ in range(5):
                                comm.all_reduce_(g, self._stream())
                            torch.cuda.current_stream(self.device).synchronize()
                            _dist.barrier()
                            t0 = time.perf_counter()
                            for _ in range(n):
                                comm.all_reduce_(g, self._stream())
                            torch.cuda.current_stream(self.device).synchronize()
                            return _dist.allreduce_max_float((time.perf_counter() - t0) / n * 1e6, device=self.device)

                        t_p2p = per_call_us()
                        comm.p2p_enable(False)
                        t_rccl = per_call_us()
                        # (the contest runs the exchange as two launches of its own; fused into the H = 64 chain it has none -- 4.7 us per minibatch less,
                        # measured at world size 1, tools/r05_run4.sh -- so a near tie goes to the exchange there)
                        margin = 3.0 if self.policy.dims.hidden == 64 else 0.0
                        on = agree(t_p2p < t_rccl + margin)
                        note = (f"{'on' if on else 'off'}: {t_p2p:.1f} us per {4 * P}-byte all-reduce against RCCL's {t_rccl:.1f} us (dependent chain of 40, "
                                "stand-alone launches both ways, max over ranks)")
        if comm.p2p_status()["slot_words"] and comm.p2p_enabled != on:
            try:
                comm.p2p_enable(on)
            except Exception:  # noqa: BLE001  (a communicator without RCCL cannot switch it off: the caller drops the communicator)
                comm.p2p_enabled = False
        comm.p2p_note = note
        if self.rank == 0 and (self.world_size > 1 or os.environ.get("TMA_P2P") in ("1", "auto")):
            print(f"three-mlagents_amd: peer exchange for the gradient all-reduce {note}", file=sys.stderr, flush=True)

    @property
    def allreduce_path(self) -> str:
        """Which implementation sums the gradient across ranks: "none" (one rank), "torch.distributed" (callback per minibatch), "rccl" (the
        library's own communicator) or "peer_exchange" (tma_comm_p2p_*).  Fixed at construction; part of a run's provenance (summation order)."""
        if self.world_size == 1 and self._native_comm is None:
            return "none"
        if self._native_comm is None:
            return "torch.distributed"
        return "peer_exchange" if self._native_comm.p2p_enabled else "rccl"

    def check_collectives(self) -> None:
        """Raise (on every rank that sees it) when a peer-exchange pull gave up waiting for a peer: the gradient it delivered is NaN and so is
        every parameter since.  Called behind the stream synchronisations of learn() and in front of save(): a checkpoint is never written from
        such a state without an exception."""
        comm = self._native_comm
        if comm is not None and comm.p2p_enabled and comm.p2p_status()["timed_out"]:
            raise RuntimeError("three-mlagents_amd: a peer-exchange all-reduce timed out waiting for a peer's words (rank skew beyond "
                               "TMA_P2P_TIMEOUT_S, or a peer died); the gradient of that minibatch -- and every parameter since -- is NaN.  "
                               "Restart from the last checkpoint; TMA_P2P=0 keeps the collectives on RCCL.")

    def _stream(self):
        return _lib.stream_ptr(self.device)

    def get_env(self):
        return self.env

    # -- rollout --------------------------------------------------------------------------
    def collect_rollouts(self, callback=None, chunk: int | None = None) -> bool:
        """SB3 OnPolicyAlgorithm.collect_rollouts: n_steps vector steps through the native driver, then GAE."""
        L, T, eng = _lib.lib(), self.n_steps, self.env.engine
        if not self._last_obs_valid:
            eng.reset(self.buf["obs"][0])
            self._last_obs_valid = True
        else:
            self.buf["obs"][0].copy_(self.buf["obs"][T])
        chunk = T if (chunk is None or callback is None) else max(1, int(chunk))
        t = 0
        keep_going = True
        while t < T and keep_going:
            te = min(T, t + chunk)
            _lib.check(L.tma_rollout_collect(eng._h, _lib.ptr(self.policy.params), C.byref(self.policy.dims), C.byref(self._rb), t, te, T,
                                             self.seed & 0xFFFFFFFF, (self._rollout_counter * T) & 0xFFFFFFFF, eng.env_offset & 0xFFFFFFFF,
                                             self.gamma, 1, 0, self._stream()))
            if callback is not None:
                # one call per native chunk (callbacks.BaseCallback.on_steps: a per-step loop unless the callback knows how to take the
                # steps in bulk -- EvalCallback does; 1024 Python round trips per rollout are 2.5 ms at 4096 envs)
                done_steps, keep_going = callback.on_steps(te - t, self.n_envs * self.world_size)
                te = t + done_steps
            else:
                self.num_timesteps += (te - t) * self.n_envs * self.world_size
            t = te
        self._rollout_counter += 1  # sampling counters (rng_step0) are never reused, also after an interrupted rollout
        if not keep_going:
            # a callback stopped the rollout: the envs already advanced to step t, whose observation sits in slot t of the buffer --
            # keep slot T current (SB3 keeps _last_obs current on every step) so a later learn(reset_num_timesteps=False) continues
            if t < T:
                self.buf["obs"][T].copy_(self.buf["obs"][t])
            return False
        b = self.buf
        _lib.check(L.tma_gae_flags(_lib.ptr(b["rewards"]), _lib.ptr(b["values"]), _lib.ptr(b["terminated"]), _lib.ptr(b["truncated"]),
                                   _lib.ptr(b["last_values"]), self.gamma, self.gae_lambda, T, self.n_envs, _lib.ptr(b["advantages"]),
                                   _lib.ptr(b["returns"]), self._stream()))
        if self._packed is not None:  # one streaming pass per rollout: the gradient kernels then read one record per sample.  Done HERE, right
            # behind the advantages / returns it copies, so that the `packed` pointer of the rollout view is never stale for a direct
            # tma_ppo_minibatch_grad caller between collect_rollouts() and train()
            _lib.check(L.tma_ppo_pack_samples(C.byref(self._rollout_view), C.byref(self.policy.dims), _lib.ptr(self._packed), self._stream()))
        return True

    # -- update ---------------------------------------------------------------------------
    def train(self) -> None:
        """SB3 PPO.train: n_epochs passes over the permuted rollout in minibatches of batch_size."""
        L = _lib.lib()
        total = self.n_steps * self.n_envs
        scale = 1.0 / self.world_size
        perm_seed = (self.seed * 2654435761 + 12345) & 0xFFFFFFFF
        can_prepare = total <= (1 << 22) and self.batch_size >= 256  # limits of tma_ppo_epoch_prepare (include/tma.h)
        # data parallel: every rank normalises a minibatch's advantages with the mean / std of the GLOBAL minibatch (the rows of all
        # ranks), as one SB3 run over the concatenated batch would -- one all-reduce of 16 B per minibatch, once per epoch
        global_stats = self.world_size > 1 and self.normalize_advantage
        assert can_prepare or not global_stats  # (checked in _setup_model)
        if global_stats and getattr(self, "_adv_sums", None) is None:
            self._adv_sums = torch.zeros(2 * ((total + self.batch_size - 1) // self.batch_size), dtype=torch.float64, device=self.device)
        if self.world_size == 1 and not os.environ.get("TMA_DP_PATH"):  # (TMA_DP_PATH=1: time the data-parallel host loop on one GPU)
            # one GPU: a whole epoch (prepare + every minibatch's gradient and optimizer step) is issued natively by ONE call -- at the
            # reference's literal batch_size = 256 that is 16 384 optimizer steps without a host-language round trip in between
            # (round 6: ALL the epochs by one call -- where a persistent epoch kernel takes the shape and the epochs' sample offsets fit the
            #  workspace, tma_ppo_train_epochs_local runs them as ONE launch: the reference's own 1- / 8-env schedules are 4 / 32 optimizer
            #  steps an epoch.  TMA_EPOCH_PER_CALL=1: one call per epoch, round 5's loop, the A/B switch)
            n_mb = (total + self.batch_size - 1) // self.batch_size
            per_call = 1 if os.environ.get("TMA_EPOCH_PER_CALL") or (self._epoch_counter & 0xFFFFFFFF) + self.n_epochs > 0xFFFFFFFF else self.n_epochs
            for _ in range(0, self.n_epochs, per_call):
                _lib.check(L.tma_ppo_train_epochs_local(_lib.ptr(self.policy.params), C.byref(self.policy.dims), C.byref(self._rollout_view), perm_seed,
                                                        self._epoch_counter & 0xFFFFFFFF, per_call, self.batch_size, C.byref(self._hp), _lib.ptr(self.grad),
                                                        _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self._adam_step + 1, self.learning_rate, 0.9,
                                                        0.999, 1e-5, self.max_grad_norm, _lib.ptr(self.workspace), self._stream()))
                self._adam_step += n_mb * per_call
                self._epoch_counter += per_call
            self._n_updates += self.n_epochs
            self._mark_update_done()
            return
        for _ in range(self.n_epochs):  # data parallel: per-minibatch calls around the collectives
            if can_prepare:  # one launch per epoch: sample offsets of the permutation + advantage partials of every minibatch
                ep = _lib.Minibatch(None, perm_seed, self._epoch_counter & 0xFFFFFFFF, 0, total, 0)
                _lib.check(L.tma_ppo_epoch_prepare(C.byref(self._rollout_view), C.byref(ep), self.batch_size, C.byref(self.policy.dims),
                                                   _lib.ptr(self.workspace), self._stream()))
                if global_stats:
                    import torch.distributed as tdist

                    for direction in (0, 1):  # export the per-minibatch (sum, sumsq) pairs, all-reduce, import
                        _lib.check(L.tma_ppo_epoch_adv_sums(_lib.ptr(self.workspace), C.byref(self.policy.dims), self.batch_size, total,
                                                            _lib.ptr(self._adv_sums), direction, self._stream()))
                        if direction == 0:
                            if self._native_comm is not None:
                                self._native_comm.all_reduce_(self._adv_sums, self._stream())
                            else:
                                self._timed_all_reduce(self._adv_sums, "adv_allreduce_us")
            # the minibatch loop itself is native (tma_ppo_train_epoch_dp): gradient launches, this callback, optimizer launches.  The callback is
            # the only host-language call per minibatch.  RCCL sum over xGMI, scaled by 1/world inside the optimizer arithmetic; with the nccl
            # backend the collective runs on the process group's own stream and is ordered against the compute stream through events on both
            # sides -- the host returns as soon as it is queued.
            if getattr(self, "_allreduce_cb", None) is None:
                def _cb(_ctx, _buf, _count, self=self):
                    try:
                        if self.world_size > 1:
                            self._timed_all_reduce(self.grad, "grad_allreduce_us")
                        return 0
                    except Exception as exc:  # (an exception must not unwind through the C frames)
                        self._allreduce_error = exc
                        return 1

                self._allreduce_cb = _lib.AllReduceFn(_cb)
            n_mb = (total + self.batch_size - 1) // self.batch_size
            self._allreduce_error = None
            cb, ctx = self._allreduce_cb, None
            if self._native_comm is not None:  # ncclAllReduce from inside the native loop, on the stream the kernels run on
                self._native_comm.bind_stream(self._stream())
                cb, ctx = self._native_comm.callback, self._native_comm.ctx
            rc = L.tma_ppo_train_epoch_dp(_lib.ptr(self.policy.params), C.byref(self.policy.dims), C.byref(self._rollout_view), perm_seed,
                                          self._epoch_counter & 0xFFFFFFFF, self.batch_size, self.batch_size if can_prepare else 0,
                                          self.world_size if global_stats else 0, C.byref(self._hp), _lib.ptr(self.grad), _lib.ptr(self.exp_avg),
                                          _lib.ptr(self.exp_avg_sq), self._adam_step + 1, self.learning_rate, 0.9, 0.999, 1e-5, self.max_grad_norm, scale,
                                          cb, ctx, _lib.ptr(self.workspace), self._stream())
            if self._allreduce_error is not None:
                raise self._allreduce_error
            _lib.check(rc)
            self._adam_step += n_mb
            self._epoch_counter += 1
        self._n_updates += self.n_epochs
        self._mark_update_done()

    def _mark_update_done(self) -> None:
        """Event behind the last launch of train(): what a reader of the parameters on ANOTHER stream waits for (EvalCallback evaluates on a
        side stream while the next rollout already runs on the compute stream)."""
        ev = getattr(self, "_update_done", None)
        if ev is None:
            ev = self._update_done = torch.cuda.Event(enable_timing=False)
        ev.record(torch.cuda.current_stream(self.device))

    def side_stream(self):
        """A second stream of the model's device for work that only READS what the compute stream produced earlier (deterministic evaluation,
        the episode-log read-back): it runs beside the compute stream's next rollout / update instead of between two iterations."""
        st = getattr(self, "_side_stream", None)
        if st is None:
            st = self._side_stream = torch.cuda.Stream(self.device)
        return st

    def _timed_all_reduce(self, tensor: torch.Tensor, key: str) -> None:
        import torch.distributed as tdist

        ev = self._dp_events[key] if self.dp_timing is not None and len(self._dp_events[key]) < self.dp_timing_samples else None
        if ev is None:
            tdist.all_reduce(tensor)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # on the compute stream, behind the kernel that produced `tensor`
        tdist.all_reduce(tensor)
        e1.record()  # behind the compute stream's wait for the collective
        ev.append((e0, e1))

    def dp_timing_collect(self) -> dict:
        """Median / max duration (us, HIP events on the compute stream: producer kernel done -> reduced tensor usable) of the collectives
        recorded since the last call.  Synchronises the device."""
        torch.cuda.synchronize(self.device)
        out = {}
        if self._native_comm is not None:
            us, calls = self._native_comm.pop_timing()
            us.sort()
            out["grad_allreduce_us"] = {"calls_timed": len(us), "median_us": us[len(us) // 2] if us else None, "max_us": us[-1] if us else None,
                                        "bytes": int(self.grad.numel() * 4),
                                        "path": ("native: peer exchange (direct stores into the peers' inboxes, fused into the slab reduction / sum-of-squares kernels; "
                                                 "timed: the receiving kernel)" if self._native_comm.p2p_enabled
                                                 else "native: ncclAllReduce issued by libtma_hip.so on the compute stream"),
                                        "peer_exchange": self._native_comm.p2p_note, "allreduces_issued": calls}
            out["adv_allreduce_us"] = {"calls_timed": 0, "median_us": None, "max_us": None, "bytes": int(getattr(self, "_adv_sums", torch.empty(0)).numel() * 8),
                                       "path": "native (f64, same communicator; not timed separately: one per epoch)"}
            return out
        for key, evs in self._dp_events.items():
            us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            out[key] = {"calls_timed": len(us), "median_us": us[len(us) // 2] if us else None, "max_us": us[-1] if us else None,
                        "bytes": int(self.grad.numel() * 4) if key.startswith("grad") else int(getattr(self, "_adv_sums", torch.empty(0)).numel() * 8)}
            evs.clear()
        return out

    def pop_train_stats(self) -> dict[str, float]:
        out = (C.c_double * 8)()
        _lib.check(_lib.lib().tma_ppo_pop_stats(_lib.ptr(self.workspace), out, self._stream()))
        self.check_collectives()  # (pop_stats synchronised the stream)
        n = max(out[5], 1.0)
        fb = C.c_int64(0)
        _lib.check(_lib.lib().tma_ppo_persist_fallbacks(_lib.ptr(self.workspace), C.byref(fb), self._stream()))
        if fb.value:  # epochs the persistent batch-256 kernel handed back to the per-minibatch launches (include/tma.h)
            self.persist_fallbacks = int(fb.value)
        return {"train/policy_gradient_loss": out[0] / n, "train/value_loss": out[1] / n, "train/entropy_loss": -out[2] / n,
                "train/approx_kl": out[3] / n, "train/clip_fraction": out[4] / n, "train/grad_norm": out[6], "train/n_samples": out[5],
                **({"train/persist_fallbacks": float(fb.value)} if fb.value else {})}

    # -- learn ----------------------------------------------------------------------------
    def learn(self, total_timesteps: int, callback=None, log_interval: int = 1, tb_log_name: str = "PPO", reset_num_timesteps: bool = True,
              progress_bar: bool = False):
        from .callbacks import as_callback

        cb = as_callback(callback)
        if reset_num_timesteps:
            self.num_timesteps = 0
        else:  # SB3 _setup_learn: "make sure training timesteps are ahead of the internal counter"
            total_timesteps = int(total_timesteps) + self.num_timesteps
        self._total_timesteps = int(total_timesteps)
        cb.init_callback(self)
        cb.on_training_start(locals(), globals())
        t0 = time.time()
        iteration = 0
        if getattr(self.env, "monitor_dir", None) and self.rank == 0 and getattr(self.env.engine, "_log_cap", 0) == 0:
            self.env.engine.episode_log(self.monitor_log_capacity)  # per-episode Monitor rows (r, l, t)
        t_prev = 0.0
        # Logging that does not drain the GPU (round 4).  Per logged iteration: the Monitor aggregates / episode records of the rollout are
        # detached (a host-side buffer swap) and read back on a side stream behind the rollout WHILE the update runs; the update's statistic
        # slots are copied to pinned memory and cleared stream-ordered; the row of iteration k (progress.csv / TensorBoard / logger_values /
        # the verbose line) is finished one iteration later, when update k is long over -- so rollout k + 1 is queued before anything waits.
        # TMA_SYNC_LOGGING=1 restores the synchronous order (pop, log, then the next rollout).
        pipelined = not os.environ.get("TMA_SYNC_LOGGING")
        eng = self.env.engine
        if pipelined:  # (a learn() that unwound between a detach and its pop left a detached set behind: drop it)
            try:
                eng.pop_detached_episode_log()
            except ValueError:
                pass
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(self.device))
        pending = None
        n_logged = 0

        def finish(p):
            p["ev_train"].synchronize()
            self.check_collectives()
            stats = self._fold_train_stats(p["staging"])
            s_ret, s_len, cnt = p["ep"]
            elapsed = max(ev0.elapsed_time(p["ev_train"]) * 1e-3, 1e-9)  # device timeline: learn() start -> the end of this iteration's update
            stats.update({"time/fps": p["num_timesteps"] / elapsed, "time/iterations": p["iteration"], "time/total_timesteps": p["num_timesteps"],
                          "rollout/ep_rew_mean": p["window"][0], "rollout/ep_len_mean": p["window"][1],
                          "rollout/episodes": cnt, "train/n_updates": p["n_updates"]})
            self.logger_values = stats
            self._write_progress(stats, p["num_timesteps"])
            if self.verbose >= 1 and self.rank == 0:
                print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in stats.items()}), flush=True)

        try:
            while self.num_timesteps < total_timesteps:
                cb.on_rollout_start()
                if not self.collect_rollouts(cb if callback is not None else None, chunk=getattr(cb, "chunk_steps", None)):
                    break
                cb.on_rollout_end()
                iteration += 1
                logging = log_interval is not None and iteration % log_interval == 0
                if logging and pipelined:
                    eng.detach_episode_log()
                    ev_roll = torch.cuda.Event()
                    ev_roll.record(torch.cuda.current_stream(self.device))
                self.train()
                cb.on_update_queued()  # (EvalCallback collects its deferred evaluation here: host work under the update the GPU is now running)
                if logging and pipelined:
                    # (order matters: device-to-host copies of every stream share one in-order copy queue, so the side stream's read-back goes
                    #  in BEFORE the compute stream's statistics copy, which sits behind the whole update -- queued the other way round the
                    #  read-back waited 21 ms for it)
                    side = self.side_stream()
                    side.wait_event(ev_roll)
                    ep, lr_, ll_, le_, seen = eng.pop_detached_episode_log(C.c_void_p(side.cuda_stream))  # waits for the ROLLOUT only
                    staging = self._stats_staging(n_logged & 1)  # (alternates per LOGGED iteration: row k's buffer is folded before row k + 2 is copied into it)
                    n_logged += 1
                    _lib.check(_lib.lib().tma_ppo_stats_enqueue(_lib.ptr(self.workspace), _lib.ptr(staging), self._stream()))
                    ev_train = torch.cuda.Event(enable_timing=True)
                    ev_train.record(torch.cuda.current_stream(self.device))
                    now = time.time() - t0
                    self._write_monitor(ep[0], ep[1], ep[2], t_prev, now, t0, log=(lr_, ll_, le_, seen))
                    t_prev = now
                    if pending is not None:
                        finish(pending)
                    pending = dict(ev_train=ev_train, staging=staging, ep=ep, iteration=iteration, num_timesteps=self.num_timesteps, n_updates=self._n_updates,
                                   window=self._episode_window(lr_, ll_, ep[0], ep[1], ep[2]))
                elif logging:
                    s_ret, s_len, cnt = eng.pop_episode_stats()
                    log = eng.pop_episode_log() if getattr(eng, "_log_cap", 0) > 0 else None
                    win = self._episode_window(log[0] if log else None, log[1] if log else None, s_ret, s_len, cnt)
                    stats = self.pop_train_stats()
                    fps = self.num_timesteps / max(time.time() - t0, 1e-9)
                    stats.update({"time/fps": fps, "time/iterations": iteration, "time/total_timesteps": self.num_timesteps,
                                  "rollout/ep_rew_mean": win[0], "rollout/ep_len_mean": win[1],
                                  "rollout/episodes": cnt, "train/n_updates": self._n_updates})
                    self.logger_values = stats
                    self._write_progress(stats, self.num_timesteps)
                    self._write_monitor(s_ret, s_len, cnt, t_prev, time.time() - t0, t0, log=log)
                    t_prev = time.time() - t0
                    if self.verbose >= 1 and self.rank == 0:
                        print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in stats.items()}), flush=True)
            if pending is not None:
                finish(pending)
                pending = None
        except BaseException:
            # learn() is unwinding (an all-reduce error, KeyboardInterrupt, a callback that raised): on_training_end will not run, so what the
            # callbacks and the Monitor writer hold in memory goes to disk HERE -- SB3 writes evaluations.npz after every evaluation
            self._join_monitor_writer(reraise=False)
            flush = getattr(cb, "flush", None)
            if callable(flush):
                try:
                    flush()
                except Exception:  # noqa: BLE001 -- never mask the exception that is unwinding
                    pass
            raise
        self._join_monitor_writer()
        cb.on_training_end()
        return self

    def _episode_window(self, returns, lengths, s_ret: float, s_len: float, cnt: float) -> tuple[float, float]:
        """(ep_rew_mean, ep_len_mean) as SB3 logs them: the mean of the last `stats_window_size` finished episodes (this interval's records in the
        order the kernels logged them, appended to what earlier intervals left); the interval mean when there is no per-episode log."""
        if not self.stats_window_size:
            return (s_ret / cnt, s_len / cnt) if cnt else (float("nan"), float("nan"))
        if returns is not None and len(returns):
            w = self.stats_window_size
            self._ep_info_r.extend(np.asarray(returns[-w:], np.float64).tolist())
            self._ep_info_l.extend(np.asarray(lengths[-w:], np.float64).tolist())
        if len(self._ep_info_r):
            return float(np.mean(self._ep_info_r)), float(np.mean(self._ep_info_l))
        return (s_ret / cnt, s_len / cnt) if cnt else (float("nan"), float("nan"))

    def _stats_staging(self, which: int) -> torch.Tensor:
        bufs = getattr(self, "_staging_bufs", None)
        if bufs is None:
            n = int(_lib.lib().tma_ppo_stats_staging_bytes())
            bufs = self._staging_bufs = [torch.zeros(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
        return bufs[which]

    def _fold_train_stats(self, staging: torch.Tensor) -> dict[str, float]:
        out = (C.c_double * 8)()
        _lib.check(_lib.lib().tma_ppo_stats_fold(_lib.ptr(staging), out))
        n = max(out[5], 1.0)
        return {"train/policy_gradient_loss": out[0] / n, "train/value_loss": out[1] / n, "train/entropy_loss": -out[2] / n,
                "train/approx_kl": out[3] / n, "train/clip_fraction": out[4] / n, "train/grad_norm": out[6], "train/n_samples": out[5]}

    def _write_progress(self, stats: dict, step: int | None = None) -> None:
        """Scalar log per iteration (the keys SB3's logger writes: rollout/*, train/*, time/*) under tensorboard_log: `progress.csv` and a
        TensorBoard event file in `<tb_log_name>_1/` (tb_events.py; SB3 sends the same scalars through its TensorBoard output format)."""
        if not self.tensorboard_log or self.rank != 0:
            return
        os.makedirs(self.tensorboard_log, exist_ok=True)
        path = os.path.join(self.tensorboard_log, "progress.csv")
        keys = sorted(stats)
        new = not os.path.exists(path)
        with open(path, "a", encoding="utf-8") as f:
            if new:
                f.write(",".join(keys) + "\n")
            f.write(",".join(repr(float(stats[k])) for k in keys) + "\n")
        if getattr(self, "_tb_writer", None) is None:
            from .tb_events import EventWriter

            self._tb_writer = EventWriter(os.path.join(self.tensorboard_log, "PPO_1"))
        self._tb_writer.add_scalars(stats, self.num_timesteps if step is None else int(step))

    monitor_log_capacity = 1 << 20  # episode records the device keeps between two log intervals
    monitor_max_rows = 100_000      # rows written per log interval (an evenly strided subsample beyond that)

    # up to this many envs: one `<rank>.monitor.csv` per env, as the reference writes them (training.py:84-86); TMA_MONITOR_PER_ENV_LIMIT raises it
    # (round 6: any vector size -- 4096 envs are 4096 files appended per log interval on the writer thread, which is why it is not the default)
    monitor_per_env_limit = int(os.environ.get("TMA_MONITOR_PER_ENV_LIMIT") or 64)

    def _write_monitor(self, sum_ret: float, sum_len: float, count: float, t_begin: float, t_end: float, t_start: float, log=None) -> None:
        """SB3 Monitor files under the directory make_vector_env passes (reference training.py:84-86: env `rank` of the vector is wrapped in
        `Monitor(env, monitor_dir / f"{rank}")` -> `<rank>.monitor.csv`): the JSON header line, `r,l,t`, then one row per finished episode.
        Up to `monitor_per_env_limit` envs every env gets its own file, as in the reference -- the device episode log carries the env index;
        beyond that ONE `0.monitor.csv` holds the rows of every env in the order the kernels logged them (`load_results` concatenates per-env
        files anyway; 4096 files per run would not be a service to anyone).  `t` is interpolated over the log interval -- the device does not
        stamp wall-clock time.  More than monitor_max_rows episodes in one interval are subsampled with an even stride and a `#` comment line
        records how many finished; if the device log overflowed, the interval's mean is added as a comment as well."""
        mdir = getattr(self.env, "monitor_dir", None)
        if not mdir or self.rank != 0 or not count:
            return
        os.makedirs(mdir, exist_ok=True)
        if getattr(self.env.engine, "_log_cap", 0) <= 0:
            return
        r, l, e, seen = log if log is not None else self.env.engine.pop_episode_log()  # (log: what pop_detached_episode_log read on the side stream)
        n = len(r)
        keep = np.arange(n) if n <= self.monitor_max_rows else np.linspace(0, n - 1, self.monitor_max_rows).astype(np.int64)
        self._join_monitor_writer()  # (rows of the previous interval are on disk before this interval's header / comment lines)
        per_env = self.n_envs <= self.monitor_per_env_limit
        header = lambda rank: "#" + json.dumps({"t_start": t_start, "env_id": getattr(self.env, "task_id", None)}) + "\nr,l,t\n"  # noqa: E731
        ts_all = t_begin + (t_end - t_begin) * (np.arange(len(keep)) + 1.0) / max(len(keep), 1)
        jobs = []  # (path, returns f64, lengths i32, t f64)
        if per_env:
            written = getattr(self, "_monitor_files", None)
            if written is None:
                written = self._monitor_files = set()
            # rows grouped by env with ONE stable sort (the order inside an env stays the order the kernels logged them): any vector size
            order = np.argsort(e[keep], kind="stable")
            bounds = np.searchsorted(e[keep][order], np.arange(self.n_envs + 1))
            for rank in range(self.n_envs):  # the reference creates every env's file at construction, finished episodes or not
                path = os.path.join(str(mdir), f"{rank}.monitor.csv")
                if rank not in written:
                    with open(path, "a", encoding="utf-8") as f:
                        if f.tell() == 0:
                            f.write(header(rank))
                    written.add(rank)
                pos = np.sort(order[bounds[rank]:bounds[rank + 1]])
                if len(pos):
                    rows = keep[pos]
                    jobs.append((path, np.ascontiguousarray(r[rows], np.float64), np.ascontiguousarray(l[rows], np.int32),
                                 np.ascontiguousarray(ts_all[pos], np.float64)))
        else:
            path = os.path.join(str(mdir), "0.monitor.csv")
            new = not os.path.exists(path)
            with open(path, "a", encoding="utf-8") as f:
                if new:
                    f.write(header(0))
                if len(keep) < seen:
                    f.write(f"# {seen} episodes finished in this interval, {len(keep)} rows kept; interval mean r={sum_ret / count:.6f} l={sum_len / count:.3f}\n")
            if len(keep):
                jobs.append((path, np.ascontiguousarray(r[keep], np.float64), np.ascontiguousarray(l[keep], np.int32),
                             np.ascontiguousarray(ts_all, np.float64)))
        if not jobs:
            return
        # the rows themselves: formatted and appended natively (tma_monitor_append_rows) on a writer thread -- the ctypes call releases the
        # GIL, so 10^5 rows per iteration (4096 envs, ~30-step episodes) cost the training loop nothing
        import threading

        def work():
            try:
                for path, rr, ll, ts in jobs:
                    _lib.check(_lib.lib().tma_monitor_append_rows(path.encode(), rr.ctypes.data_as(C.c_void_p), ll.ctypes.data_as(C.c_void_p),
                                                                  ts.ctypes.data_as(C.c_void_p), len(rr)))
            except BaseException as exc:  # noqa: BLE001 -- handed to the training thread by _join_monitor_writer
                self._monitor_error = exc

        self._monitor_thread = threading.Thread(target=work, name="tma-monitor-writer", daemon=False)
        self._monitor_thread.start()

    def _join_monitor_writer(self, reraise: bool = True) -> None:
        th = getattr(self, "_monitor_thread", None)
        if th is not None:
            th.join()
            self._monitor_thread = None
        err = getattr(self, "_monitor_error", None)
        if err is not None and reraise:  # a failed row write surfaces in the training thread (an exception inside the thread is only printed)
            self._monitor_error = None
            raise RuntimeError(f"Monitor writer failed: {err}") from err

    # -- inference ------------------------------------------------------------------------
    def predict(self, observation, state=None, episode_start=None, deterministic: bool = False):
        """BaseAlgorithm.predict: obs [D] or [n, D] (numpy or tensor) -> (action(s), None)."""
        obs = torch.as_tensor(np.asarray(observation, dtype=np.float32) if not torch.is_tensor(observation) else observation)
        single = obs.dim() == 1
        if single:
            obs = obs.unsqueeze(0)
        if obs.dim() != 2 or obs.shape[1] != self.policy.obs_dim:
            raise ValueError(f"observation must have shape [{self.policy.obs_dim}] or [n, {self.policy.obs_dim}], got {tuple(np.shape(observation))}")
        self._predict_counter = getattr(self, "_predict_counter", 0) + 1
        actions, _, _ = self.policy.act(obs, rng_seed=self.seed ^ 0x5EED, rng_step=self._predict_counter, deterministic=deterministic)
        a = actions.cpu().numpy()
        if self.policy.continuous:
            a = np.clip(a, -1.0, 1.0)
        else:
            a = a.astype(np.int64)
        return (a[0] if single else a), None

    # -- artefacts (SB3 zip member names) ---------------------------------------
    def _data(self) -> dict[str, Any]:
        """The `data` member of the zip in stable-baselines3's vocabulary: what BaseAlgorithm.load puts into `model.__dict__` before
        `_setup_model` (sb3_format.py builds the three members that must be live objects).  Engine-only values sit under "tma"."""
        from . import sb3_format
        from .spaces import Box, Discrete

        pol = self.policy
        env = self.env
        obs_space = getattr(env, "observation_space", None) or Box(-np.inf, np.inf, (pol.obs_dim,), np.float32)
        act_space = getattr(env, "action_space", None) or (Box(-1.0, 1.0, (pol.act_dim,), np.float32) if pol.continuous else Discrete(pol.act_dim))
        data = {
            "policy_kwargs": {"net_arch": [pol.hidden, pol.hidden]}, "num_timesteps": self.num_timesteps, "_total_timesteps": getattr(self, "_total_timesteps", 0),
            "_num_timesteps_at_start": 0, "seed": self.seed, "action_noise": None, "start_time": time.time_ns(), "learning_rate": self.learning_rate,
            "tensorboard_log": self.tensorboard_log, "_last_obs": None, "_last_episode_starts": None, "_last_original_obs": None, "_episode_num": 0,
            "use_sde": False, "sde_sample_freq": -1, "_current_progress_remaining": 0.0, "_stats_window_size": 100, "ep_info_buffer": None,
            "ep_success_buffer": None, "_n_updates": self._n_updates, "n_envs": getattr(self, "n_envs", 1), "n_steps": self.n_steps, "gamma": self.gamma,
            "gae_lambda": self.gae_lambda, "ent_coef": self.ent_coef, "vf_coef": self.vf_coef, "max_grad_norm": self.max_grad_norm,
            "rollout_buffer_class": None, "rollout_buffer_kwargs": {}, "batch_size": self.batch_size, "n_epochs": self.n_epochs, "clip_range": self.clip_range,
            "clip_range_vf": None, "normalize_advantage": self.normalize_advantage, "target_kl": None, "verbose": self.verbose, "_custom_logger": False,
            "tma": {"engine": "three-mlagents_amd", "task_id": getattr(env, "task_id", None), "mfma_dtype": self.policy_kwargs.get("mfma_dtype", "f32"),
                    "adam_step": self._adam_step, "allreduce_path": self.allreduce_path if hasattr(self, "_native_comm") else "none",
                    "world_size": getattr(self, "world_size", 1), "obs_dim": pol.obs_dim, "act_dim": pol.act_dim, "continuous": pol.continuous, "hidden": pol.hidden},
        }
        data.update(sb3_format.data_members(obs_space, act_space))
        return data

    def freeze_for_save(self) -> dict:
        """Everything save() reads, as of NOW in stream order: device clones of the parameter buffer (weight images included: a deferred evaluation
        runs on it) and of the optimizer moments, queued on the current stream, plus the host-side members.  save(path, _frozen=...) writes the zip
        from it later -- EvalCallback's best_model.zip of an evaluation whose result arrives after train() has already moved the live buffers."""
        has_moments = getattr(self, "exp_avg", None) is not None
        return {"params": self.policy.params.clone(), "exp_avg": self.exp_avg.clone() if has_moments else None,
                "exp_avg_sq": self.exp_avg_sq.clone() if has_moments else None, "adam_step": self._adam_step if has_moments else 0,
                "data": json.dumps(self._data(), indent=2, default=str)}

    def save(self, path, exclude=None, include=None, _frozen: dict | None = None) -> None:
        path = str(path)
        if not os.path.splitext(path)[1]:
            path += ".zip"
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

        def _pth(obj) -> bytes:
            bio = io.BytesIO()
            torch.save(obj, bio)
            return bio.getvalue()

        from . import sb3_format

        fz = _frozen
        if fz is None and getattr(self, "_native_comm", None) is not None:
            torch.cuda.current_stream(self.device).synchronize()
            self.check_collectives()
        sd = self.policy.state_dict() if fz is None else self.policy.named_from_flat(fz["params"][: self.policy.n_trainable])
        order = sb3_format.parameter_order(self.policy.continuous)
        sd = {k: sd[k] for k in order}  # torch's registration order of an ActorCriticPolicy: also the optimizer's parameter indices
        exp_avg, exp_avg_sq = (getattr(self, "exp_avg", None), getattr(self, "exp_avg_sq", None)) if fz is None else (fz["exp_avg"], fz["exp_avg_sq"])
        has_moments = exp_avg is not None  # (a model loaded without an env has a policy but no optimizer state)
        adam_step = (self._adam_step if fz is None else fz["adam_step"]) if has_moments else 0
        opt = sb3_format.adam_state_dict(order, self.policy.named_from_flat(exp_avg) if has_moments else {},
                                         self.policy.named_from_flat(exp_avg_sq) if has_moments else {}, adam_step, self.learning_rate)
        with zipfile.ZipFile(path, "w") as z:  # stored, not deflated: what SB3's save_to_zip_file writes (and EvalCallback saves a zip per new best)
            z.writestr("data", json.dumps(self._data(), indent=2, default=str) if fz is None else fz["data"])
            z.writestr("policy.pth", _pth(sd))
            z.writestr("policy.optimizer.pth", _pth(opt))
            z.writestr("pytorch_variables.pth", _pth({}))
            z.writestr("_stable_baselines3_version", "2.9.0+three-mlagents_amd")
            z.writestr("system_info.txt", f"OS: {platform.platform()}\nPython: {platform.python_version()}\nPyTorch: {torch.__version__}\nGPU Enabled: True\n")

    @classmethod
    def load(cls, path, env=None, device="auto", **kwargs) -> "PPO":
        path = str(path)
        if not os.path.exists(path) and os.path.exists(path + ".zip"):
            path += ".zip"
        if not os.path.exists(path):
            raise FileNotFoundError(f"Model not found: {path}")
        with zipfile.ZipFile(path) as z:
            data = json.loads(z.read("data").decode())
            sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
            try:
                # the payload this engine writes is dicts / tuples / ints / tensors: the restricted unpickler is enough, and a zip whose
                # name came from an API caller never executes pickled code.  Anything else (a zip written by SB3 itself) -> no optimizer state
                opt = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
                if not isinstance(opt, dict):
                    opt = {}
            except Exception:  # noqa: BLE001
                opt = {}
        # zips written by stable-baselines3 carry their own `data` schema: recover the policy shape from the state_dict,
        # whose keys/shapes are SB3's (mlp_extractor.policy_net.{0,2}, mlp_extractor.value_net.{0,2}, action_net, value_net, log_std)
        w1, wa = sd["mlp_extractor.policy_net.0.weight"], sd["action_net.weight"]
        tma_extra = data.get("tma") if isinstance(data.get("tma"), dict) else {}
        data.setdefault("obs_dim", int(w1.shape[1]))
        data.setdefault("hidden", int(w1.shape[0]))
        data.setdefault("act_dim", int(wa.shape[0]))
        data.setdefault("continuous", "log_std" in sd)
        if sd["mlp_extractor.policy_net.2.weight"].shape != (data["hidden"], data["hidden"]) or sd["mlp_extractor.value_net.0.weight"].shape != w1.shape:
            raise ValueError("unsupported net_arch in policy zip: the engine needs two equal hidden layers for pi and vf")

        pk = data.get("policy_kwargs")
        mfma = tma_extra.get("mfma_dtype") or (pk.get("mfma_dtype", "f32") if isinstance(pk, dict) else "f32")

        def _num(key, default):
            v = data.get(key, default)
            return default if isinstance(v, dict) or v is None else v  # SB3 stores schedules as pickled objects

        model = cls(data.get("policy_class", "MlpPolicy") if isinstance(data.get("policy_class"), str) else "MlpPolicy", None,
                    learning_rate=_num("learning_rate", 3e-4), n_steps=_num("n_steps", 2048), batch_size=_num("batch_size", 64),
                    n_epochs=_num("n_epochs", 10), gamma=_num("gamma", 0.99), gae_lambda=_num("gae_lambda", 0.95), clip_range=_num("clip_range", 0.2),
                    normalize_advantage=_num("normalize_advantage", True), ent_coef=_num("ent_coef", 0.0), vf_coef=_num("vf_coef", 0.5),
                    max_grad_norm=_num("max_grad_norm", 0.5),
                    policy_kwargs={"net_arch": [data["hidden"], data["hidden"]], "mfma_dtype": mfma}, seed=_num("seed", 0), _init_setup_model=False)
        model.num_timesteps, model._n_updates = data.get("num_timesteps", 0), data.get("_n_updates", 0)
        model._adam_step = int(tma_extra.get("adam_step", data.get("_adam_step", 0)))
        if env is not None:
            model.env = env
            model._setup_model()
            model.policy.load_state_dict(sd)
            state = opt.get("state") or {}
            from . import sb3_format

            order = sb3_format.parameter_order(model.policy.continuous)
            if len(state) == len(order):  # torch.optim.Adam state, indexed in the policy's parameter order (SB3's own zips included)
                model.exp_avg.copy_(model.policy.flat_from_named({k: state[i]["exp_avg"] for i, k in enumerate(order)}).to(model.device))
                model.exp_avg_sq.copy_(model.policy.flat_from_named({k: state[i]["exp_avg_sq"] for i, k in enumerate(order)}).to(model.device))
                model._adam_step = int(float(state[0]["step"]))
        else:
            from .vec_env import _require_gpu

            dev = _require_gpu(None if device == "auto" else device)
            model.device = dev
            model.policy = HipActorCriticPolicy(data["obs_dim"], data["act_dim"], data["continuous"], data["hidden"], dev, seed=0, mfma_dtype=mfma)
            model.policy.load_state_dict(sd)
        return model

