"""ctypes binding of oracle/libtma_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (three-mlagents_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("TMA_ORACLE_PATH") or os.path.join(_HERE, "libtma_oracle.so")  # (TMA_ORACLE_PATH: the sanitizer build, `make asan`)

TASK_IDS = {"basic": 0, "gridworld": 1, "ball3d": 2, "push": 3, "crawler": 4, "walljump": 5, "bicycle": 6, "brickbreak": 7, "glider": 8, "ant": 9}
EP_STRIDE = 1 << 20


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "tma_oracle.c")
    if os.environ.get("TMA_ORACLE_PATH"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i32, u32, f32, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_double
        L.orc_vec_create.restype = vp
        L.orc_vec_create.argtypes = [i32, i32, u32, u32]
        L.orc_vec_destroy.argtypes = [vp]
        L.orc_vec_set_threads.argtypes = [vp, i32]
        L.orc_vec_reset.argtypes = [vp, vp]
        L.orc_vec_step.argtypes = [vp] + [vp] * 9
        L.orc_vec_get_state.argtypes = [vp, vp]
        L.orc_vec_set_state.argtypes = [vp, vp]
        L.orc_vec_episode_index.argtypes = [vp, vp]
        L.orc_reset_from_seed.argtypes = [i32, u32, vp, vp]
        L.orc_legacy_step.argtypes = [i32, vp, vp, vp, vp, vp]
        L.orc_gae.argtypes = [vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp]
        L.orc_gae_threads.argtypes = [vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, vp, i32]
        L.orc_mt_raw.argtypes = [u32, vp, i32]
        L.orc_mt_shuffle.argtypes = [u32, vp, i32, vp]
        L.orc_mt_uniform.argtypes = [u32, f64, f64, vp, i32]
        L.orc_action_tape.argtypes = [u32, i32, i32, i32, i32, vp]
        L.orc_mix32.restype = u32
        L.orc_mix32.argtypes = [u32, u32, u32]
        for fn in ("orc_obs_dim", "orc_num_actions", "orc_act_dim", "orc_state_dim", "orc_max_episode_steps"):
            getattr(L, fn).restype = i32
            getattr(L, fn).argtypes = [i32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def task_id(task) -> int:
    return TASK_IDS[task] if isinstance(task, str) else int(task)


def obs_dim(task):
    return lib().orc_obs_dim(task_id(task))


def num_actions(task):
    return lib().orc_num_actions(task_id(task))


def act_dim(task):
    return lib().orc_act_dim(task_id(task))


def state_dim(task):
    return lib().orc_state_dim(task_id(task))


def max_episode_steps(task):
    return lib().orc_max_episode_steps(task_id(task))


def episode_seed(base, i, k):
    return (base + i + k * EP_STRIDE) & 0xFFFFFFFF


def action_tape(tape_seed, n_envs, T, n_actions, env_offset=0):
    out = np.zeros((T, n_envs), np.int32)
    lib().orc_action_tape(tape_seed, n_envs, env_offset, T, n_actions, _p(out))
    return out


def mt_raw(seed, n):
    out = np.zeros(n, np.uint32)
    lib().orc_mt_raw(int(seed) & 0xFFFFFFFF, _p(out), n)
    return out


def mt_shuffle(seed, n, next_max=None):
    perm = np.zeros(n, np.int32)
    nxt = np.array([next_max if next_max is not None else 0], np.int32)
    lib().orc_mt_shuffle(int(seed) & 0xFFFFFFFF, _p(perm), n, _p(nxt) if next_max is not None else None)
    return perm, int(nxt[0])


def mt_uniform(seed, lo, hi, n):
    out = np.zeros(n, np.float64)
    lib().orc_mt_uniform(int(seed) & 0xFFFFFFFF, lo, hi, _p(out), n)
    return out


def reset_from_seed(task, seed):
    t = task_id(task)
    st = np.zeros(state_dim(t), np.float64)
    obs = np.zeros(obs_dim(t), np.float32)
    lib().orc_reset_from_seed(t, int(seed) & 0xFFFFFFFF, _p(st), _p(obs))
    return st, obs


def legacy_step(task, state, action):
    t = task_id(task)
    st = np.array(state, np.float64)
    obs = np.zeros(obs_dim(t), np.float32)
    r = np.zeros(1, np.float64)
    d = np.zeros(1, np.int32)
    act = np.asarray(action, np.float32 if t in (4, 9) else np.int32).reshape(-1).copy()
    lib().orc_legacy_step(t, _p(st), _p(act), _p(obs), _p(r), _p(d))
    return st, obs, float(r[0]), bool(d[0])


class OracleVecEnv:
    """DummyVecEnv + Monitor + adapter semantics on the CPU (C implementation)."""

    def __init__(self, task, n_envs, seed=1, env_offset=0, threads=1):
        self.task = task_id(task)
        self.n = n_envs
        self.D = obs_dim(self.task)
        self.S = state_dim(self.task)
        self.A = act_dim(self.task)
        self._h = lib().orc_vec_create(self.task, n_envs, int(seed) & 0xFFFFFFFF, env_offset)
        lib().orc_vec_set_threads(self._h, threads)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_vec_destroy(self._h)
            self._h = None

    def reset(self):
        obs = np.zeros((self.n, self.D), np.float32)
        lib().orc_vec_reset(self._h, _p(obs))
        return obs

    def step(self, actions):
        if self.task in (4, 9):
            act = np.ascontiguousarray(actions, np.float32).reshape(self.n, self.A)
        else:
            act = np.ascontiguousarray(actions, np.int32).reshape(self.n)
        out = dict(
            obs=np.zeros((self.n, self.D), np.float32),
            rew32=np.zeros(self.n, np.float32),
            rew64=np.zeros(self.n, np.float64),
            term=np.zeros(self.n, np.uint8),
            trunc=np.zeros(self.n, np.uint8),
            term_obs=np.zeros((self.n, self.D), np.float32),
            ep_ret=np.zeros(self.n, np.float64),
            ep_len=np.zeros(self.n, np.int32),
        )
        lib().orc_vec_step(self._h, _p(act), _p(out["obs"]), _p(out["rew32"]), _p(out["rew64"]), _p(out["term"]),
                           _p(out["trunc"]), _p(out["term_obs"]), _p(out["ep_ret"]), _p(out["ep_len"]))
        return out

    def step_fast(self, act, obs, rew32, term, trunc):
        """Timing path: preallocated outputs, no terminal-obs / episode outputs."""
        lib().orc_vec_step(self._h, _p(act), _p(obs), _p(rew32), None, _p(term), _p(trunc), None, None, None)

    def get_state(self):
        st = np.zeros((self.n, self.S), np.float64)
        lib().orc_vec_get_state(self._h, _p(st))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, np.float64).reshape(self.n, self.S)
        lib().orc_vec_set_state(self._h, _p(st))

    def episode_index(self):
        out = np.zeros(self.n, np.uint32)
        lib().orc_vec_episode_index(self._h, _p(out))
        return out


def gae(rewards, values, episode_starts, last_values, dones, gamma=0.99, gae_lambda=0.95, threads=1):
    T, N = rewards.shape
    adv = np.zeros((T, N), np.float32)
    ret = np.zeros((T, N), np.float32)
    r = np.ascontiguousarray(rewards, np.float32)
    v = np.ascontiguousarray(values, np.float32)
    es = np.ascontiguousarray(episode_starts, np.float32)
    lv = np.ascontiguousarray(last_values, np.float32)
    d = np.ascontiguousarray(dones, np.uint8)
    gl = np.float32(float(gamma) * float(gae_lambda))  # python-float product, then weak-cast to f32
    if threads > 1:
        lib().orc_gae_threads(_p(r), _p(v), _p(es), _p(lv), _p(d), np.float32(gamma), gl, T, N, _p(adv), _p(ret), int(threads))
    else:
        lib().orc_gae(_p(r), _p(v), _p(es), _p(lv), _p(d), np.float32(gamma), gl, T, N, _p(adv), _p(ret))
    return adv, ret
