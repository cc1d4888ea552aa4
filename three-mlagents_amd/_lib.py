"""ctypes binding of csrc/libtma_hip.so (the C ABI declared in include/tma.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object is missing this
module raises ImportError telling the user to build it (`python -c "import __graft_entry__ as g; g.build()"`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TMA_LIB_PATH") or os.path.join(_HERE, "csrc", "libtma_hip.so")  # (TMA_LIB_PATH: diagnostic builds of the same ABI)

TMA_OK, TMA_ERR_INVALID, TMA_ERR_UNKNOWN_TASK, TMA_ERR_HIP = 0, 1, 2, 3
ACT_I32, ACT_I64, ACT_F32 = 0, 1, 2
EP_STRIDE = 1 << 20

_vp, _i32, _i64, _u32, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_double



class PolicyDims(C.Structure):
    _fields_ = [("obs_dim", _i32), ("hidden", _i32), ("act_dim", _i32), ("continuous", _i32), ("mfma_dtype", _i32), ("device", _i32)]


class Rollout(C.Structure):
    _fields_ = [("obs", _vp), ("actions", _vp), ("log_probs", _vp), ("advantages", _vp), ("returns", _vp), ("T", _i32), ("N", _i64),
                ("packed", _vp)]  # (optional sample records, tma_ppo_pack_samples; None = absent)


class Minibatch(C.Structure):
    _fields_ = [("indices", _vp), ("perm_seed", _u32), ("perm_epoch", _u32), ("start", _i64), ("count", _i64), ("prepared_batch", _i64), ("stats_count", _i64)]


class PPOHParams(C.Structure):
    _fields_ = [("clip_range", _f64), ("ent_coef", _f64), ("vf_coef", _f64), ("normalize_advantage", _i32)]


class RolloutBuffers(C.Structure):
    _fields_ = [("obs", _vp), ("actions", _vp), ("rewards", _vp), ("values", _vp), ("log_probs", _vp), ("terminated", _vp),
                ("truncated", _vp), ("terminal_obs", _vp), ("last_values", _vp), ("N", _i64), ("terminal_obs_slots", _i32)]


_pd = C.POINTER(PolicyDims)

# name -> (restype, argtypes); every symbol include/tma.h declares
SIGNATURES = {
    "tma_version": (_i32, []),
    "tma_last_error": (C.c_char_p, []),
    "tma_task_id": (_i32, [C.c_char_p, C.POINTER(_i32)]),
    "tma_task_obs_dim": (_i32, [_i32]),
    "tma_task_num_actions": (_i32, [_i32]),
    "tma_task_act_dim": (_i32, [_i32]),
    "tma_task_state_dim": (_i32, [_i32]),
    "tma_task_max_episode_steps": (_i32, [_i32]),
    "tma_env_create": (_i32, [_i32, _i64, _i32, _u32, _u32, _i32, C.POINTER(_vp)]),
    "tma_env_destroy": (_i32, [_vp]),
    "tma_env_seed": (_i32, [_vp, _u32]),
    "tma_env_set_reward64": (_i32, [_vp, _vp, _i64]),
    "tma_env_reset": (_i32, [_vp, _vp, _vp]),
    "tma_env_step": (_i32, [_vp, _vp, _i32, _u32, _u32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tma_env_step_repeat": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tma_env_steps_until_refill": (_i32, [_vp, C.POINTER(_i32)]),
    "tma_env_refill": (_i32, [_vp, _vp]),
    "tma_env_set_option": (_i32, [_vp, C.c_char_p, _i64]),
    "tma_env_get_state": (_i32, [_vp, _vp, _vp]),
    "tma_env_set_state": (_i32, [_vp, _vp, _vp]),
    "tma_env_episode_index": (_i32, [_vp, _vp, _vp]),
    "tma_env_episode_log": (_i32, [_vp, _i64]),
    "tma_monitor_append_rows": (_i32, [C.c_char_p, _vp, _vp, _vp, _i64]),
    "tma_env_pop_episode_log": (_i32, [_vp, _vp, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "tma_env_pop_episode_stats": (_i32, [_vp, C.POINTER(_f64), _vp]),
    "tma_env_clear_episode_log": (_i32, [_vp, _vp]),
    "tma_env_detach_episode_log": (_i32, [_vp]),
    "tma_env_pop_detached_episode_log": (_i32, [_vp, _vp, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_f64), _vp]),
    "tma_gae": (_i32, [_vp, _vp, _vp, _vp, _vp, _f64, _f64, _i32, _i64, _vp, _vp, _vp]),
    "tma_gae_flags": (_i32, [_vp, _vp, _vp, _vp, _vp, _f64, _f64, _i32, _i64, _vp, _vp, _vp]),
    "tma_policy_param_count": (_i32, [_pd, C.POINTER(_i64), C.POINTER(_i64)]),
    "tma_policy_param_offsets": (_i32, [_pd, C.POINTER(_i32)]),
    "tma_policy_sync": (_i32, [_vp, _pd, _vp]),
    "tma_policy_act": (_i32, [_vp, _pd, _vp, _i64, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _vp]),
    "tma_policy_act_bootstrap": (_i32, [_vp, _pd, _vp, _i64, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp]),
    "tma_policy_values": (_i32, [_vp, _pd, _vp, _i64, _vp, _vp]),
    "tma_policy_bootstrap": (_i32, [_vp, _pd, _vp, _vp, _i64, _f64, _vp, _vp]),
    "tma_ppo_workspace_bytes": (_i64, [_pd]),
    "tma_ppo_packed_floats": (_i64, [_pd, _i32, _i64]),
    "tma_ppo_pack_samples": (_i32, [C.POINTER(Rollout), _pd, _vp, _vp]),
    "tma_ppo_minibatch_grad": (_i32, [_vp, _pd, C.POINTER(Rollout), C.POINTER(Minibatch), C.POINTER(PPOHParams), _vp, _vp, _vp]),
    "tma_debug_time_grad_kernel": (_i32, [_i32]),
    "tma_debug_poison_lds": (_i32, [_u32, _vp]),
    "tma_debug_last_grad_kernel_us": (_i32, [C.POINTER(C.c_float)]),
    "tma_ppo_epoch_prepare": (_i32, [C.POINTER(Rollout), C.POINTER(Minibatch), _i64, _pd, _vp, _vp]),
    "tma_ppo_epoch_adv_sums": (_i32, [_vp, _pd, _i64, _i64, _vp, _i32, _vp]),
    "tma_ppo_adam_step": (_i32, [_vp, _vp, _vp, _vp, _pd, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp]),
    "tma_ppo_adam_step_local": (_i32, [_vp, _vp, _vp, _vp, _pd, _i64, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _i64]),
    "tma_ppo_permutation": (_i32, [_u32, _u32, _i64, _vp]),
    "tma_ppo_persist_fallbacks": (_i32, [_vp, C.POINTER(_i64), _vp]),
    "tma_ppo_train_epoch_local": (_i32, [_vp, _pd, C.POINTER(Rollout), _u32, _u32, _i64, C.POINTER(PPOHParams), _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64,
                                         _f64, _vp, _vp]),
    "tma_ppo_train_epochs_local": (_i32, [_vp, _pd, C.POINTER(Rollout), _u32, _u32, _i32, _i64, C.POINTER(PPOHParams), _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64,
                                          _f64, _vp, _vp]),
    "tma_ppo_train_epoch_dp": (_i32, [_vp, _pd, C.POINTER(Rollout), _u32, _u32, _i64, _i64, _i32, C.POINTER(PPOHParams), _vp, _vp, _vp, _i64, _f64, _f64, _f64,
                                      _f64, _f64, _f64, None, _vp, _vp, _vp]),  # (None: the AllReduceFn slot, filled in below)
    "tma_ppo_pop_stats": (_i32, [_vp, C.POINTER(_f64), _vp]),
    "tma_ppo_stats_staging_bytes": (_i64, []),
    "tma_ppo_stats_enqueue": (_i32, [_vp, _vp, _vp]),
    "tma_ppo_stats_fold": (_i32, [_vp, C.POINTER(_f64)]),
    "tma_comm_available": (_i32, []),
    "tma_comm_unique_id": (_i32, [_vp]),
    "tma_comm_create": (_i32, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "tma_comm_destroy": (_i32, [_vp]),
    "tma_comm_bind_stream": (_i32, [_vp, _vp]),
    "tma_comm_allreduce": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "tma_comm_allreduce_cb": (_i32, [_vp, _vp, _i64]),
    "tma_comm_timing": (_i32, [_vp, _i32]),
    "tma_comm_pop_timing": (_i32, [_vp, _vp, _i32, C.POINTER(_i32), C.POINTER(_i64)]),
    "tma_comm_create_p2p": (_i32, [_i32, _i32, _i32, C.POINTER(_vp)]),
    "tma_comm_p2p_prepare": (_i32, [_vp, _i64, _vp]),
    "tma_comm_p2p_attach": (_i32, [_vp, _vp]),
    "tma_comm_p2p_attach_local": (_i32, [_vp, _vp]),
    "tma_comm_p2p_enable": (_i32, [_vp, _i32]),
    "tma_comm_p2p_set_timeout": (_i32, [_vp, C.c_double]),
    "tma_comm_p2p_status": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64)]),
    "tma_rollout_collect": (_i32, [_vp, _vp, _pd, C.POINTER(RolloutBuffers), _i32, _i32, _i32, _u32, _u32, _u32, _f64, _i32, _i32, _vp]),
}

# tma_allreduce_fn (include/tma.h): int (*)(void *ctx, float *buffer, int64_t count)
AllReduceFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
SIGNATURES["tma_ppo_train_epoch_dp"] = (_i32, [AllReduceFn if a is None else a for a in SIGNATURES["tma_ppo_train_epoch_dp"][1]])

_lib = None


def lib():
    """Load libtma_hip.so once; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: the HIP extension is the only compute path of three-mlagents_amd. "
                "Build it with `make -C three-mlagents_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)."
            )
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().tma_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int) -> None:
    """Map a TMA_* status to the exception type the reference raises for the same condition."""
    if status == TMA_OK:
        return
    msg = last_error()
    if status == TMA_ERR_UNKNOWN_TASK:
        raise KeyError(msg)  # registry.get_task: backend/mlagents/registry.py:359-362
    if status == TMA_ERR_INVALID:
        raise ValueError(msg)  # registry.make_env / train_task: registry.py:368-369, training.py:105-114
    raise RuntimeError(msg)  # HIP / RCCL failures


def task_id(name: str) -> int:
    out = _i32(0)
    check(lib().tma_task_id(name.encode(), C.byref(out)))
    return out.value


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return t.ctypes.data_as(C.c_void_p)


def stream_ptr(device=None):
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
