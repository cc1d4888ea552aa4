"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm) / gloo on CPU.

The reference has no distributed code (SURVEY.md §2 "explicit negatives"); the build shards the independent envs over
ranks and all-reduces ONLY the flat policy gradient once per minibatch.  Env state, observations and
rollout buffers never leave their GPU.
"""
from __future__ import annotations

import os

import torch


def is_initialized() -> bool:
    import torch.distributed as td

    return td.is_available() and td.is_initialized()


def world_size() -> int:
    import torch.distributed as td

    return td.get_world_size() if is_initialized() else 1


def rank() -> int:
    import torch.distributed as td

    return td.get_rank() if is_initialized() else 0


def init_from_env(backend: str | None = None, single_rank_group: bool = False) -> tuple[int, int, int]:
    """Initialise from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  Returns (rank, local_rank, world).
    A world of one needs no process group and gets none, unless `single_rank_group` asks for it (exercises the RCCL
    communicator on a one-GPU box)."""
    import torch.distributed as td

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", str(rk)))
    if (ws > 1 or single_rank_group) and not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:  # (TMA_DIST_BACKEND=gloo: test hook -- several ranks on ONE GPU, which RCCL does not allow)
            backend = os.environ.get("TMA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(lr)
            td.init_process_group(backend, rank=rk, world_size=ws, device_id=torch.device("cuda", lr))
        else:
            td.init_process_group(backend, rank=rk, world_size=ws)
    return rk, lr, ws


def shard_envs(n_envs_per_rank: int, rank_: int | None = None) -> tuple[int, int]:
    """(env_offset, n_local): rank r owns global envs [r * n, (r + 1) * n) -- weak scaling, fixed envs per GPU."""
    r = rank() if rank_ is None else int(rank_)
    return r * int(n_envs_per_rank), int(n_envs_per_rank)


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    import torch.distributed as td

    if is_initialized() and td.get_world_size() > 1:
        td.all_reduce(t)
    return t


def allreduce_max_float(x: float, device=None) -> float:
    import torch.distributed as td

    if not (is_initialized() and td.get_world_size() > 1):
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    import torch.distributed as td

    if is_initialized() and td.get_world_size() > 1:
        td.barrier()
