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


class NativeComm:
    """RCCL communicator owned by libtma_hip.so (include/tma.h `tma_comm_*`): the gradient all-reduce of a data-parallel epoch is issued by
    the native epoch loop itself, on the compute stream, with no Python / torch.distributed call per minibatch.  The 128-byte unique id is
    drawn on rank 0 and broadcast over the torch.distributed group that already exists (any backend); a world of one needs no group."""

    def __init__(self, device: torch.device, rccl: bool = True):
        import ctypes as C

        import torch.distributed as td

        from . import _lib

        L = _lib.lib()
        self.world, self.rank, self.device = world_size(), rank(), torch.device(device)
        self.has_rccl, self.p2p_enabled, self.p2p_note = bool(rccl), False, ""
        if not rccl:  # a communicator with the peer exchange only (several ranks on ONE GPU: the one-GPU tests; RCCL wants a device per rank)
            handle = C.c_void_p()
            _lib.check(L.tma_comm_create_p2p(self.world, self.rank, self.device.index if self.device.index is not None else -1, C.byref(handle)))
            self._h, self._L = handle, L
            self.callback = C.cast(L.tma_comm_allreduce_cb, _lib.AllReduceFn)
            return
        if not L.tma_comm_available():
            raise RuntimeError("librccl.so.1 could not be bound by libtma_hip.so")
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            _lib.check(L.tma_comm_unique_id(_lib.ptr(ident.numpy())))
        if self.world > 1:
            on_gpu = "nccl" in str(td.get_backend())  # (also the composite "cpu:gloo,cuda:nccl")
            t = ident.to(self.device) if on_gpu else ident
            td.broadcast(t, src=0)
            ident = t.cpu()
        handle = C.c_void_p()
        _lib.check(L.tma_comm_create(_lib.ptr(ident.contiguous().numpy()), self.world, self.rank,
                                     self.device.index if self.device.index is not None else -1, C.byref(handle)))
        self._h, self._L = handle, L
        self.callback = C.cast(L.tma_comm_allreduce_cb, _lib.AllReduceFn)  # the tma_allreduce_fn the epoch loop calls (ctx = the communicator)

    @property
    def ctx(self):
        return self._h

    def bind_stream(self, stream_ptr) -> None:
        from . import _lib

        _lib.check(self._L.tma_comm_bind_stream(self._h, stream_ptr))

    def all_reduce_(self, t: torch.Tensor, stream_ptr) -> torch.Tensor:
        """In-place SUM of a contiguous float32 / float64 device tensor, enqueued on `stream_ptr`."""
        from . import _lib

        code = {torch.float32: 0, torch.float64: 1}[t.dtype]
        _lib.check(self._L.tma_comm_allreduce(self._h, _lib.ptr(t), t.numel(), code, stream_ptr))
        return t

    # ---- peer exchange (include/tma.h tma_comm_p2p_*): all-reduces of up to `max_words` 32-bit words as direct stores into the peers' inboxes ----
    def p2p_prepare(self, max_words: int) -> bytes:
        """Allocate this rank's inbox and return its 128-byte ticket (IPC handle + PCI bus id of the device)."""
        import ctypes as C

        from . import _lib

        buf = (C.c_ubyte * 128)()
        _lib.check(self._L.tma_comm_p2p_prepare(self._h, int(max_words), buf))
        return bytes(buf)

    def p2p_attach(self, handles: list[bytes]) -> None:
        """Map the peers' inboxes (`handles`: every rank's p2p_prepare() result, in rank order)."""
        import ctypes as C

        from . import _lib

        blob = b"".join(handles)
        assert len(blob) == 128 * self.world
        _lib.check(self._L.tma_comm_p2p_attach(self._h, (C.c_ubyte * len(blob)).from_buffer_copy(blob)))

    def p2p_enable(self, on: bool = True) -> None:
        from . import _lib

        _lib.check(self._L.tma_comm_p2p_enable(self._h, 1 if on else 0))
        self.p2p_enabled = bool(on)

    def p2p_set_timeout(self, seconds: float) -> None:
        from . import _lib

        _lib.check(self._L.tma_comm_p2p_set_timeout(self._h, float(seconds)))

    def p2p_status(self) -> dict:
        import ctypes as C

        from . import _lib

        en, calls, bad, words = C.c_int(0), C.c_int64(0), C.c_int(0), C.c_int64(0)
        _lib.check(self._L.tma_comm_p2p_status(self._h, C.byref(en), C.byref(calls), C.byref(bad), C.byref(words)))
        return {"enabled": bool(en.value), "calls": int(calls.value), "timed_out": bool(bad.value), "slot_words": int(words.value)}

    def p2p_setup(self, max_words: int) -> None:
        """prepare + gather the handles over torch.distributed (the group that already exists, any backend) + attach.  Collective over the ranks;
        raises on the rank where a step fails (the caller agrees on the outcome across ranks before it enables the exchange)."""
        import torch.distributed as td

        mine = self.p2p_prepare(max_words)
        if self.world == 1:
            self.p2p_attach([mine])
            return
        on_gpu = "nccl" in str(td.get_backend())
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        t = t.to(self.device) if on_gpu else t
        out = [torch.zeros_like(t) for _ in range(self.world)]
        td.all_gather(out, t)
        self.p2p_attach([bytes(o.cpu().numpy().tobytes()) for o in out])

    def timing(self, samples: int) -> None:
        from . import _lib

        _lib.check(self._L.tma_comm_timing(self._h, int(samples)))

    def pop_timing(self, capacity: int = 4096) -> tuple[list[float], int]:
        import ctypes as C

        from . import _lib

        buf, n, calls = (C.c_float * capacity)(), C.c_int(0), C.c_int64(0)
        _lib.check(self._L.tma_comm_pop_timing(self._h, buf, capacity, C.byref(n), C.byref(calls)))
        return [float(buf[i]) for i in range(n.value)], int(calls.value)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.tma_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
