#!/usr/bin/env python3
"""A/B of the plain VecEnv.step launch (tma::step_kernel<Task, ACT_I32>, one vector step per launch) between two builds of the library
on ONE box: `python tools/step_ab.py --lib A.so --lib B.so [--task gridworld --n-envs 4096]`.  Every library is timed in a fresh child
process (raw ctypes on the env entry points only, whose ABI has not changed since round 1), alternating, `--rounds` times; prints one JSON
line per library with the median microseconds per launch of native back-to-back bursts (tma_env_step_repeat, HIP events on the launch
stream) -- the figure bench.py's roofline_step_kernel reports."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys


def child(lib_path, task, n, depth, log_cap):
    import torch

    L = C.CDLL(lib_path)
    vp, i32, i64, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32
    L.tma_task_id.argtypes = [C.c_char_p, C.POINTER(i32)]
    L.tma_env_create.argtypes = [i32, i64, i32, u32, u32, i32, C.POINTER(vp)]
    L.tma_env_reset.argtypes = [vp, vp, vp]
    L.tma_env_step_repeat.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.tma_env_refill.argtypes = [vp, vp]
    L.tma_env_episode_log.argtypes = [vp, i64]
    L.tma_task_obs_dim.argtypes = [i32]
    L.tma_last_error.restype = C.c_char_p
    tid = i32(0)
    assert L.tma_task_id(task.encode(), C.byref(tid)) == 0
    D = L.tma_task_obs_dim(tid.value)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    h = vp()
    assert L.tma_env_create(tid.value, n, 0, 1, 0, depth, C.byref(h)) == 0, L.tma_last_error()
    obs = torch.empty((n, D), device=dev)
    assert L.tma_env_reset(h, C.c_void_p(obs.data_ptr()), None) == 0, L.tma_last_error()
    if log_cap:
        assert L.tma_env_episode_log(h, log_cap) == 0
    acts = torch.randint(0, 5, (n,), dtype=torch.int32, device=dev)
    rew, term, trunc, tobs = torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), torch.empty((n, D), device=dev)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    reps = min(depth, 64) - 1
    out = []
    for it in range(24):
        assert L.tma_env_refill(h, s) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.tma_env_step_repeat(h, p(acts), 0, reps, p(obs), p(rew), p(term), p(trunc), p(tobs), s) == 0, L.tma_last_error()
        e1.record()
        torch.cuda.synchronize()
        if it >= 4:
            out.append(e0.elapsed_time(e1) * 1e3 / reps)
    out.sort()
    print(json.dumps({"lib": lib_path, "task": task, "n_envs": n, "episode_log": bool(log_cap), "median_us": out[len(out) // 2], "min_us": out[0], "max_us": out[-1]}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--task", default="gridworld")
    ap.add_argument("--n-envs", type=int, default=4096)
    ap.add_argument("--depth", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--log-cap", type=int, default=0)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.task, a.n_envs, a.depth, a.log_cap)
        sys.exit(0)
    for _ in range(a.rounds):
        for lib in a.lib:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib), "--task", a.task, "--n-envs", str(a.n_envs),
                             "--depth", str(a.depth), "--log-cap", str(a.log_cap)])
