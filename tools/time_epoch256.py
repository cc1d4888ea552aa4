#!/usr/bin/env python3
"""Time one epoch of the reference's literal batch_size = 256 (GridWorld 64x64) through tma_ppo_train_epoch_local: optimizer steps / s for the
persistent epoch kernel (default) and, with TMA_NO_PERSIST=1, the per-minibatch launches.  TMA_H64P_TICKS=1 prints the phase ticks.
usage: time_epoch256.py [n_envs n_steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
env = make_vector_env("gridworld", n_envs=n_envs, seed=1)
m = PPO("MlpPolicy", env, n_steps=n_steps, batch_size=256, n_epochs=1, seed=1, policy_kwargs={"net_arch": [64, 64]})
m.collect_rollouts()
n_mb = n_envs * n_steps // 256
for mode in ("persist", "launch"):
    if mode == "launch":
        os.environ["TMA_NO_PERSIST"] = "1"
    else:
        os.environ.pop("TMA_NO_PERSIST", None)
    m.train()
    torch.cuda.synchronize()
    if mode == 'persist' and hasattr(_lib.lib(), 'tma_debug_h64p_tile_ticks'):
        _lib.lib().tma_debug_h64p_tile_ticks(None, 1)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        m.train()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = m.pop_train_stats()
    t = min(ts)
    print(f"{mode}: {n_mb} optimizer steps in {t * 1e3:.1f} ms = {t / n_mb * 1e6:.2f} us/step = {n_mb / t / 1e3:.1f} k steps/s   kl {st['train/approx_kl']:.3e}", flush=True)
    if mode == "persist" and os.environ.get("TMA_H64P_TICKS"):
        out = (C.c_ulonglong * 10)()
        _lib.lib().tma_debug_h64p_ticks.argtypes = [C.c_void_p, C.c_void_p]
        _lib.lib().tma_debug_h64p_ticks(_lib.ptr(m.workspace), out)
        names = ["setup", "tile", "blocksum+slab", "syncA", "quarter", "syncB", "loadG", "adam", "pre-tile", "-"]
        print("  ticks/step:", "  ".join(f"{n} {out[i] / n_mb:.0f}" for i, n in enumerate(names) if n != "-"))
        if hasattr(_lib.lib(), "tma_debug_h64p_tile_ticks"):
            t32 = (C.c_ulonglong * 32)()
            _lib.lib().tma_debug_h64p_tile_ticks(t32, 1)
            runs = 3  # timed train() calls since the reset
            ph = ["L1+tanh", "L2 chain", "tanh2+store", "head", "loss", "dh2+dW3", "dz2", "dh1 chain", "dW2+dz1", "dW1"]
            print("  tile ticks/step (policy net):", "  ".join(f"{n} {t32[i] / n_mb / runs:.0f}" for i, n in enumerate(ph)))
