#!/usr/bin/env python3
"""Wall time of PPO.collect_rollouts (rollout + GAE) for a task / env count / policy shape.
usage: time_rollout.py task n_envs n_steps hidden dtype [task n_envs ...]   (TMA_NO_WIDE_FUSED=1: the per-step composition)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from three_mlagents_amd.harness import make_vector_env
from three_mlagents_amd.ppo import PPO

args = sys.argv[1:] or ["basic", "8", "1024", "256", "f32"]
for i in range(0, len(args), 5):
    task, N, T, H, dt = args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3]), args[i + 4]
    env = make_vector_env(task, n_envs=N, seed=1)
    m = PPO("MlpPolicy", env, n_steps=T, batch_size=max(256, N * T // 32), n_epochs=1, seed=1, policy_kwargs={"net_arch": [H, H], "mfma_dtype": dt})
    for _ in range(2):
        m.collect_rollouts()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        m.collect_rollouts()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{task} N={N} T={T} H={H} {dt}: rollout {med * 1e3:.2f} ms = {med / T * 1e6:.2f} us per vector step, {N * T / med / 1e6:.2f} M env-steps/s", flush=True)
    env.close()
