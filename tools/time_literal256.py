#!/usr/bin/env python3
"""The reference's literal batch_size = 256 on a 256 x 256 policy: one epoch of per-minibatch optimizer steps (bench.py's literal_batch_256 leg
on its own).  usage: time_literal256.py [task n_envs n_steps hidden [n_epochs]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task = sys.argv[1] if len(sys.argv) > 1 else "gridworld"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
H = int(sys.argv[4]) if len(sys.argv) > 4 else 256
E = int(sys.argv[5]) if len(sys.argv) > 5 else 1
env = make_vector_env(task, n_envs=N, seed=1)
m = PPO("MlpPolicy", env, n_steps=T, batch_size=256, n_epochs=E, seed=1, policy_kwargs={"net_arch": [H, H]})
m.collect_rollouts(); m.train(); torch.cuda.synchronize()
best = None
for _ in range(3):
    m.collect_rollouts(); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
n_mb = N * T // 256 * E
import hashlib
digest = hashlib.sha256(b"".join(v.detach().cpu().contiguous().numpy().tobytes() for _, v in sorted(m.policy.state_dict().items()))).hexdigest()[:16]
print(f"{task} {N}x{T} H={H}: {n_mb} optimizer steps per {'train() of ' + str(E) + ' epochs' if E > 1 else 'epoch'}, {best / n_mb * 1e6:.2f} us per step = {n_mb / best:.0f} steps/s; approx_kl {m.pop_train_stats()['train/approx_kl']:.5f}; parameters sha256 {digest}")
