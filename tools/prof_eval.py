#!/usr/bin/env python3
"""What one device-side evaluate_policy call launches (run under rocprofv3 --kernel-trace --stats): GridWorld, 64-env evaluation vector,
100 episodes, MLP 64x64 -- the EvalCallback's call in train_task.  Prints the wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd.evaluation import evaluate_policy
from three_mlagents_amd.harness import make_vector_env
from three_mlagents_amd.ppo import PPO

task = sys.argv[1] if len(sys.argv) > 1 else "gridworld"
env = make_vector_env(task, n_envs=256, seed=1)
model = PPO("MlpPolicy", env, n_steps=64, batch_size=1024, n_epochs=1, seed=1, policy_kwargs={"net_arch": [64, 64]})
ev = make_vector_env(task, n_envs=64, seed=1001)
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r, l = evaluate_policy(model, ev, n_eval_episodes=100, deterministic=True, return_episode_rewards=True)
    torch.cuda.synchronize()
    print(f"evaluate_policy: {1e3 * (time.perf_counter() - t0):.2f} ms, {len(r)} episodes, mean length {sum(l) / len(l):.1f}")
