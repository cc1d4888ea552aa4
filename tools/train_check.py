#!/usr/bin/env python3
"""End-to-end sanity: PPO on GridWorld / Push / Ball3D / WallJump with thousands of envs improves the episode return.
`train_check.py 8f` runs the three float64 tasks of SURVEY 8f (Bicycle, BrickBreak, Glider) instead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env
from three_mlagents_amd.evaluation import evaluate_policy

RUNS = (("bicycle", 64, 12, "f32"), ("brickbreak", 64, 12, "f32"), ("glider", 64, 12, "f32")) if sys.argv[1:] == ["8f"] else None
for task, H, iters, dt in RUNS or (("gridworld", 64, 12, "f32"), ("push", 64, 12, "f32"), ("ball3d", 64, 12, "f32"), ("walljump", 64, 12, "f32"),
                           ("ball3d", 256, 12, "bf16"), ("push", 256, 12, "bf16"), ("gridworld", 256, 6, "f32")):
    env = make_vector_env(task, n_envs=4096, seed=1)
    model = PPO("MlpPolicy", env, n_steps=256, batch_size=32768, n_epochs=4, ent_coef=0.01, seed=1, policy_kwargs={"net_arch": [H, H], "mfma_dtype": dt})
    ev = make_vector_env(task, n_envs=256, seed=10_001)
    r0, _ = evaluate_policy(model, ev, n_eval_episodes=512, deterministic=False)
    t0 = time.time()
    hist = []
    for it in range(iters):
        model.learn(4096 * 256, reset_num_timesteps=False)
        hist.append(round(model.logger_values["rollout/ep_rew_mean"], 3))
    torch.cuda.synchronize()
    r1, _ = evaluate_policy(model, ev, n_eval_episodes=512, deterministic=False)
    print(f"{task} {H}x{H} {dt}: eval return {r0:.3f} -> {r1:.3f} after {iters * 4096 * 256 / 1e6:.1f}M steps in {time.time() - t0:.2f}s; rollout ep_rew_mean per iteration {hist}", flush=True)
    env.close(); ev.close()
