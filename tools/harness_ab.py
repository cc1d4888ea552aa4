#!/usr/bin/env python3
"""A/B of harness.train_task at the headline shape (GridWorld, 4096 envs, MLP 64x64, batch 131072): wall time of N PPO iterations through
train_task with the deferred evaluation (default), with TMA_SYNC_EVAL=1, without any evaluation, and through PPO directly; four interleaved runs
per mode, minimum and median.  Usage: python tools/harness_ab.py [n_iterations]"""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import torch
from three_mlagents_amd import harness, ppo, callbacks
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
total = iters * 4096 * 1024
kw = {"batch_size": 131072, "policy_kwargs": {"net_arch": [64, 64]}}
os.chdir(tempfile.mkdtemp(prefix="tma_ab_"))
orig_tick = callbacks.EvalCallback._tick
def run(mode, tag):
    os.environ.pop("TMA_SYNC_EVAL", None)
    callbacks.EvalCallback._tick = orig_tick
    if mode == "sync": os.environ["TMA_SYNC_EVAL"] = "1"
    if mode == "noeval": callbacks.EvalCallback._tick = lambda self: None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "direct":
        env = harness.make_vector_env("gridworld", n_envs=4096, seed=0)
        m = ppo.PPO("MlpPolicy", env, n_steps=1024, batch_size=131072, n_epochs=10, seed=0, policy_kwargs={"net_arch": [64, 64]})
        m.learn(total); torch.cuda.synchronize(); env.close()
    else:
        cfg = harness.TrainConfig("gridworld", total_timesteps=total, n_envs=4096, run_name=tag, verbose=0)
        harness.train_task(cfg, model_kwargs=kw)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
run("deferred", "warm")
res = {}
for rep in range(4):
    for mode in ("deferred", "sync", "noeval", "direct"):
        res.setdefault(mode, []).append(run(mode, f"{mode}{rep}"))
for mode, v in res.items():
    print(mode, " ".join(f"{x:.1f}" for x in v), "min %.1f" % min(v), "median %.1f" % sorted(v)[len(v)//2])
