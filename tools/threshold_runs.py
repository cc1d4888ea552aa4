#!/usr/bin/env python3
"""The reference's declared reward thresholds (backend/mlagents/registry.py:64,80,96,112,128: basic 0.85, gridworld 0.75, ball3d 150, push 0.65,
walljump 0.7) reached through harness.train_task, and how long that takes.

    python tools/threshold_runs.py [--tasks basic,gridworld,...] [--schedules literal,scaled] [--out gpurun_out/thresholds.json]

`literal`: the reference's own schedule -- its n_envs (registry.py `n_envs`: 1 or 8), its total_timesteps, PPO defaults of training.py:361-391
(n_steps 1024 / 2048, batch_size 256, 10 epochs, MLP 256x256 f32).  `scaled`: 4096 envs, batch 256 * 4096 // 8, same n_steps / epochs / net.
Per run: final deterministic evaluation (the reference's eval_episodes), wall seconds of train_task, and from evaluations.npz + progress.csv
the first evaluation at or above the threshold (timesteps, and seconds on the device timeline = timesteps / fps of that iteration)."""
import argparse
import csv
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

SCALED_ITERATIONS = {"basic": 6, "gridworld": 6, "ball3d": 12, "push": 8, "walljump": 8}


def run(task_id, schedule, seed=1, iterations=None, model_kwargs=None, timesteps=None):
    import torch

    from three_mlagents_amd import harness, tasks

    task = tasks.resolve(task_id)
    n_envs = task.n_envs if schedule == "literal" else 4096
    hp = harness.ppo_defaults(task, n_envs)
    per_iter = n_envs * hp["n_steps"]
    total = timesteps or (task.total_timesteps if schedule == "literal" else (iterations or SCALED_ITERATIONS.get(task_id, 8)) * per_iter)
    tmp = tempfile.mkdtemp(prefix="tma_thr_")
    cwd = os.getcwd()
    try:
        os.chdir(tmp)
        cfg = harness.TrainConfig(task_id, total_timesteps=total, n_envs=n_envs, seed=seed, run_name="thr", verbose=0,
                                  eval_freq=10_000 if schedule == "literal" else per_iter)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = harness.train_task(cfg, model_kwargs=model_kwargs)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ev = np.load(os.path.join(res.run_dir, "eval", "evaluations.npz"))
        ts, means = ev["timesteps"], ev["results"].mean(axis=1)
        fps_at = {}
        prog = os.path.join(res.run_dir, "tb", "progress.csv")
        if os.path.exists(prog):
            with open(prog) as f:
                for row in csv.DictReader(f):
                    try:
                        fps_at[int(float(row["time/total_timesteps"]))] = float(row["time/fps"])
                    except (KeyError, ValueError):
                        pass
        hit = next((i for i, m in enumerate(means) if task.reward_threshold is not None and m >= task.reward_threshold), None)
        first = None
        if hit is not None:
            t_hit = int(ts[hit])
            it_ts = min((k for k in fps_at if k >= t_hit), default=max(fps_at, default=None))  # the iteration that contains the evaluation
            first = {"timesteps": t_hit, "eval_mean": float(means[hit]),
                     "device_seconds": (t_hit / fps_at[it_ts]) if it_ts and fps_at.get(it_ts) else None}
        with open(res.metadata_path) as f:
            sched = json.load(f)["schedule"]
        return {"task": task_id, "schedule": schedule, "threshold": task.reward_threshold, "n_envs": n_envs, "total_timesteps": int(total),
                "ppo": {k: sched[k] for k in ("batch_size", "n_steps", "n_epochs", "minibatches_per_epoch")}, "net": "256x256 f32" if not model_kwargs else str(model_kwargs.get("policy_kwargs")),
                "final_eval_mean": res.mean_reward, "final_eval_std": res.std_reward, "eval_episodes": res.eval_episodes,
                "reached": bool(task.reward_threshold is not None and res.mean_reward >= task.reward_threshold), "first_eval_at_threshold": first,
                "train_task_wall_seconds": wall, "env_steps_per_sec_wall": total / wall, "eval_curve": [[int(a), round(float(b), 4)] for a, b in zip(ts, means)][:: max(1, len(ts) // 24)]}
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", default="basic,gridworld,ball3d,push,walljump")
    ap.add_argument("--schedules", default="literal,scaled")
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seeds", default=None, help="comma-separated seeds: every (task, schedule) once per seed (overrides --seed)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = []
    seeds = [int(x) for x in a.seeds.split(",")] if a.seeds else [a.seed]
    for t in a.tasks.split(","):
        for s in a.schedules.split(","):
            for seed in seeds:
                try:
                    r = run(t, s, seed=seed, iterations=a.iterations)
                except Exception as exc:  # noqa: BLE001
                    r = {"task": t, "schedule": s, "error": repr(exc)}
                r["seed"] = seed
                rows.append(r)
                print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)
