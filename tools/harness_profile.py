#!/usr/bin/env python3
"""Where the wall time of harness.train_task goes at the headline shape (two PPO iterations, 4096 envs): cProfile of the second of two
calls (the first loads every kernel module), printed by cumulative time.  Usage: python tools/harness_profile.py [n_iterations]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from three_mlagents_amd import harness

    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    total = iters * 4096 * 1024
    kw = {"batch_size": 131072, "policy_kwargs": {"net_arch": [64, 64]}}
    os.chdir(tempfile.mkdtemp(prefix="tma_prof_"))
    for rep in range(2):
        cfg = harness.TrainConfig("gridworld", total_timesteps=total, n_envs=4096, run_name=f"p{rep}", verbose=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if rep == 1:
            pr = cProfile.Profile()
            pr.enable()
        harness.train_task(cfg, model_kwargs=kw)
        torch.cuda.synchronize()
        if rep == 1:
            pr.disable()
        print(f"pass {rep}: {time.perf_counter() - t0:.3f} s for {iters} iterations")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
