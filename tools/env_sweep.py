#!/usr/bin/env python3
"""Env-only throughput sweep of the step kernel over env counts (HIP events on the launch stream)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from three_mlagents_amd.vec_env import HipEnvEngine

B_STEP = {"basic": 110, "gridworld": 90, "ball3d": 114, "push": 74, "crawler": 80 + 2 * 69 * 4 + 688 + 6,
          # same formula (act + 2 * state + 4 * D + 6) for the 8f tasks: state words SW of csrc/tma_tasks.h
          "walljump": 4 + 2 * 4 + 16 + 6, "bicycle": 4 + 2 * 76 + 28 + 6, "brickbreak": 4 + 2 * 52 + 180 + 6, "glider": 4 + 2 * 104 + 64 + 6}


def run(task, n, depth, iters, per_launch):
    eng = HipEnvEngine(task, n, seed=1, ring_depth=depth)
    eng.reset()
    outs = eng._out(per_launch)
    outs = {k: outs[k] for k in ("obs", "rew", "term", "trunc")}
    t0 = 0
    for _ in range(3):
        for _ in range(depth // per_launch):
            eng.step(None, n_steps=per_launch, tape_seed=1, tape_t0=t0, outputs=outs, want_terminal_obs=False, want_episode=False)
            t0 += per_launch
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for _ in range(depth // per_launch):
            eng.step(None, n_steps=per_launch, tape_seed=1, tape_t0=t0, outputs=outs, want_terminal_obs=False, want_episode=False)
            t0 += per_launch
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    steps = iters * depth
    sps = n * steps / (ms * 1e-3)
    eng.close()
    rec = dict(task=task, n_envs=n, ring_depth=depth, steps_per_launch=per_launch, vector_steps=steps, ms=ms, env_steps_per_s=sps)
    if per_launch == 1:  # state in and out of HBM every step: the SURVEY 8d bytes are what the launch moves
        rec.update(alg_GBps=sps * B_STEP[task] / 1e9, frac_of_8TBps=sps * B_STEP[task] / 8e12)
    else:  # the state stays in registers between the steps of a launch: only outputs (obs + reward + 2 flags) leave per step; NOT a roofline figure
        out_bytes = {"basic": 84 + 6, "gridworld": 16 + 6, "ball3d": 24 + 6, "push": 16 + 6, "crawler": 688 + 6, "walljump": 16 + 6, "bicycle": 28 + 6,
                     "brickbreak": 180 + 6, "glider": 64 + 6}[task]
        rec.update(output_bytes_per_env_step=out_bytes, output_GBps=sps * out_bytes / 1e9,
                   note="multi-step launch: state is register-resident, SURVEY-formula bytes do not apply; throughput figure only")
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", default="gridworld,push,ball3d,basic")
    ap.add_argument("--sizes", default="4096,65536,1048576,4194304")
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--per-launch", default="1,32")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    for task in a.tasks.split(","):
        for n in [int(x) for x in a.sizes.split(",")]:
            for pl in [int(x) for x in a.per_launch.split(",")]:
                print(json.dumps(run(task, n, a.depth, a.iters, pl)), flush=True)
