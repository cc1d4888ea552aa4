#!/usr/bin/env python3
"""Per-phase cycle sums of the 256-wide persistent epoch kernel (csrc/tma_h256p.hip), stamped by thread 0 of workgroup (policy net, row group
0, slice 0) with s_memtime (shader-clock cycles; the stamps themselves cost ~0.5 us of a step).  usage: TMA_H256P_TICKS=1 h256p_ticks.py [task n_envs n_steps]"""
import ctypes as C, os, sys
os.environ.setdefault("TMA_H256P_TICKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task = sys.argv[1] if len(sys.argv) > 1 else "gridworld"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
env = make_vector_env(task, n_envs=N, seed=1)
m = PPO("MlpPolicy", env, n_steps=T, batch_size=256, n_epochs=1, seed=1, policy_kwargs={"net_arch": [256, 256]})
m.collect_rollouts(); m.train(); torch.cuda.synchronize()
L = _lib.lib()
L.tma_debug_h256p_ticks.argtypes = [C.c_void_p, C.c_void_p]
out = (C.c_ulonglong * 24)()
_lib.check(L.tma_debug_h256p_ticks(_lib.ptr(m.workspace), out))
n_mb = N * T // 256
# tick index i is stamped at the END of phase i
labels = {0: "prologue + first layer 1", 1: "X1 wait", 2: "gather h1 / W2 quarters", 3: "P2 layer 2", 4: "P3a head partial + X2 arrive/wait", 5: "P3b outputs + loss",
          6: "P4 dW3, dz2", 7: "P5a dW2 + stores", 8: "P5b dh1 share", 9: "store drain + barrier", 15: "dz1 in place (+ barrier)", 16: "P6 dW1 + store issue", 10: "X3 arrive/wait + commit", 11: "reduce + sumsq",
          12: "X4 granule + wait", 13: "Adam", 14: "layer 1 + publish"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 9, 10, 11, 12, 13, 14]
tot = sum(out[i] for i in order[1:])
print(f"{task} {N}x{T}: {n_mb} steps; cycles per step")
for i in order:
    per = out[i] / (1 if i == 0 else n_mb)
    print(f"  [{i:2d}] {labels[i]:38s} {per:9.0f}" + (" (once)" if i == 0 else f"  ({100.0 * out[i] / tot:4.1f} %)"))
print(f"  sum of per-step phases: {tot / n_mb:.0f} cycles")
