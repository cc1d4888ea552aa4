#!/usr/bin/env python3
"""Per-phase cycle breakdown of the three-term split gradient kernel (diagnostic build libtma_hip_s3ticks.so; wave 0 of block 0 of each net).
Stamps charge a barrier's wait to the phase in front of it.  Run: make -C three-mlagents_amd/csrc libtma_hip_s3ticks.so &&
TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_s3ticks.so python tools/s3_ticks.py [task]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task = sys.argv[1] if len(sys.argv) > 1 else "gridworld"
B = 131072
env = make_vector_env(task, n_envs=4096, seed=1)
m = PPO("MlpPolicy", env, n_steps=B // 4096, batch_size=B, n_epochs=1, seed=1, policy_kwargs={"net_arch": [256, 256], "mfma_dtype": "bf16x3"})
m.collect_rollouts()
mb = _lib.Minibatch(None, 1, 0, 0, B)
L = _lib.lib()
def grad():
    _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(m.policy.params), C.byref(m.policy.dims), C.byref(m._rollout_view), C.byref(mb), C.byref(m._hp),
                                        _lib.ptr(m.grad), _lib.ptr(m.workspace), m._stream()))
for _ in range(3):
    grad()
torch.cuda.synchronize()
out = (C.c_ulonglong * 32)()
L.tma_debug_s3_ticks.argtypes = [C.c_void_p, C.c_int]
L.tma_debug_s3_ticks(None, 1)
reps = 10
for _ in range(reps):
    grad()
torch.cuda.synchronize()
L.tma_debug_s3_ticks(out, 0)
names = ["loop top", "P0 commit + barrier", "P1 layer 1 + barrier", "P2 GEMM (48 k-steps)", "P2 epilogue (tanh, split)", "head partial + barrier", "P3 loss + barrier", "P3c dz3 planes + barrier",
         "P4 dW3 + dz2 + barrier", "P5 dW2", "dh1 GEMM", "barrier after dh1", "P6 dz1 + dW1", "group-end barrier"]
for net, o in (("pi", 0), ("vf", 16)):
    v = [out[o + i] / reps for i in range(14)]
    tot = sum(v)
    print(net, "cycles per launch (block 0, 32 groups):", {n: round(x / 32) for n, x in zip(names, v)}, "per group; sum", round(tot / 32))
    print("    share:", {n: f"{100 * x / tot:.1f}%" for n, x in zip(names, v)})
