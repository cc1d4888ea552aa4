#!/usr/bin/env python3
"""Per-phase cycle breakdown of the H = 64 gradient kernel's tile chain (diagnostic build libtma_hip_ticks.so, wave 0 of block pair 0).
Run: make -C three-mlagents_amd/csrc libtma_hip_ticks.so && TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_ticks.so python tools/h64_ticks.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
env = make_vector_env("gridworld", n_envs=4096, seed=1)
m = PPO("MlpPolicy", env, n_steps=max(1, B // 4096), batch_size=B, n_epochs=1, seed=1, policy_kwargs={"net_arch": [64, 64]})
m.collect_rollouts()
mb = _lib.Minibatch(None, 1, 0, 0, B)
L = _lib.lib()
def grad():
    _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(m.policy.params), C.byref(m.policy.dims), C.byref(m._rollout_view), C.byref(mb), C.byref(m._hp),
                                        _lib.ptr(m.grad), _lib.ptr(m.workspace), m._stream()))
for _ in range(3):
    grad()
out = (C.c_ulonglong * 32)()
L.tma_debug_h64_ticks.argtypes = [C.c_void_p, C.c_int]
L.tma_debug_h64_ticks(None, 1)
reps = 10
for _ in range(reps):
    grad()
L.tma_debug_h64_ticks(out, 0)
names = ["L1+tanh1", "st h1+L2", "tanh2+st", "head", "loss", "st dz3+dW3", "dh2+dz2+st", "dW2", "dh1+dz1+st", "dW1", "", "", "", "", "", "loop top"]
tiles = max(1, reps * (B // 16) // (min(128, (B // 16 + 7) // 8) * 8))
for net, o in (("pi", 0), ("vf", 16)):
    v = [out[o + i] / tiles for i in range(16)]
    print(net, "cycles per tile:", {n: round(x) for n, x in zip(names, v) if n}, "sum", round(sum(v[:10]) + v[15]))
    loop_c, loop_rt, pro, epi = (out[o + i] / reps for i in (10, 11, 12, 13))
    print("    of the epilogue,", round(out[o + 14] / reps), "cycles are the wait for the block's slowest wave")
    print("    per launch: loop", round(loop_c), "cycles =", round(loop_rt / 100, 1), "us (100 MHz counter) -> clock", round(loop_c / max(loop_rt, 1) * 0.1, 2),
          "GHz; prologue", round(pro), "cycles, epilogue", round(epi), "cycles")
