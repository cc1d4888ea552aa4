#!/usr/bin/env python3
"""Per-phase cycle breakdown of the f32 wide gradient kernel (diagnostic build libtma_hip_wticks.so; wave 0 of block 0 of each net).
The stamps charge a barrier's wait to the phase in front of it and lengthen the launch; use the shares to rank phases.
Run: make -C three-mlagents_amd/csrc libtma_hip_wticks.so && TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_wticks.so python tools/wide_ticks.py [task] [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task = sys.argv[1] if len(sys.argv) > 1 else "gridworld"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
env = make_vector_env(task, n_envs=4096, seed=1)
m = PPO("MlpPolicy", env, n_steps=max(32, B // 4096), batch_size=B, n_epochs=1, seed=1, policy_kwargs={"net_arch": [256, 256]})
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
L.tma_debug_wide_ticks.argtypes = [C.c_void_p, C.c_int]
L.tma_debug_wide_ticks(None, 1)
reps = 20
for _ in range(reps):
    grad()
torch.cuda.synchronize()
L.tma_debug_wide_ticks(out, 0)
names = ["prologue", "P0 gathers", "P1 layer 1", "P2 layer 2", "P3 head + loss", "P4 dW3 + dz2", "P5a dW2", "P5b dh1", "P6 dz1 + dW1", "slab store issue"]
for net, o in (("pi", 0), ("vf", 16)):
    v = [out[o + i] / reps for i in range(10)]
    tot = sum(v)
    print(net, f"B={B} cycles per launch (block 0):", {n: round(x) for n, x in zip(names, v)}, "sum", round(tot))
    print("    share:", {n: f"{100 * x / tot:.1f}%" for n, x in zip(names, v)})
