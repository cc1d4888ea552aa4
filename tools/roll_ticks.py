#!/usr/bin/env python3
"""Per-phase cycles of one vector step of the f32 256-wide fused rollout kernel (diagnostic build libtma_hip_rticks.so: thread 0 of block 0;
a stamp waits for the wave's outstanding LDS / scalar operations, so small phases read a little long).
Run: make -C three-mlagents_amd/csrc libtma_hip_rticks.so && TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_rticks.so python tools/roll_ticks.py [task n_envs]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task = sys.argv[1] if len(sys.argv) > 1 else "basic"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = 1024
env = make_vector_env(task, n_envs=N, seed=1)
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = PPO("MlpPolicy", env, n_steps=T, batch_size=256, n_epochs=1, seed=1, policy_kwargs={"net_arch": [H, H]})
m.collect_rollouts(); torch.cuda.synchronize()
L = _lib.lib()
L.tma_debug_roll_ticks.argtypes = [C.c_void_p, C.c_int]
L.tma_debug_roll_ticks(None, 1)
m.collect_rollouts(); torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
L.tma_debug_roll_ticks(out, 0)
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
names = ["loop top", "layer 1 + tanh + barrier", "layer 2 + tanh + barrier", "head chain (+ barrier on the Box tasks)", "softmax / Gaussian sampling", "env step + observation", "last barrier"]
if H == 64:  # rollout_chunk2_h64_kernel: thread 0 = the policy wave
    names = ["loop top + observation read", "forward (whole; with slots 5, 6 stamped: the head)", "action (Gumbel-max argmax)", "env step", "barrier", "(forward: layer 1 + layer 2 half)", "(forward: hand-over barrier)"]
v = [out[i] / T for i in range(7)]
print(f"{task} N={N}: cycles per vector step (s_memtime: shader cycles)")
for n, x in zip(names, v):
    print(f"  {n:30s} {x:8.1f}  ({100 * x / sum(v):.1f} %)")
print(f"  sum {sum(v):.1f} cycles")
