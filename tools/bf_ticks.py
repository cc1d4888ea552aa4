#!/usr/bin/env python3
"""Per-phase cycle breakdown of the bf16 wide gradient kernel (diagnostic build libtma_hip_bfticks.so; wave 0 of block 0 of each net).
The stamps distort what they measure (s_memtime drains lgkmcnt at every phase boundary; the instrumented launch runs ~20 % longer) and charge
barrier waits to the phase in front of the barrier: a timing-only build that skipped the P0 commit gained 2 % where the stamps show 10 %.
Use the shares to rank the large phases, not to size the small ones.
Run: make -C three-mlagents_amd/csrc libtma_hip_bfticks.so && TMA_LIB_PATH=three-mlagents_amd/csrc/libtma_hip_bfticks.so python tools/bf_ticks.py [task]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task = sys.argv[1] if len(sys.argv) > 1 else "ball3d"
B = 131072
env = make_vector_env(task, n_envs=4096, seed=1)
m = PPO("MlpPolicy", env, n_steps=B // 4096, batch_size=B, n_epochs=1, seed=1, policy_kwargs={"net_arch": [256, 256], "mfma_dtype": "bf16"})
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
L.tma_debug_bf_ticks.argtypes = [C.c_void_p, C.c_int]
L.tma_debug_bf_ticks(None, 1)
reps = 10
if len(sys.argv) > 2 and sys.argv[2] == "train":  # the path bench.py times: prepared epochs (records in minibatch order), 32 minibatches x 10 epochs
    m2_env = make_vector_env(task, n_envs=4096, seed=1)
    m2 = PPO("MlpPolicy", m2_env, n_steps=1024, batch_size=B, n_epochs=10, seed=1, policy_kwargs={"net_arch": [256, 256], "mfma_dtype": "bf16"})
    m2.collect_rollouts(); m2.train(); torch.cuda.synchronize()
    L.tma_debug_bf_ticks(None, 1)
    m2.collect_rollouts(); m2.train()
    reps = 320
else:
    for _ in range(reps):
        grad()
torch.cuda.synchronize()
L.tma_debug_bf_ticks(out, 0)
names = ["loop top", "P0 commit", "P1 layer 1", "P2 layer 2", "P3 fragment requests", "P3 dz3 images + barrier", "P4 dW3 + dz2", "P5 dW2 (+ barrier)", "tail (after the loop)", "dh1 + dz1 + dW1",
         "P3 head MFMAs + output sums (64-row groups)", "P3 loss", "P3 barrier behind the loss"]
for net, o in (("pi", 0), ("vf", 16)):
    v = [out[o + i] / reps for i in range(13)]
    tot = sum(v)
    print(net, "cycles per launch (block 0):", {n: round(x) for n, x in zip(names, v)}, "sum", round(tot))
    print("    share:", {n: f"{100 * x / tot:.1f}%" for n, x in zip(names, v)})
