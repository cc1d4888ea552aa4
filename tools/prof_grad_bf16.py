#!/usr/bin/env python3
"""Launch only the wide gradient kernel a few times (for rocprofv3 passes). usage: prof_grad_bf16.py [task hidden dtype [n_envs batch]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

task, H, dt = (sys.argv[1:4] + ["ball3d", "256", "bf16"][len(sys.argv[1:4]):])[:3]
N, B = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (4096, 131072)
env = make_vector_env(task, n_envs=N, seed=1)
m = PPO("MlpPolicy", env, n_steps=max(32, B // N), batch_size=B, n_epochs=1, seed=1, policy_kwargs={"net_arch": [int(H), int(H)], "mfma_dtype": dt})
m.collect_rollouts()
mb = _lib.Minibatch(None, 1, 0, 0, B)
for _ in range(6):
    _lib.check(_lib.lib().tma_ppo_minibatch_grad(_lib.ptr(m.policy.params), C.byref(m.policy.dims), C.byref(m._rollout_view), C.byref(mb), C.byref(m._hp),
                                                 _lib.ptr(m.grad), _lib.ptr(m.workspace), m._stream()))
torch.cuda.synchronize()
