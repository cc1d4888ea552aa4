"""Helper of tests/test_wave_layouts_gpu.py: one minibatch gradient of a fixed rollout -> .npy (run in a subprocess: the kernel-selection switches
are read once per process)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from three_mlagents_amd import _lib  # noqa: E402
from three_mlagents_amd.harness import make_vector_env  # noqa: E402
from three_mlagents_amd.ppo import PPO  # noqa: E402

task, dtype, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
env = make_vector_env(task, n_envs=512, seed=11)
m = PPO("MlpPolicy", env, n_steps=max(32, batch // 512), batch_size=batch, n_epochs=1, seed=11, policy_kwargs={"net_arch": [256, 256], "mfma_dtype": dtype})
m.collect_rollouts()
mb = _lib.Minibatch(None, 7, 0, 0, batch)
_lib.check(_lib.lib().tma_ppo_minibatch_grad(_lib.ptr(m.policy.params), C.byref(m.policy.dims), C.byref(m._rollout_view), C.byref(mb), C.byref(m._hp),
                                             _lib.ptr(m.grad), _lib.ptr(m.workspace), m._stream()))
torch.cuda.synchronize()
np.save(out, m.grad.cpu().numpy())
env.close()
