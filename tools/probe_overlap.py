#!/usr/bin/env python3
"""Does an epoch-prepare launch on a SECOND stream hide under the gradient launches of the running epoch?  Timing probe only: the extra
prepare writes the same workspace region the running epoch reads (a permutation either way: in-bounds, results meaningless)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.harness import make_vector_env
from three_mlagents_amd.ppo import PPO

env = make_vector_env("gridworld", n_envs=4096, seed=1)
m = PPO("MlpPolicy", env, n_steps=1024, batch_size=131072, n_epochs=10, seed=1, policy_kwargs={"net_arch": [64, 64]})
m.collect_rollouts()
L = _lib.lib()
side = torch.cuda.Stream()
total = m.n_steps * m.n_envs


def train(extra):
    perm_seed = 12345
    for e in range(m.n_epochs):
        if extra:
            with torch.cuda.stream(side):
                ep = _lib.Minibatch(None, perm_seed, (e + 1) & 0xFFFFFFFF, 0, total, 0)
                _lib.check(L.tma_ppo_epoch_prepare(C.byref(m._rollout_view), C.byref(ep), m.batch_size, C.byref(m.policy.dims), _lib.ptr(m.workspace),
                                                   side.cuda_stream))
        _lib.check(L.tma_ppo_train_epoch_local(_lib.ptr(m.policy.params), C.byref(m.policy.dims), C.byref(m._rollout_view), perm_seed, e, m.batch_size,
                                               C.byref(m._hp), _lib.ptr(m.grad), _lib.ptr(m.exp_avg), _lib.ptr(m.exp_avg_sq), 1 + 32 * e, 3e-4, 0.9, 0.999,
                                               1e-5, 0.5, _lib.ptr(m.workspace), m._stream()))


for extra in (0, 1, 0, 1):
    train(extra)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        train(extra)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    print(f"extra prepare on a side stream = {extra}: update {ts[len(ts) // 2]:.3f} ms (min {ts[0]:.3f})")
