"""evaluate_policy with SB3's contract (used at /root/reference/backend/mlagents/training.py:177-184,240-247): run the policy on an
evaluation env until `n_eval_episodes` episodes finished, episodes split evenly over the envs of the vector (SB3's rule: env i
contributes its first (n_eval_episodes + i) // n_envs episodes), episode return / length from the Monitor bookkeeping the step
kernels keep.

Device-side: the evaluation runs as native rollout chunks (`tma_rollout_collect(..., deterministic)`: policy forward, action
selection, env step and auto-reset fused in the same kernels the training rollout uses) over ALL envs of the evaluation vector;
the host synchronises once per chunk to pop the device episode log (return, length, env of every episode that finished in the
chunk, per env in order).  `evaluate_policy_stepwise` is the per-step host loop (one forward launch + one step launch + three
host syncs per vector step); it gives the same episodes bit for bit and is kept as the cross-check (tests/test_dropin_gpu.py)
and for an `env` whose episode log belongs to somebody else (the training env itself).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _targets(n_eval_episodes: int, n: int) -> np.ndarray:
    return np.array([(n_eval_episodes + i) // n for i in range(n)], dtype=np.int64)  # SB3: episodes split evenly over envs


class _EvalBuffers:
    """Rollout planes of one evaluation chunk (K vector steps of the evaluation vector); cached on the env object."""

    def __init__(self, eng, policy, K: int):
        N, D, dev = eng.num_envs, eng.obs_dim, eng.device
        cont = eng.num_actions == 0
        f32 = torch.float32
        self.K = K
        self.key = (K, policy.act_dim, cont)
        self.obs = torch.zeros((K + 1, N, D), dtype=f32, device=dev)
        self.actions = torch.zeros((K, N, policy.act_dim), dtype=f32, device=dev) if cont else torch.zeros((K, N), dtype=torch.int32, device=dev)
        self.rewards, self.values, self.log_probs = (torch.zeros((K, N), dtype=f32, device=dev) for _ in range(3))
        self.terminated, self.truncated = (torch.zeros((K, N), dtype=torch.uint8, device=dev) for _ in range(2))
        self.tobs_slots = max(1, min(max(128, eng.ring_depth), K, (64 << 20) // max(1, N * D * 4)))  # (chunk kernels without a value net need a slot per step)
        self.terminal_obs = torch.zeros((self.tobs_slots, N, D), dtype=f32, device=dev)
        self.last_values = torch.zeros((N,), dtype=f32, device=dev)
        self.rb = _lib.RolloutBuffers(_lib.ptr(self.obs), _lib.ptr(self.actions), _lib.ptr(self.rewards), _lib.ptr(self.values), _lib.ptr(self.log_probs),
                                      _lib.ptr(self.terminated), _lib.ptr(self.truncated), _lib.ptr(self.terminal_obs), _lib.ptr(self.last_values), N, self.tobs_slots)


def evaluate_policy(model, env, n_eval_episodes: int = 10, deterministic: bool = True, return_episode_rewards: bool = False, warn: bool = True,
                    max_steps: int = 10_000_000, chunk_steps: int | None = None):
    eng = env.engine
    if getattr(model, "env", None) is env:  # the training env: its episode log feeds the Monitor file -- leave it alone
        return evaluate_policy_stepwise(model, env, n_eval_episodes, deterministic, return_episode_rewards, warn, max_steps)
    n = eng.num_envs
    targets = _targets(n_eval_episodes, n)
    pol = model.policy
    if pol.device != eng.device:
        raise ValueError(f"policy lives on {pol.device}, the evaluation env on {eng.device}")
    # a chunk covers one episode of the task in most cases (its time limit), capped so that the planes stay small; short limits still
    # get a few dozen steps per host round trip
    K = int(chunk_steps) if chunk_steps else int(min(max(eng.max_episode_steps, 32), 512))
    while K > 1 and K * n > (1 << 22):  # (bounds the episode log: 16 bytes per record)
        K //= 2
    bufs = getattr(env, "_eval_bufs", None)
    if bufs is None or bufs.key != (K, pol.act_dim, eng.num_actions == 0):
        bufs = env._eval_bufs = _EvalBuffers(eng, pol, K)
    cap = max(4096, K * n)  # at most one episode per env and step can finish inside a chunk
    if getattr(eng, "_log_cap", 0) < cap:
        eng.episode_log(cap)
    L = _lib.lib()
    stream = _lib.stream_ptr(eng.device)
    eng.reset(bufs.obs[0])
    eng.pop_episode_log()       # (drop what an earlier user of the env left)
    eng.pop_episode_stats()
    counts, t_end = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    found: list[tuple[int, int, float, int]] = []
    seed = (model.seed ^ 0xE7A1) & 0xFFFFFFFF
    steps = 0
    while (counts < targets).any() and steps < max_steps:
        _lib.check(L.tma_rollout_collect(eng._h, _lib.ptr(pol.params), C.byref(pol.dims), C.byref(bufs.rb), 0, K, K, seed, steps & 0xFFFFFFFF,
                                         eng.env_offset & 0xFFFFFFFF, float(getattr(model, "gamma", 0.99)), 0, 1 if deterministic else 0, stream))
        bufs.obs[0].copy_(bufs.obs[K])  # the next chunk continues from the last observation
        steps += K
        r, l, e, seen = eng.pop_episode_log()  # synchronises the stream: ONE host round trip per chunk
        if seen > len(r):
            raise RuntimeError(f"evaluation episode log overflowed ({seen} episodes in one chunk of {K} steps, capacity {len(r)})")
        for ret, length, i in zip(r.tolist(), l.tolist(), e.tolist()):  # per env in the order the episodes finished
            if counts[i] < targets[i]:
                t_end[i] += int(length)  # the evaluation started from reset(): episode k of env i ends at the sum of its first k lengths
                found.append((int(t_end[i]), int(i), float(ret), int(length)))
                counts[i] += 1
    found.sort()  # by finishing step, then env: the order a per-step loop sees them (evaluate_policy_stepwise)
    rewards, lengths = [f[2] for f in found], [f[3] for f in found]
    if return_episode_rewards:
        return rewards, lengths
    return float(np.mean(rewards)), float(np.std(rewards))


def evaluate_policy_stepwise(model, env, n_eval_episodes: int = 10, deterministic: bool = True, return_episode_rewards: bool = False,
                             warn: bool = True, max_steps: int = 10_000_000):
    """The per-step host loop (policy.act -> engine.step -> read the done flags): same episodes as the chunked evaluation."""
    eng = env.engine
    n = eng.num_envs
    targets = _targets(n_eval_episodes, n)
    counts = np.zeros(n, dtype=int)
    rewards, lengths = [], []
    obs = eng.reset()
    steps = 0
    while (counts < targets).any() and steps < max_steps:
        actions, _, _ = model.policy.act(obs, rng_seed=model.seed ^ 0xE7A1, rng_step=steps, env_offset=eng.env_offset, deterministic=deterministic)
        out = eng.step(actions)
        obs = out["obs"][0]
        done = (out["term"][0] | out["trunc"][0]).bool()
        if bool(done.any()):
            idx = torch.nonzero(done).flatten().cpu().numpy()
            er = out["ep_ret"][0].cpu().numpy()
            el = out["ep_len"][0].cpu().numpy()
            for i in idx:
                if counts[i] < targets[i]:
                    rewards.append(float(er[i]))
                    lengths.append(int(el[i]))
                    counts[i] += 1
        steps += 1
    if return_episode_rewards:
        return rewards, lengths
    return float(np.mean(rewards)), float(np.std(rewards))
