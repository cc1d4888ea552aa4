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


def evaluate_policy_begin(model, env, n_eval_episodes: int = 10, deterministic: bool = True, max_steps: int = 10_000_000,
                          chunk_steps: int | None = None, params: torch.Tensor | None = None, assume_clean_log: bool = False) -> dict:
    """First half of evaluate_policy: reset + a stream-ordered clear of the env's episode log (tma_env_clear_episode_log) + the FIRST native
    rollout chunk, enqueued on the current stream; nothing here waits for the GPU, so a caller may queue it on a side stream and collect the
    result later (callbacks.EvalCallback).  (`assume_clean_log` is accepted for callers of the round-4 signature and ignored: the clear costs
    no synchronisation, so there is no flag to trust.)  `params`: the flat parameter buffer to evaluate
    (default: the model's live one) -- a snapshot lets the optimizer move on while the chunk runs."""
    eng = env.engine
    if getattr(model, "env", None) is env:  # the training env: its episode log feeds the Monitor file -- leave it alone
        return {"stepwise": evaluate_policy_stepwise(model, env, n_eval_episodes, deterministic, True, True, max_steps)}
    n = eng.num_envs
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
    eng.reset(bufs.obs[0])
    # drop whatever an earlier user of the env left (a step() from a callback, a direct tma_rollout_collect, the previous evaluation's unused
    # episodes and its Monitor aggregate): stream-ordered clear, no host round trip -- the chunk queued next starts from an empty log
    eng.clear_episode_log()
    st = {"model": model, "env": env, "bufs": bufs, "K": K, "n": n, "targets": _targets(n_eval_episodes, n), "counts": np.zeros(n, dtype=np.int64),
          "t_end": np.zeros(n, dtype=np.int64), "found": [], "seed": (model.seed ^ 0xE7A1) & 0xFFFFFFFF, "steps": 0, "max_steps": max_steps,
          "deterministic": deterministic, "params": pol.params if params is None else params, "gamma": float(getattr(model, "gamma", 0.99)), "queued": False}
    _eval_chunk(st)
    return st


def _eval_chunk(st: dict) -> None:
    eng, bufs, K, pol = st["env"].engine, st["bufs"], st["K"], st["model"].policy
    _lib.check(_lib.lib().tma_rollout_collect(eng._h, _lib.ptr(st["params"]), C.byref(pol.dims), C.byref(bufs.rb), 0, K, K, st["seed"], st["steps"] & 0xFFFFFFFF,
                                             eng.env_offset & 0xFFFFFFFF, st["gamma"], 0, 1 if st["deterministic"] else 0, _lib.stream_ptr(eng.device)))
    bufs.obs[0].copy_(bufs.obs[K])  # the next chunk continues from the last observation
    st["steps"] += K
    st["queued"] = True


def evaluate_policy_finish(st: dict, return_episode_rewards: bool = False):
    """Second half: pop the episode log of the chunk in flight (ONE host round trip per chunk, on the current stream -- the stream the chunk was
    queued on), run further chunks while episodes are missing."""
    if "stepwise" in st:
        rewards, lengths = st["stepwise"]
    else:
        eng, counts, targets, t_end, found = st["env"].engine, st["counts"], st["targets"], st["t_end"], st["found"]
        while True:
            if not st["queued"]:
                if not ((counts < targets).any() and st["steps"] < st["max_steps"]):
                    break
                _eval_chunk(st)
            r, l, e, seen = eng.pop_episode_log()  # synchronises the stream
            st["queued"] = False
            if seen > len(r):
                raise RuntimeError(f"evaluation episode log overflowed ({seen} episodes in one chunk of {st['K']} steps, capacity {len(r)})")
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


def evaluate_policy(model, env, n_eval_episodes: int = 10, deterministic: bool = True, return_episode_rewards: bool = False, warn: bool = True,
                    max_steps: int = 10_000_000, chunk_steps: int | None = None):
    return evaluate_policy_finish(evaluate_policy_begin(model, env, n_eval_episodes, deterministic, max_steps, chunk_steps), return_episode_rewards)


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
