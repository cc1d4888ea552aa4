"""evaluate_policy with SB3's contract (used at /root/reference/backend/mlagents/training.py:177-184,240-247): run the
policy on an evaluation env until `n_eval_episodes` episodes finished; episode return/length come from the Monitor
bookkeeping the step kernel keeps (info["episode"])."""
from __future__ import annotations

import numpy as np
import torch


def evaluate_policy(model, env, n_eval_episodes: int = 10, deterministic: bool = True, return_episode_rewards: bool = False, warn: bool = True,
                    max_steps: int = 10_000_000):
    eng = env.engine
    n = eng.num_envs
    targets = np.array([(n_eval_episodes + i) // n for i in range(n)], dtype=int)  # SB3: episodes split evenly over envs
    counts = np.zeros(n, dtype=int)
    rewards, lengths = [], []
    obs = eng.reset()
    steps = 0
    while (counts < targets).any() and steps < max_steps:
        actions, _, _ = model.policy.act(obs, rng_seed=model.seed ^ 0xE7A1, rng_step=steps, deterministic=deterministic)
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
