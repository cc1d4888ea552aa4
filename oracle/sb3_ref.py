"""torch-CPU restatement of the Stable-Baselines3 2.9.0 pieces on the hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: stable-baselines3 (pin /root/reference/backend/uv.lock:1686-1687) is a third-party dependency that is
neither vendored in the reference nor installable here, and the reference's only test touching it asserts non-None
(/root/reference/backend/tests/test_mlagents.py:74-101).  This file restates the published algorithm (SURVEY.md
Appendix C.3-C.5) with plain torch ops + autograd, at the reference's call sites' hyper-parameters
(/root/reference/backend/mlagents/training.py:361-391), and is what the HIP kernels are compared with.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np
import torch

KEYS = [
    "mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias",
    "mlp_extractor.policy_net.2.weight", "mlp_extractor.policy_net.2.bias",
    "action_net.weight", "action_net.bias",
    "mlp_extractor.value_net.0.weight", "mlp_extractor.value_net.0.bias",
    "mlp_extractor.value_net.2.weight", "mlp_extractor.value_net.2.bias",
    "value_net.weight", "value_net.bias",
]


def init_policy(D: int, H: int, A: int, continuous: bool, seed: int = 0) -> dict[str, torch.Tensor]:
    """ActorCriticPolicy._build: orthogonal init, gains sqrt(2) (extractor nets) / 0.01 (action_net) / 1 (value_net), zero bias."""
    gen = torch.Generator().manual_seed(seed)

    def ortho(out_f, in_f, gain):
        w = torch.empty(out_f, in_f)
        torch.nn.init.orthogonal_(w, gain=gain, generator=gen)
        return w

    g2 = math.sqrt(2.0)
    sd = {
        KEYS[0]: ortho(H, D, g2), KEYS[1]: torch.zeros(H),
        KEYS[2]: ortho(H, H, g2), KEYS[3]: torch.zeros(H),
        KEYS[4]: ortho(A, H, 0.01), KEYS[5]: torch.zeros(A),
        KEYS[6]: ortho(H, D, g2), KEYS[7]: torch.zeros(H),
        KEYS[8]: ortho(H, H, g2), KEYS[9]: torch.zeros(H),
        KEYS[10]: ortho(1, H, 1.0), KEYS[11]: torch.zeros(1),
    }
    if continuous:
        sd["log_std"] = torch.zeros(A)
    return sd


def forward(sd, obs):
    """latent_pi/latent_vf -> (logits or mean, values[B])."""
    lin = torch.nn.functional.linear
    hp = torch.tanh(lin(torch.tanh(lin(obs, sd[KEYS[0]], sd[KEYS[1]])), sd[KEYS[2]], sd[KEYS[3]]))
    hv = torch.tanh(lin(torch.tanh(lin(obs, sd[KEYS[6]], sd[KEYS[7]])), sd[KEYS[8]], sd[KEYS[9]]))
    return lin(hp, sd[KEYS[4]], sd[KEYS[5]]), lin(hv, sd[KEYS[10]], sd[KEYS[11]]).flatten()


def evaluate_actions(sd, obs, actions):
    """ActorCriticPolicy.evaluate_actions -> values, log_prob, entropy."""
    out, values = forward(sd, obs)
    if "log_std" in sd:
        dist = torch.distributions.Normal(out, torch.ones_like(out) * sd["log_std"].exp())
        return values, dist.log_prob(actions).sum(dim=1), dist.entropy().sum(dim=1)
    dist = torch.distributions.Categorical(logits=out)
    return values, dist.log_prob(actions.long().flatten()), dist.entropy()


def ppo_loss(sd, obs, actions, old_log_prob, advantages, returns, *, clip_range=0.2, ent_coef=0.01, vf_coef=0.5, normalize_advantage=True):
    """PPO.train loop body (clip_range_vf=None, target_kl=None)."""
    values, log_prob, entropy = evaluate_actions(sd, obs, actions)
    adv = advantages
    if normalize_advantage and len(adv) > 1:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = torch.exp(log_prob - old_log_prob)
    pl1 = adv * ratio
    pl2 = adv * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)
    policy_loss = -torch.min(pl1, pl2).mean()
    value_loss = torch.nn.functional.mse_loss(returns, values)
    entropy_loss = -torch.mean(entropy)
    loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss
    with torch.no_grad():
        log_ratio = log_prob - old_log_prob
        stats = dict(
            policy_loss=float(policy_loss), value_loss=float(value_loss), entropy_loss=float(entropy_loss),
            approx_kl=float(torch.mean((torch.exp(log_ratio) - 1) - log_ratio)),
            clip_fraction=float(torch.mean((torch.abs(ratio - 1) > clip_range).float())), loss=float(loss),
        )
    return loss, stats


class RefTrainer:
    """Parameters + torch.optim.Adam(lr, eps=1e-5) + clip_grad_norm_(max_grad_norm) exactly as PPO.train applies them."""

    def __init__(self, sd, lr=3e-4, max_grad_norm=0.5):
        self.sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        self.opt = torch.optim.Adam(list(self.sd.values()), lr=lr, eps=1e-5)
        self.max_grad_norm = max_grad_norm

    def step(self, obs, actions, old_log_prob, advantages, returns, **hp):
        loss, stats = ppo_loss(self.sd, obs, actions, old_log_prob, advantages, returns, **hp)
        self.opt.zero_grad()
        loss.backward()
        grads = {k: v.grad.clone() for k, v in self.sd.items()}
        stats["grad_norm"] = float(torch.nn.utils.clip_grad_norm_(list(self.sd.values()), self.max_grad_norm))
        self.opt.step()
        return stats, grads


def flat_index(t, i, T):
    """RolloutBuffer.swap_and_flatten: (T, N, ...) -> swapaxes(0,1).reshape(T*N, ...): f = i*T + t."""
    return i * T + t


def gae_numpy(rewards, values, episode_starts, last_values, dones, gamma=0.99, gae_lambda=0.95):
    """RolloutBuffer.compute_returns_and_advantage, literally (numpy float32 arrays, python-float gamma/lambda)."""
    T = rewards.shape[0]
    adv = np.zeros_like(rewards)
    last = 0
    for step in reversed(range(T)):
        if step == T - 1:
            nnt = 1.0 - dones.astype(np.float32)
            nv = last_values
        else:
            nnt = 1.0 - episode_starts[step + 1]
            nv = values[step + 1]
        delta = rewards[step] + gamma * nv * nnt - values[step]
        last = delta + gamma * gae_lambda * nnt * last
        adv[step] = last
    return adv, adv + values
