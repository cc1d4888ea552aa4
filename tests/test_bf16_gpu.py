"""GPU tests of the bf16-MFMA wide-policy path (tma_policy_dims.mfma_dtype = 1; BASELINE.json configs[2] "MLP(256,256) bf16").

The f32 master weights, biases, loss and gradient accumulators are unchanged; MFMA operands are rounded to bf16.  So the
checks are (a) against a torch-CPU emulation that rounds at the same points (tight), (b) against the f32 SB3 restatement
(loose, tolerance = bf16 operand rounding), and (c) exact properties: determinism, rollout/update log-prob agreement."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sb3_ref
from test_ppo_gpu import HP, _flatten_env_major, _hip_grad, _ref_grad_flat, _rollout

pytestmark = pytest.mark.gpu

BF_CONFIGS = [(6, 256, 5, False), (172, 256, 20, True), (4, 128, 5, False), (21, 192, 3, False), (40, 256, 7, True), (100, 128, 4, False),
              (105, 256, 8, True)]  # (the reference's ant task, Ant-v5: the two-pass layout with four layer-1 k-steps, round 6)


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _policies(D, H, A, cont, seed=5):
    from three_mlagents_amd.ppo import HipActorCriticPolicy

    pol = HipActorCriticPolicy(D, A, cont, H, torch.device("cuda", 0), seed=seed, mfma_dtype="bf16")
    sd = pol.state_dict()
    g = torch.Generator().manual_seed(seed)
    if cont:
        sd["log_std"] = torch.linspace(-0.7, 0.3, A)
    sd["action_net.weight"] = sd["action_net.weight"] * 40 + 0.05 * torch.randn(sd["action_net.weight"].shape, generator=g)
    sd["action_net.bias"] = 0.1 * torch.randn(sd["action_net.bias"].shape, generator=g)
    for k in list(sd):
        if k.endswith("bias") and k != "action_net.bias":
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    pol.load_state_dict(sd)
    return pol, sd


def _emulated_forward(sd, obs):
    """SB3 MlpPolicy forward with every MFMA operand rounded to bf16 (weights, inputs, hidden activations), f32 accumulate."""
    def net(prefix, head):
        h = _bf(obs)
        for i in (0, 2):
            h = _bf(torch.tanh(h @ _bf(sd[f"mlp_extractor.{prefix}.{i}.weight"]).t() + sd[f"mlp_extractor.{prefix}.{i}.bias"]))
        return h @ _bf(sd[f"{head}.weight"]).t() + sd[f"{head}.bias"]

    return net("policy_net", "action_net"), net("value_net", "value_net").squeeze(-1)


@pytest.mark.parametrize("D,H,A,cont", BF_CONFIGS)
def test_bf16_forward(D, H, A, cont):
    pol, sd = _policies(D, H, A, cont)
    n = 133
    obs = torch.randn(n, D, generator=torch.Generator().manual_seed(1))
    out_emu, v_emu = _emulated_forward(sd, obs)
    out_ref, v_ref = sb3_ref.forward(sd, obs)
    a, v, lp = pol.act(obs.cuda(), deterministic=True)
    # same rounding points, f32 sums in a different order: a sum that lands on a bf16 rounding boundary may flip one
    # hidden activation by one bf16 ulp (2^-8 relative), hence 4e-3 rather than 1e-5
    assert torch.allclose(v.cpu(), v_emu, rtol=4e-3, atol=4e-3), float((v.cpu() - v_emu).abs().max())
    assert torch.allclose(pol.predict_values(obs.cuda()).cpu(), v.cpu(), rtol=0, atol=0)  # values-only kernel mode == act mode
    assert torch.allclose(v.cpu(), v_ref, rtol=3e-2, atol=3e-2)  # vs the f32 reference: bf16 operand rounding
    if cont:
        assert torch.allclose(a.cpu(), out_emu, rtol=4e-3, atol=4e-3), float((a.cpu() - out_emu).abs().max())
        assert torch.allclose(a.cpu(), out_ref, rtol=3e-2, atol=3e-2)
    else:
        lp_emu = torch.log_softmax(out_emu, dim=1)
        top2 = out_emu.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2e-2  # rows whose argmax is not a near-tie
        assert torch.equal(a.cpu().long()[clear], out_emu.argmax(dim=1)[clear])
        lpa = lp_emu.gather(1, a.cpu().long().unsqueeze(1)).squeeze(1)
        assert torch.allclose(lp.cpu(), lpa, rtol=4e-3, atol=4e-3), float((lp.cpu() - lpa).abs().max())
    # stochastic actions: the log-prob reported for the sampled action vs the emulation's log-prob of that action
    a, v, lp = pol.act(obs.cuda(), rng_seed=9, rng_step=3, deterministic=False)
    if cont:
        std = sd["log_std"].exp()
        lp_e = (-((a.cpu() - out_emu) ** 2) / (2 * std * std) - sd["log_std"] - 0.5 * np.log(2 * np.pi)).sum(dim=1)
        assert torch.allclose(lp.cpu(), lp_e, rtol=1e-2, atol=3e-2), float((lp.cpu() - lp_e).abs().max())
    else:
        lp_e = torch.log_softmax(out_emu, dim=1).gather(1, a.cpu().long().unsqueeze(1)).squeeze(1)
        assert torch.allclose(lp.cpu(), lp_e, rtol=4e-3, atol=4e-3)


def _emulated_grad(sd, obs, actions, old_lp, adv, ret, hp):
    """Gradient of the PPO loss with the kernel's rounding points: bf16 weights / activations / back-propagated deltas as
    GEMM operands, f32 everywhere else (loss, accumulation).  The hidden-layer bias gradients are the column sums of the SAME bf16
    deltas the weight gradients use (ones^T . dz on the MFMA -- what torch.autocast(bfloat16) computes too: grad_output is bf16
    there); the head bias gradient sums the f32 loss gradient.  Returns (grads in SB3 naming, stats)."""
    cont = "log_std" in sd
    X = _bf(obs)
    acts, outs = {}, {}
    for prefix, head in (("policy_net", "action_net"), ("value_net", "value_net")):
        h1 = _bf(torch.tanh(X @ _bf(sd[f"mlp_extractor.{prefix}.0.weight"]).t() + sd[f"mlp_extractor.{prefix}.0.bias"]))
        h2 = _bf(torch.tanh(h1 @ _bf(sd[f"mlp_extractor.{prefix}.2.weight"]).t() + sd[f"mlp_extractor.{prefix}.2.bias"]))
        acts[prefix] = (h1, h2)
        outs[prefix] = (h2 @ _bf(sd[f"{head}.weight"]).t() + sd[f"{head}.bias"]).detach().requires_grad_()
    out_pi, out_v = outs["policy_net"], outs["value_net"]
    ls = sd["log_std"].clone().requires_grad_() if cont else None
    if cont:
        dist = torch.distributions.Normal(out_pi, torch.ones_like(out_pi) * ls.exp())
        log_prob, entropy = dist.log_prob(actions).sum(dim=1), dist.entropy().sum(dim=1)
    else:
        dist = torch.distributions.Categorical(logits=out_pi)
        log_prob, entropy = dist.log_prob(actions.long().flatten()), dist.entropy()
    a = adv
    if hp["normalize_advantage"] and len(a) > 1:
        a = (a - a.mean()) / (a.std() + 1e-8)
    ratio = torch.exp(log_prob - old_lp)
    pl = -torch.min(a * ratio, a * torch.clamp(ratio, 1 - hp["clip_range"], 1 + hp["clip_range"])).mean()
    vl = torch.nn.functional.mse_loss(ret, out_v.squeeze(-1))
    loss = pl + hp["ent_coef"] * (-entropy.mean()) + hp["vf_coef"] * vl
    loss.backward()
    grads = {}
    for prefix, head, dz3 in (("policy_net", "action_net", out_pi.grad), ("value_net", "value_net", out_v.grad)):
        h1, h2 = acts[prefix]
        z3 = _bf(dz3)
        grads[f"{head}.weight"], grads[f"{head}.bias"] = z3.t() @ h2, dz3.sum(0)
        dz2 = (z3 @ _bf(sd[f"{head}.weight"])) * (1 - h2 * h2)
        z2 = _bf(dz2)
        grads[f"mlp_extractor.{prefix}.2.weight"], grads[f"mlp_extractor.{prefix}.2.bias"] = z2.t() @ h1, z2.sum(0)
        dz1 = (z2 @ _bf(sd[f"mlp_extractor.{prefix}.2.weight"])) * (1 - h1 * h1)
        z1 = _bf(dz1)
        grads[f"mlp_extractor.{prefix}.0.weight"], grads[f"mlp_extractor.{prefix}.0.bias"] = z1.t() @ X, z1.sum(0)
    if cont:
        grads["log_std"] = ls.grad
    stats = dict(policy_loss=pl.item(), value_loss=vl.item(), entropy_loss=(-entropy.mean()).item(),
                 clip_fraction=float(((ratio - 1).abs() > hp["clip_range"]).float().mean()), log_prob=log_prob.detach())
    return grads, stats


@pytest.mark.parametrize("D,H,A,cont", BF_CONFIGS)
@pytest.mark.parametrize("B", [33000, 77])
def test_bf16_minibatch_gradient(D, H, A, cont, B):
    T, N = (64, 600) if B > 1000 else (16, 24)
    pol, sd = _policies(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    start = 37
    idx = perm[start:start + B]
    f = lambda x: _flatten_env_major(x, T, N)[idx]  # noqa: E731
    # keep the minibatch's samples clear of the clip boundary as the bf16 forward sees it: a sample within rounding noise of
    # |ratio - 1| = 0.2 takes either branch, which is a whole-sample difference in the gradient, not a tolerance question
    _, st0 = _emulated_grad(sd, f(obs), f(actions), f(old_lp), f(adv), f(ret), HP)
    olp = f(old_lp).clone()
    for _ in range(4):
        ratio = torch.exp(st0["log_prob"].double() - olp.double())
        near = ((ratio - 1.0).abs() - 0.2).abs() < 2e-2
        olp = torch.where(near, olp + 0.08, olp)
    flat_lp = _flatten_env_major(old_lp, T, N).clone()
    flat_lp[idx] = olp
    old_lp = flat_lp.reshape(N, T).t().contiguous()
    grads_emu, stats_emu = _emulated_grad(sd, f(obs), f(actions), f(old_lp), f(adv), f(ret), HP)
    tr = sb3_ref.RefTrainer(sd)
    stats_ref, grads_ref = tr.step(f(obs), f(actions), f(old_lp), f(adv), f(ret), **HP)
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    grad, st, _ = _hip_grad(pol, bufs, T, N, perm, start, B, HP)
    grad2, _, _ = _hip_grad(pol, bufs, T, N, perm, start, B, HP)
    assert torch.equal(grad, grad2)  # slab reduction: bitwise reproducible
    emu, ref = _ref_grad_flat(pol, grads_emu), _ref_grad_flat(pol, grads_ref)
    g = grad.cpu()
    assert torch.isfinite(g).all()
    # (a) against the emulation with the same rounding points -- whole vector and every parameter tensor separately (a
    # mis-laid segment cannot hide behind the big ones).  Residual: f32 sums in another order flip single bf16 roundings.
    rel = ((g - emu).norm() / emu.norm()).item()
    assert rel < 1e-2, rel
    segs = [(k, off, int(np.prod(shape))) for k, off, shape in pol._segments()]
    if cont:
        segs.append(("log_std", pol.offsets[12], pol.act_dim))
    for key, off, cnt in segs:
        r, x = emu[off:off + cnt], g[off:off + cnt]
        assert ((x - r).norm() / max(r.norm().item(), 1e-7)).item() < 2e-2, (key, ((x - r).norm() / r.norm()).item())
    # (b) against the f32 SB3 restatement: same direction, bf16-sized deviation
    cos = torch.dot(g, ref) / (g.norm() * ref.norm())
    assert cos > 0.97, float(cos)
    n = st[5]
    assert n == B
    assert abs(st[0] / n - stats_emu["policy_loss"]) < 2e-3 and abs(st[1] / n - stats_emu["value_loss"]) < 2e-3
    assert abs(-st[2] / n - stats_emu["entropy_loss"]) < 2e-3 and abs(st[4] / n - stats_emu["clip_fraction"]) < 2e-3


@pytest.mark.parametrize("D,H,A,cont", [(172, 256, 20, True), (40, 256, 7, True), (172, 128, 20, True), (50, 192, 4, False), (105, 256, 8, True), (128, 256, 3, True)])
def test_bf16_dw1_from_cached_dz1_equals_recompute_pass(D, H, A, cont, monkeypatch):
    """Two-pass layouts (observations 33..64 / 161..192 wide): dW1 from the dz1 images PASS 0 leaves in the workspace (PASS 2)
    is bit-identical to dW1 from the recomputed chain (PASS 1, the path minibatches beyond the cache take)."""
    T, N, B = 64, 600, 33000
    pol, sd = _policies(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    g_cache, st_cache, _ = _hip_grad(pol, bufs, T, N, perm, 5, B, HP)
    monkeypatch.setenv("TMA_NO_DZ1_CACHE", "1")
    g_recompute, st_recompute, _ = _hip_grad(pol, bufs, T, N, perm, 5, B, HP)
    assert torch.equal(g_cache, g_recompute)
    assert list(st_cache) == list(st_recompute)
    assert g_cache.abs().sum().item() > 0


@pytest.mark.parametrize("D,H,A,cont", [(6, 256, 5, False), (172, 256, 20, True)])
def test_bf16_rollout_and_update_share_the_forward(D, H, A, cont):
    """old_log_prob from policy_act fed back into the update: ratio must be exactly 1 on the first epoch (approx_kl == 0,
    nothing clipped) -- the act kernel and the gradient kernel run the same bf16 forward code."""
    T, N = 8, 64
    pol, sd = _policies(D, H, A, cont)
    obs = torch.randn(T, N, D, generator=torch.Generator().manual_seed(3))
    a, v, lp = pol.act(obs.reshape(T * N, D).cuda(), rng_seed=4, rng_step=0, deterministic=False)
    actions = a.reshape(T, N, A) if cont else a.reshape(T, N).int()
    g = torch.Generator().manual_seed(4)
    bufs = dict(obs=obs, actions=actions.cpu(), old_lp=lp.reshape(T, N).cpu(), adv=torch.randn(T, N, generator=g), ret=torch.randn(T, N, generator=g))
    perm = torch.randperm(T * N, generator=g)
    _, st, _ = _hip_grad(pol, bufs, T, N, perm, 0, T * N, HP)
    assert st[5] == T * N
    assert abs(st[3]) / (T * N) < 1e-9 and st[4] == 0, st  # approx_kl, clip count


def test_bf16_rejects_unsupported_width():
    from three_mlagents_amd.ppo import HipActorCriticPolicy

    with pytest.raises((ValueError, RuntimeError)):
        HipActorCriticPolicy(4, 5, False, 64, torch.device("cuda", 0), seed=1, mfma_dtype="bf16")


def test_bf16_ppo_learns_ball3d():
    from three_mlagents_amd.evaluation import evaluate_policy
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.vec_env import HipVecEnv

    env = HipVecEnv("ball3d", 1024, seed=1)
    model = PPO("MlpPolicy", env, n_steps=128, batch_size=16384, n_epochs=4, ent_coef=0.01, seed=1,
                policy_kwargs={"net_arch": dict(pi=[256, 256], vf=[256, 256]), "mfma_dtype": "bf16"})
    eval_env = HipVecEnv("ball3d", 16, seed=10_001)
    before, _ = evaluate_policy(model, eval_env, n_eval_episodes=32, deterministic=True)
    model.learn(1024 * 128 * 30)
    after, _ = evaluate_policy(model, eval_env, n_eval_episodes=32, deterministic=True)
    print(f"ball3d bf16: deterministic eval reward {before:.2f} -> {after:.2f}; {model.logger_values}")
    assert after > before + 20 and after > 60
    import os
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ball3d_bf16")
        model.save(path)
        loaded = PPO.load(path)
        assert loaded.policy.mfma_dtype == "bf16"
        obs = eval_env.reset()
        a1, _ = model.predict(obs, deterministic=True)
        a2, _ = loaded.predict(obs, deterministic=True)
        assert np.array_equal(a1, a2)
