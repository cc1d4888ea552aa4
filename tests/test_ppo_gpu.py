"""GPU tests of the actor-critic / PPO kernels against the torch-CPU restatement of SB3 2.9.0 (oracle/sb3_ref.py).
Floating point: tolerances are written next to each check ("parity unpinned" boundary)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import sb3_ref

pytestmark = pytest.mark.gpu

CONFIGS = [(4, 64, 5, False), (6, 256, 5, False), (21, 64, 3, False), (172, 256, 20, True), (4, 128, 5, False)]


def _policy(D, H, A, cont, seed=5):
    from three_mlagents_amd.ppo import HipActorCriticPolicy

    pol = HipActorCriticPolicy(D, A, cont, H, torch.device("cuda", 0), seed=seed)
    sd = pol.state_dict()
    if cont:
        sd["log_std"] = torch.linspace(-0.7, 0.3, A)
    # make the heads non-trivial (gain 0.01 init gives almost uniform logits)
    g = torch.Generator().manual_seed(seed)
    sd["action_net.weight"] = sd["action_net.weight"] * 40 + 0.05 * torch.randn(sd["action_net.weight"].shape, generator=g)
    sd["action_net.bias"] = 0.1 * torch.randn(sd["action_net.bias"].shape, generator=g)
    for k in list(sd):
        if k.endswith("bias") and k != "action_net.bias":
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    pol.load_state_dict(sd)
    return pol, sd


@pytest.mark.parametrize("D,H,A,cont", CONFIGS)
def test_forward_matches_torch_reference(D, H, A, cont):
    pol, sd = _policy(D, H, A, cont)
    sd2 = pol.state_dict()
    for k in sd:
        assert torch.equal(sd[k], sd2[k]), k  # state_dict round trip through the [in][out] layout is exact
    n = 101
    obs = torch.randn(n, D, generator=torch.Generator().manual_seed(1))
    out_ref, v_ref = sb3_ref.forward(sd, obs)
    a, v, lp = pol.act(obs.cuda(), deterministic=True)
    assert torch.allclose(v.cpu(), v_ref, rtol=1e-5, atol=1e-5), float((v.cpu() - v_ref).abs().max())
    assert torch.allclose(pol.predict_values(obs.cuda()).cpu(), v_ref, rtol=1e-5, atol=1e-5)
    if cont:
        assert torch.allclose(a.cpu(), out_ref, rtol=1e-5, atol=1e-5)  # deterministic action = mean
        _, lp_ref, _ = sb3_ref.evaluate_actions(sd, obs, a.cpu())
    else:
        assert torch.equal(a.cpu().long(), out_ref.argmax(dim=1))
        _, lp_ref, _ = sb3_ref.evaluate_actions(sd, obs, a.cpu())
    assert torch.allclose(lp.cpu(), lp_ref, rtol=1e-5, atol=2e-5), float((lp.cpu() - lp_ref).abs().max())
    # stochastic: log_prob of whatever was sampled must equal the reference's log_prob of that action
    a, v, lp = pol.act(obs.cuda(), rng_seed=9, rng_step=3, deterministic=False)
    _, lp_ref, _ = sb3_ref.evaluate_actions(sd, obs, a.cpu())
    assert torch.allclose(lp.cpu(), lp_ref, rtol=1e-5, atol=5e-5), float((lp.cpu() - lp_ref).abs().max())


def test_sampling_distribution_and_streams():
    pol, sd = _policy(4, 64, 5, False)
    obs = torch.zeros(1, 4).repeat(4096, 1)
    probs = torch.softmax(sb3_ref.forward(sd, obs[:1])[0], dim=1)[0]
    counts = torch.zeros(5)
    for step in range(20):
        a, _, _ = pol.act(obs.cuda(), rng_seed=1, rng_step=step)
        counts += torch.bincount(a.cpu().long(), minlength=5).float()
    freq = counts / counts.sum()
    assert torch.allclose(freq, probs, atol=0.01), (freq, probs)  # 81920 draws: 3-sigma ~ 0.005
    a1, _, _ = pol.act(obs.cuda(), rng_seed=1, rng_step=7, env_offset=100)
    a2, _, _ = pol.act(obs.cuda(), rng_seed=1, rng_step=7, env_offset=100)
    a3, _, _ = pol.act(obs[:96].cuda(), rng_seed=1, rng_step=7, env_offset=104)
    assert torch.equal(a1, a2) and torch.equal(a1[4:100], a3)  # counter-based: depends on (seed, global env, step) only


def _rollout(pol, sd, D, A, cont, T, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    obs = torch.randn(T, N, D, generator=g)
    flat = obs.reshape(T * N, D)
    if cont:
        actions = torch.randn(T, N, A, generator=g) * 0.7
        act_flat = actions.reshape(T * N, A)
    else:
        actions = torch.randint(0, A, (T, N), generator=g, dtype=torch.int32)
        act_flat = actions.reshape(T * N)
    with torch.no_grad():
        _, lp, _ = sb3_ref.evaluate_actions(sd, flat, act_flat)
    old_lp = lp + 0.25 * torch.randn(T * N, generator=g)  # ratios spread around 1 -> both clip branches
    # keep every sample clear of the clip boundary: min(r*A, clip(r)*A) switches branch there, so a last-ulp difference in
    # exp(logp - old) between two correct f32 implementations flips one sample's whole gradient (seen at |logp| ~ 60)
    for _ in range(3):
        ratio = torch.exp(lp.double() - old_lp.double())
        near = ((ratio - 1.0).abs() - 0.2).abs() < 5e-3
        old_lp = torch.where(near, old_lp + 0.03, old_lp)
    old_lp = old_lp.reshape(T, N)
    adv = torch.randn(T, N, generator=g)
    ret = torch.randn(T, N, generator=g)
    return obs, actions, old_lp, adv, ret


def _flatten_env_major(x, T, N):
    return x.transpose(0, 1).reshape(T * N, *x.shape[2:])


def _hip_grad(pol, bufs, T, N, indices, start, count, hp, perm=None):
    from three_mlagents_amd import _lib

    dev = torch.device("cuda", 0)
    d = {k: v.to(dev).contiguous() for k, v in bufs.items()}
    rv = _lib.Rollout(_lib.ptr(d["obs"]), _lib.ptr(d["actions"]), _lib.ptr(d["old_lp"]), _lib.ptr(d["adv"]), _lib.ptr(d["ret"]), T, N)
    idx = None if indices is None else indices.to(dev)
    mb = _lib.Minibatch(_lib.ptr(idx), perm[0] if perm else 0, perm[1] if perm else 0, start, count)
    hpar = _lib.PPOHParams(hp["clip_range"], hp["ent_coef"], hp["vf_coef"], 1 if hp["normalize_advantage"] else 0)
    grad = torch.zeros(pol.n_trainable, device=dev)
    ws = torch.zeros(int(_lib.lib().tma_ppo_workspace_bytes(C.byref(pol.dims))), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().tma_ppo_minibatch_grad(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rv), C.byref(mb), C.byref(hpar), _lib.ptr(grad),
                                                 _lib.ptr(ws), _lib.stream_ptr()))
    out = (C.c_double * 8)()
    _lib.check(_lib.lib().tma_ppo_pop_stats(_lib.ptr(ws), out, _lib.stream_ptr()))
    return grad, list(out), ws


def _ref_grad_flat(pol, grads):
    """autograd grads (SB3 naming, [out][in]) -> the engine's flat [in][out] layout."""
    flat = torch.zeros(pol.n_trainable)
    for key, off, shape in pol._segments():
        gk = grads[key].reshape(shape)
        gk = gk.t().contiguous() if len(shape) == 2 else gk
        flat[off:off + gk.numel()] = gk.reshape(-1)
    if pol.continuous:
        flat[pol.offsets[12]:pol.offsets[12] + pol.act_dim] = grads["log_std"]
    return flat


HP = dict(clip_range=0.2, ent_coef=0.01, vf_coef=0.5, normalize_advantage=True)


@pytest.mark.parametrize("D,H,A,cont", CONFIGS)
@pytest.mark.parametrize("B", [256, 77])
def test_minibatch_gradient_matches_autograd(D, H, A, cont, B):
    T, N = 16, 24
    pol, sd = _policy(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    start = 37
    idx = perm[start:start + B]
    f = lambda x: _flatten_env_major(x, T, N)[idx]  # noqa: E731
    tr = sb3_ref.RefTrainer(sd)
    stats_ref, grads_ref = tr.step(f(obs), f(actions), f(old_lp), f(adv), f(ret), **HP)
    grad, st, _ = _hip_grad(pol, dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret), T, N, perm, start, B, HP)
    ref = _ref_grad_flat(pol, grads_ref)
    err = (grad.cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * max(scale, 1.0) + 1e-6, (err, scale)  # f32 sums in a different order (float atomics)
    n = st[5]
    assert n == B
    assert abs(st[0] / n - stats_ref["policy_loss"]) < 1e-5 and abs(st[1] / n - stats_ref["value_loss"]) < 1e-4
    assert abs(-st[2] / n - stats_ref["entropy_loss"]) < 1e-5 and abs(st[3] / n - stats_ref["approx_kl"]) < 1e-5
    assert abs(st[4] / n - stats_ref["clip_fraction"]) < 1e-6
    assert 0.05 < stats_ref["clip_fraction"] < 0.95  # the test exercises both branches of the clipped surrogate


@pytest.mark.parametrize("D,A", [(4, 5), (6, 5)])
def test_large_minibatch_register_accumulating_kernel(D, A):
    """B >= 16384 with H = 64 takes the persistent register-accumulating kernel + slab reduction (no atomics)."""
    H, T, N, B = 64, 64, 400, 20000
    pol, sd = _policy(D, H, A, False)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, False, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    idx = perm[100:100 + B]
    f = lambda x: _flatten_env_major(x, T, N)[idx]  # noqa: E731
    tr = sb3_ref.RefTrainer(sd)
    stats_ref, grads_ref = tr.step(f(obs), f(actions), f(old_lp), f(adv), f(ret), **HP)
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    grad, st, _ = _hip_grad(pol, bufs, T, N, perm, 100, B, HP)
    grad2, _, _ = _hip_grad(pol, bufs, T, N, perm, 100, B, HP)
    assert torch.equal(grad, grad2)  # slab reduction: bitwise reproducible
    ref = _ref_grad_flat(pol, grads_ref)
    err, scale = (grad.cpu() - ref).abs().max().item(), ref.abs().max().item()
    assert err <= 2e-5 * max(scale, 1.0) + 1e-6, (err, scale)
    assert st[5] == B and abs(st[0] / B - stats_ref["policy_loss"]) < 1e-5 and abs(st[1] / B - stats_ref["value_loss"]) < 1e-4
    assert abs(-st[2] / B - stats_ref["entropy_loss"]) < 1e-5 and abs(st[4] / B - stats_ref["clip_fraction"]) < 1e-6
    # the generic (atomic) kernel on the same minibatch split in two halves gives the same gradient
    hp2 = dict(HP, normalize_advantage=False)
    g_full, _, _ = _hip_grad(pol, bufs, T, N, perm, 100, B, hp2)
    g_a, _, _ = _hip_grad(pol, bufs, T, N, perm, 100, B // 2, hp2)
    g_b, _, _ = _hip_grad(pol, bufs, T, N, perm, 100 + B // 2, B - B // 2, hp2)
    assert torch.allclose(g_full, 0.5 * (g_a + g_b), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("D,H,A,cont", [(6, 256, 5, False), (172, 256, 20, True), (4, 128, 5, False), (21, 256, 3, False),
                                        (105, 256, 8, True),    # the reference's ant task (Ant-v5): two passes with seven k-tiles (round 6)
                                        (105, 256, 6, False)])  # the same width with a Discrete head: the runtime-width kernel
def test_wide_policy_column_parallel_kernel(D, H, A, cont):
    """B >= 32768 with H in {128, 256} takes the column-parallel register-accumulating kernel (slab reduction, no atomics
    except none at all for D <= 32)."""
    T, N, B = 64, 600, 33000
    pol, sd = _policy(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    idx = perm[50:50 + B]
    f = lambda x: _flatten_env_major(x, T, N)[idx]  # noqa: E731
    tr = sb3_ref.RefTrainer(sd)
    stats_ref, grads_ref = tr.step(f(obs), f(actions), f(old_lp), f(adv), f(ret), **HP)
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    grad, st, _ = _hip_grad(pol, bufs, T, N, perm, 50, B, HP)
    grad2, _, _ = _hip_grad(pol, bufs, T, N, perm, 50, B, HP)
    assert torch.equal(grad, grad2)  # deterministic
    ref = _ref_grad_flat(pol, grads_ref)
    err, scale = (grad.cpu() - ref).abs().max().item(), ref.abs().max().item()
    assert err <= 2e-5 * max(scale, 1.0) + 1e-6, (err, scale)
    assert st[5] == B and abs(st[0] / B - stats_ref["policy_loss"]) < 1e-5 and abs(st[1] / B - stats_ref["value_loss"]) < 1e-4
    assert abs(-st[2] / B - stats_ref["entropy_loss"]) < 1e-4 and abs(st[4] / B - stats_ref["clip_fraction"]) < 1e-6


@pytest.mark.parametrize("D,H,A,cont", [(172, 256, 20, True), (165, 128, 4, False), (176, 192, 3, False), (105, 256, 8, True), (98, 256, 3, True)])
def test_wide_dw1_from_cached_dz1_equals_recompute_pass(D, H, A, cont, monkeypatch):
    """Crawler-width observations (161..176): the second launch computes dW1 from the dz1 operands the first one left in the
    workspace; beyond the cache (or with the test hook) it recomputes the chain instead -- the two must agree bit for bit."""
    T, N, B = 64, 600, 33000
    pol, sd = _policy(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    g_cache, st_cache, _ = _hip_grad(pol, bufs, T, N, perm, 5, B, HP)
    monkeypatch.setenv("TMA_NO_DZ1_CACHE", "1")
    g_recompute, st_recompute, _ = _hip_grad(pol, bufs, T, N, perm, 5, B, HP)
    assert torch.equal(g_cache, g_recompute)
    assert list(st_cache) == list(st_recompute)
    assert g_cache.abs().sum().item() > 0


def test_full_batch_feistel_permutation_is_a_bijection():
    """The on-device minibatch permutation visits every sample exactly once: with normalisation off, the gradient of
    the whole buffer taken as 5 permuted minibatches (scaled by their sizes) equals the one-shot identity-order gradient."""
    D, H, A, cont, T, N = 4, 64, 5, False, 20, 19  # 380 samples: not a power of two -> cycle walking is exercised
    pol, sd = _policy(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    hp = dict(HP, normalize_advantage=False)
    total = T * N
    g_id, st_id, _ = _hip_grad(pol, bufs, T, N, torch.arange(total), 0, total, hp)
    acc = torch.zeros_like(g_id)
    seen = 0
    for start in range(0, total, 80):
        cnt = min(80, total - start)
        g, st, _ = _hip_grad(pol, bufs, T, N, None, start, cnt, hp, perm=(123, 4))
        acc += g * (cnt / total)
        seen += st[5]
    assert seen == total
    assert torch.allclose(acc, g_id, rtol=1e-4, atol=2e-6), float((acc - g_id).abs().max())


@pytest.mark.parametrize("D,H,A,cont", [(4, 64, 5, False), (172, 256, 20, True)])
def test_adam_and_clip_match_torch(D, H, A, cont):
    from three_mlagents_amd import _lib

    T, N, B = 8, 32, 256
    pol, sd = _policy(D, H, A, cont)
    tr = sb3_ref.RefTrainer(sd, lr=3e-4, max_grad_norm=0.5)
    dev = torch.device("cuda", 0)
    m = torch.zeros(pol.n_trainable, device=dev)
    v = torch.zeros(pol.n_trainable, device=dev)
    for step in range(1, 4):
        obs, actions, old_lp, adv, ret = _rollout(pol, tr.sd if step == 1 else {k: t.detach() for k, t in tr.sd.items()}, D, A, cont, T, N, seed=step)
        idx = torch.arange(B)
        f = lambda x: _flatten_env_major(x, T, N)[idx]  # noqa: E731
        stats_ref, _ = tr.step(f(obs), f(actions), f(old_lp), f(adv), f(ret), **HP)
        grad, st, ws = _hip_grad(pol, dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret), T, N, idx, 0, B, HP)
        _lib.check(_lib.lib().tma_ppo_adam_step(_lib.ptr(pol.params), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), C.byref(pol.dims), step, 3e-4, 0.9, 0.999,
                                                1e-5, 0.5, 1.0, _lib.ptr(ws), _lib.stream_ptr()))
        out = (C.c_double * 8)()
        _lib.check(_lib.lib().tma_ppo_pop_stats(_lib.ptr(ws), out, _lib.stream_ptr()))
        assert abs(out[6] - stats_ref["grad_norm"]) <= 1e-4 * max(1.0, stats_ref["grad_norm"])
        assert float(grad.abs().max()) == 0.0  # gradient buffer re-zeroed for the next minibatch
        sd_new = pol.state_dict()
        for k in sd_new:
            ref = tr.sd[k].detach()
            assert torch.allclose(sd_new[k], ref, rtol=0, atol=3e-6), (step, k, float((sd_new[k] - ref).abs().max()))
    # the [out][in] copies used by the backward pass follow the update
    obs = torch.randn(33, D)
    _, v_ref = sb3_ref.forward({k: t.detach() for k, t in tr.sd.items()}, obs)
    assert torch.allclose(pol.predict_values(obs.cuda()).cpu(), v_ref, rtol=1e-4, atol=1e-5)
    # ... and every derived region (transposed copies, LDS images, bf16 images) equals what a full tma_policy_sync rebuilds
    # (the H = 64 optimizer kernel scatters the updated parameters into them itself instead of launching the refresh)
    after_step = pol.params.clone()
    _lib.check(_lib.lib().tma_policy_sync(_lib.ptr(pol.params), C.byref(pol.dims), _lib.stream_ptr()))
    assert torch.equal(after_step, pol.params)


def test_timeout_bootstrap():
    from three_mlagents_amd import _lib

    pol, sd = _policy(4, 64, 5, False)
    n = 300
    g = torch.Generator().manual_seed(0)
    tobs = torch.randn(n, 4, generator=g)
    trunc = (torch.rand(n, generator=g) < 0.1).to(torch.uint8)
    rew = torch.randn(n, generator=g)
    _, v_ref = sb3_ref.forward(sd, tobs)
    expect = torch.where(trunc.bool(), rew + np.float32(0.99) * v_ref, rew)
    r, tobs_d, trunc_d = rew.cuda(), tobs.cuda(), trunc.cuda()  # keep the device tensors alive across the launch
    _lib.check(_lib.lib().tma_policy_bootstrap(_lib.ptr(pol.params), C.byref(pol.dims), _lib.ptr(tobs_d), _lib.ptr(trunc_d), n, 0.99,
                                               _lib.ptr(r), _lib.stream_ptr()))
    assert torch.allclose(r.cpu(), expect, rtol=0, atol=1e-5)
    assert torch.equal(r.cpu()[~trunc.bool()], rew[~trunc.bool()])


@pytest.mark.parametrize("T,N", [(129, 70), (1, 1), (7, 16), (128, 17), (300, 129), (1024, 4096), (33, 70000)])
@pytest.mark.parametrize("scan", [False, True])
def test_gae_flags_equals_sb3_layout(monkeypatch, T, N, scan):
    """Both GAE kernels -- the producer / consumer kernel of small vectors (16 envs per workgroup, 128-step chunks through LDS: ragged env
    groups, partial and single chunks, T = 1) and the one-thread-per-env scan (TMA_GAE_SCAN=1, and every vector beyond 65 536 envs) -- in
    both buffer layouts (engine flags / SB3 episode_starts + dones) against the C oracle, which equals the literal NumPy loop: bit for bit."""
    from three_mlagents_amd import _lib

    if scan:
        monkeypatch.setenv("TMA_GAE_SCAN", "1")
    else:
        monkeypatch.delenv("TMA_GAE_SCAN", raising=False)
    rng = np.random.default_rng(1 + T + N)
    r, v = rng.normal(size=(T, N)).astype(np.float32), rng.normal(size=(T, N)).astype(np.float32)
    term = (rng.random((T, N)) < 0.05).astype(np.uint8)
    trunc = ((rng.random((T, N)) < 0.03) & (term == 0)).astype(np.uint8)
    lv = rng.normal(size=N).astype(np.float32)
    done = (term | trunc).astype(np.float32)
    es = np.concatenate([np.zeros((1, N), np.float32), done[:-1]])
    adv_ref, ret_ref = orc.gae(r, v, es, lv, done[-1].astype(np.uint8))
    if T * N <= 100_000:
        adv_np, ret_np = sb3_ref.gae_numpy(r, v, es, lv, done[-1].astype(bool))
        assert np.array_equal(adv_ref, adv_np) and np.array_equal(ret_ref, ret_np)  # C oracle == literal numpy restatement
    t = [torch.from_numpy(x).cuda() for x in (r, v, term, trunc, lv)]
    adv, ret = torch.full_like(t[0], float("nan")), torch.full_like(t[0], float("nan"))
    _lib.check(_lib.lib().tma_gae_flags(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(t[4]), 0.99, 0.95, T, N, _lib.ptr(adv),
                                        _lib.ptr(ret), _lib.stream_ptr()))
    assert np.array_equal(adv.cpu().numpy(), adv_ref) and np.array_equal(ret.cpu().numpy(), ret_ref)
    # SB3 layout: float episode_starts[T][N] + the final dones[N]
    es_t, dn_t = torch.from_numpy(es).cuda(), torch.from_numpy(done[-1].astype(np.uint8)).cuda()
    adv2, ret2 = torch.full_like(t[0], float("nan")), torch.full_like(t[0], float("nan"))
    _lib.check(_lib.lib().tma_gae(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(es_t), _lib.ptr(t[4]), _lib.ptr(dn_t), 0.99, 0.95, T, N, _lib.ptr(adv2),
                                  _lib.ptr(ret2), _lib.stream_ptr()))
    assert np.array_equal(adv2.cpu().numpy(), adv_ref) and np.array_equal(ret2.cpu().numpy(), ret_ref)


@pytest.mark.parametrize("task,hidden,N,mfma", [("gridworld", 64, 200, "f32"), ("push", 64, 200, "f32"), ("ball3d", 64, 200, "f32"), ("walljump", 64, 200, "f32"),
                                                ("basic", 64, 200, "f32"), ("gridworld", 128, 200, "f32"), ("gridworld", 64, 16400, "f32"),
                                                # the 256-wide bf16 fused chunk (register-resident layer-2 weights, 32-env row groups; 200 = a ragged last group)
                                                ("ball3d", 256, 200, "bf16"), ("gridworld", 256, 200, "bf16"), ("push", 256, 96, "bf16"), ("basic", 256, 40, "bf16"),
                                                ("walljump", 256, 64, "bf16"), ("ball3d", 256, 200, "f32"),
                                                # the f32 256-wide fused chunk (the reference's default net and dtype: policy net only, values / bootstrap in batches;
                                                # up to 2048 envs in tiles of 8 on the broadcast 4x4x1 MFMA: 8 = the reference's own env count, 200 / 5 = full tiles / a
                                                # ragged one, 40 with Basic's inline resets; beyond that 16-env tiles: 2100 = 131 full tiles + a ragged one)
                                                ("basic", 256, 8, "f32"), ("basic", 256, 40, "f32"), ("basic", 256, 5, "f32"), ("gridworld", 256, 200, "f32"),
                                                ("gridworld", 256, 2100, "f32"), ("push", 256, 72, "f32"),
                                                ("walljump", 256, 64, "f32"), ("bicycle", 256, 40, "f32"), ("glider", 256, 40, "f32"),
                                                # the Box-action fused chunk (Crawler shape: layer-1 fragments streamed, env state in LDS); 72 = a ragged group
                                                ("crawler", 256, 72, "bf16"), ("crawler", 256, 40, "f32"), ("ant", 256, 72, "bf16"), ("ant", 256, 40, "f32"),
                                                # (f32: 8-env tiles since round 6 -- 12 = a ragged second tile)
                                                ("crawler", 256, 12, "f32"),
                                                # the float64-physics tasks of SURVEY 8f: fused on 64-wide and 256-wide bf16 nets (Bicycle, Glider), per-step otherwise
                                                ("bicycle", 64, 200, "f32"), ("glider", 64, 200, "f32"), ("bicycle", 256, 72, "bf16"), ("glider", 256, 72, "bf16"),
                                                ("brickbreak", 64, 100, "f32"),
                                                # BrickBreak (45 observations) on the 8-env tiles of the f32 256-wide chunk (round 6); 2100 envs: per-step
                                                ("brickbreak", 256, 8, "f32"), ("brickbreak", 256, 44, "f32"), ("brickbreak", 256, 2100, "f32")])
def test_native_rollout_equals_stepwise_composition(task, hidden, N, mfma):
    """tma_rollout_collect (fused multi-step kernels: H=64 on gridworld/push/ball3d/walljump, 256-wide bf16 on every Discrete task;
    per-step launches otherwise) == policy.act -> env.step -> bootstrap composed step by step: bit-identical (same arithmetic, same
    RNG counters)."""
    from three_mlagents_amd import _lib
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.vec_env import HipVecEnv

    # N = 200: two-wave fused kernel (one tile per block); N = 16400: >= 1024 tiles, the four-tiles-per-block single-wave kernel
    T = {"ball3d": 230, "gridworld": 130, "walljump": 170, "basic": 60, "crawler": 40, "ant": 40, "bicycle": 110, "glider": 110, "brickbreak": 70}.get(task, 48)  # long enough to reach the time limit (timeout-bootstrap path)
    env = HipVecEnv(task, N, seed=3, ring_depth=16)
    model = PPO("MlpPolicy", env, n_steps=T, batch_size=256, n_epochs=1, seed=3, policy_kwargs={"net_arch": [hidden, hidden], "mfma_dtype": mfma})
    _lib.check(_lib.lib().tma_debug_poison_lds(0, _lib.stream_ptr()))  # NaNs in every LDS word the rollout kernels do not write themselves
    assert model.collect_rollouts()
    b = {k: v.clone() for k, v in model.buf.items()}
    assert all(torch.isfinite(b[k]).all() for k in ("values", "log_probs", "rewards", "obs"))
    env2 = HipVecEnv(task, N, seed=3, ring_depth=16)
    eng = env2.engine
    obs = eng.reset()
    assert torch.equal(obs, b["obs"][0])
    for t in range(T):
        a, v, lp = model.policy.act(obs, rng_seed=3, rng_step=t, env_offset=0)
        out = eng.step(a)
        rew = out["rew"][0].clone()
        _lib.check(_lib.lib().tma_policy_bootstrap(_lib.ptr(model.policy.params), C.byref(model.policy.dims), _lib.ptr(out["term_obs"][0]),
                                                   _lib.ptr(out["trunc"][0]), N, 0.99, _lib.ptr(rew), _lib.stream_ptr()))
        assert torch.equal(a, b["actions"][t]) and torch.equal(v, b["values"][t]) and torch.equal(lp, b["log_probs"][t]), t
        assert torch.equal(rew, b["rewards"][t]) and torch.equal(out["term"][0], b["terminated"][t]) and torch.equal(out["trunc"][0], b["truncated"][t])
        obs = out["obs"][0].clone()
        assert torch.equal(obs, b["obs"][t + 1])
    assert torch.equal(model.policy.predict_values(obs), b["last_values"])
    if task in ("gridworld", "ball3d", "walljump", "basic") and N >= 40:
        assert int(b["truncated"].sum()) > 0  # the timeout-bootstrap branch was exercised
    # GAE of the rollout vs the oracle on the same planes
    done = (b["terminated"] | b["truncated"]).float().cpu().numpy()
    es = np.concatenate([np.zeros((1, N), np.float32), done[:-1]])
    adv_ref, ret_ref = orc.gae(b["rewards"].cpu().numpy(), b["values"].cpu().numpy(), es, b["last_values"].cpu().numpy(), done[-1].astype(np.uint8))
    assert np.array_equal(b["advantages"].cpu().numpy(), adv_ref) and np.array_equal(b["returns"].cpu().numpy(), ret_ref)


@pytest.mark.parametrize("task,n_envs,iters,thresh", [("basic", 256, 30, 0.5), ("gridworld", 4096, 50, 0.6)])
def test_ppo_learns(task, n_envs, iters, thresh):
    from three_mlagents_amd.evaluation import evaluate_policy
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.vec_env import HipVecEnv

    env = HipVecEnv(task, n_envs, seed=1)
    model = PPO("MlpPolicy", env, n_steps=64, batch_size=max(2048, n_envs * 8), n_epochs=4, ent_coef=0.01, seed=1, policy_kwargs={"net_arch": [64, 64]})
    eval_env = HipVecEnv(task, 16, seed=10_001)
    before, _ = evaluate_policy(model, eval_env, n_eval_episodes=32, deterministic=True)
    model.learn(n_envs * 64 * iters)
    after, _ = evaluate_policy(model, eval_env, n_eval_episodes=32, deterministic=True)
    print(f"{task}: deterministic eval reward {before:.3f} -> {after:.3f}; train stats {model.logger_values}")
    assert after > thresh and after > before
    # save / load round trip gives the same deterministic actions
    import tempfile, os

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, f"{task}_policy_test")
        model.save(path)
        loaded = PPO.load(path)
        obs = eval_env.reset()
        a1, _ = model.predict(obs, deterministic=True)
        a2, _ = loaded.predict(obs, deterministic=True)
        assert np.array_equal(a1, a2)
        with pytest.raises(FileNotFoundError):
            PPO.load(os.path.join(d, "missing.zip"))


def test_epoch_prepare_equals_per_minibatch_pass():
    """tma_ppo_epoch_prepare (one launch per epoch) + prepared minibatches give bit-identical gradients and statistics to
    the self-contained per-minibatch calls, including the ragged last minibatch."""
    from three_mlagents_amd import _lib

    D, H, A, T, N = 4, 64, 5, 48, 700
    pol, sd = _policy(D, H, A, False)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, False, T, N)
    dev = torch.device("cuda", 0)
    d = {k: v.to(dev).contiguous() for k, v in dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret).items()}
    rv = _lib.Rollout(_lib.ptr(d["obs"]), _lib.ptr(d["actions"]), _lib.ptr(d["old_lp"]), _lib.ptr(d["adv"]), _lib.ptr(d["ret"]), T, N)
    hpar = _lib.PPOHParams(0.2, 0.01, 0.5, 1)
    L = _lib.lib()
    total, batch = T * N, 9000  # 33600 samples -> 3 full minibatches + one of 6600
    ws = torch.zeros(int(L.tma_ppo_workspace_bytes(C.byref(pol.dims))), dtype=torch.uint8, device=dev)

    def run(prepared):
        out = []
        if prepared:
            ep = _lib.Minibatch(None, 77, 3, 0, total, 0)
            _lib.check(L.tma_ppo_epoch_prepare(C.byref(rv), C.byref(ep), batch, C.byref(pol.dims), _lib.ptr(ws), _lib.stream_ptr()))
        for start in range(0, total, batch):
            mb = _lib.Minibatch(None, 77, 3, start, min(batch, total - start), batch if prepared else 0)
            grad = torch.zeros(pol.n_trainable, device=dev)
            _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rv), C.byref(mb), C.byref(hpar), _lib.ptr(grad),
                                                _lib.ptr(ws), _lib.stream_ptr()))
            st = (C.c_double * 8)()
            _lib.check(L.tma_ppo_pop_stats(_lib.ptr(ws), st, _lib.stream_ptr()))
            out.append((grad.cpu(), list(st)[:6]))
        return out

    a, b = run(False), run(True)
    assert len(a) == 4
    for (g0, s0), (g1, s1) in zip(a, b):
        assert torch.equal(g0, g1) and s0 == s1
    bad = _lib.Minibatch(None, 77, 3, 100, 500, batch)  # start not on the prepared split
    grad = torch.zeros(pol.n_trainable, device=dev)
    assert L.tma_ppo_minibatch_grad(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rv), C.byref(bad), C.byref(hpar), _lib.ptr(grad), _lib.ptr(ws),
                                    _lib.stream_ptr()) != 0


@pytest.mark.parametrize("D,H,A,cont,dtype", [(4, 64, 5, False, "f32"), (6, 256, 5, False, "f32"), (6, 256, 5, False, "bf16"),
                                                (172, 256, 20, True, "bf16"), (21, 192, 3, False, "bf16"), (40, 128, 7, True, "f32")])
def test_adam_step_local_equals_adam_step(D, H, A, cont, dtype):
    """tma_ppo_adam_step_local (norm from the reduction's partials, derived copies written by the optimizer kernel) against
    the three-launch tma_ppo_adam_step on the same gradient: same norm / parameters up to the f64 summation order of the norm."""
    from three_mlagents_amd import _lib

    T, N, B = 32, 64, 1024
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    res = []
    for local in (False, True):
        if dtype == "bf16":
            from test_bf16_gpu import _policies

            pol, sd = _policies(D, H, A, cont)
        else:
            pol, sd = _policy(D, H, A, cont)
        obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
        m = torch.zeros(pol.n_trainable, device=dev)
        v = torch.zeros(pol.n_trainable, device=dev)
        norms = []
        for step in range(1, 4):
            grad, st, ws = _hip_grad(pol, dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret), T, N, None, 100 * step, B, HP, perm=(5, step))
            if local:
                _lib.check(L.tma_ppo_adam_step_local(_lib.ptr(pol.params), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), C.byref(pol.dims), step, 3e-4, 0.9,
                                                     0.999, 1e-5, 0.5, _lib.ptr(ws), _lib.stream_ptr(), B))
            else:
                _lib.check(L.tma_ppo_adam_step(_lib.ptr(pol.params), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v), C.byref(pol.dims), step, 3e-4, 0.9, 0.999,
                                               1e-5, 0.5, 1.0, _lib.ptr(ws), _lib.stream_ptr()))
            out = (C.c_double * 8)()
            _lib.check(L.tma_ppo_pop_stats(_lib.ptr(ws), out, _lib.stream_ptr()))
            norms.append((out[6], out[7]))
            assert float(grad.abs().max()) == 0.0
        after = pol.params.clone()
        _lib.check(L.tma_policy_sync(_lib.ptr(pol.params), C.byref(pol.dims), _lib.stream_ptr()))
        assert torch.equal(after, pol.params)  # every derived region is what a full refresh rebuilds
        res.append((after.cpu(), m.cpu(), v.cpu(), norms))
    (p0, m0, v0, n0), (p1, m1, v1, n1) = res
    for (a0, c0), (a1, c1) in zip(n0, n1):
        assert abs(a0 - a1) <= 1e-6 * max(1.0, a0) and abs(c0 - c1) <= 1e-6
    assert torch.allclose(p0, p1, rtol=0, atol=1e-7) and torch.allclose(m0, m1, rtol=1e-6, atol=1e-9) and torch.allclose(v0, v1, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("D,H,A,cont", [(4, 64, 5, False), (6, 256, 5, False), (172, 256, 20, True)])
def test_global_advantage_statistics_of_a_two_rank_minibatch(D, H, A, cont):
    """Data-parallel advantage normalisation: two "ranks" = two different rollouts under the same policy.  Each rank's
    epoch is prepared, the per-minibatch (sum, sumsq) pairs are exported, added (what the all-reduce does) and imported; each rank's
    gradient with stats_count = global rows must equal autograd of ITS rows with advantages normalised by the mean / unbiased std of
    BOTH ranks' rows -- i.e. what one SB3 run over the concatenated minibatch computes, split in two."""
    from three_mlagents_amd import _lib

    T, N, B = 16, 64, 512  # two minibatches of 512 per rank
    pol, sd = _policy(D, H, A, cont)
    dev, L, total = torch.device("cuda", 0), _lib.lib(), T * N
    hpar = _lib.PPOHParams(HP["clip_range"], HP["ent_coef"], HP["vf_coef"], 1)
    ranks = []
    for r in range(2):
        obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N, seed=10 + r)
        adv = adv * (1.0 + r) + 0.5 * r  # the shards' advantage distributions differ, so local != global statistics
        d = {k: v.to(dev).contiguous() for k, v in dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret).items()}
        rv = _lib.Rollout(_lib.ptr(d["obs"]), _lib.ptr(d["actions"]), _lib.ptr(d["old_lp"]), _lib.ptr(d["adv"]), _lib.ptr(d["ret"]), T, N)
        ws = torch.zeros(int(L.tma_ppo_workspace_bytes(C.byref(pol.dims))), dtype=torch.uint8, device=dev)
        idx = torch.randperm(total, generator=torch.Generator().manual_seed(20 + r)).to(dev)
        ep = _lib.Minibatch(_lib.ptr(idx), 0, 0, 0, total, 0)
        _lib.check(L.tma_ppo_epoch_prepare(C.byref(rv), C.byref(ep), B, C.byref(pol.dims), _lib.ptr(ws), _lib.stream_ptr()))
        sums = torch.zeros(4, dtype=torch.float64, device=dev)
        _lib.check(L.tma_ppo_epoch_adv_sums(_lib.ptr(ws), C.byref(pol.dims), B, total, _lib.ptr(sums), 0, _lib.stream_ptr()))
        ranks.append(dict(cpu=dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret), dev=d, rv=rv, ws=ws, idx=idx, sums=sums))
    for r in range(2):  # exported sums are the float64 sums of each minibatch's advantages
        flat = _flatten_env_major(ranks[r]["cpu"]["adv"], T, N).double()[ranks[r]["idx"].cpu()]
        want = torch.stack([flat[:B].sum(), (flat[:B] ** 2).sum(), flat[B:].sum(), (flat[B:] ** 2).sum()])
        assert torch.allclose(ranks[r]["sums"].cpu(), want, rtol=1e-12, atol=1e-9)
    reduced = ranks[0]["sums"] + ranks[1]["sums"]
    for r in range(2):
        rk = ranks[r]
        _lib.check(L.tma_ppo_epoch_adv_sums(_lib.ptr(rk["ws"]), C.byref(pol.dims), B, total, _lib.ptr(reduced.clone()), 1, _lib.stream_ptr()))
        for k in range(2):
            mb = _lib.Minibatch(_lib.ptr(rk["idx"]), 0, 0, k * B, B, B, 2 * B)
            grad = torch.zeros(pol.n_trainable, device=dev)
            _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rk["rv"]), C.byref(mb), C.byref(hpar), _lib.ptr(grad),
                                                _lib.ptr(rk["ws"]), _lib.stream_ptr()))
            sel = [_flatten_env_major(ranks[q]["cpu"]["adv"], T, N)[ranks[q]["idx"].cpu()[k * B:(k + 1) * B]] for q in range(2)]
            both = torch.cat(sel)
            mean, std = both.mean(), both.std()  # torch .std() is the unbiased estimator SB3 uses
            rows = rk["idx"].cpu()[k * B:(k + 1) * B]
            f = lambda x: _flatten_env_major(x, T, N)[rows]  # noqa: E731
            advn = (f(rk["cpu"]["adv"]) - mean) / (std + 1e-8)
            tr = sb3_ref.RefTrainer(sd)
            _, grads_ref = tr.step(f(rk["cpu"]["obs"]), f(rk["cpu"]["actions"]), f(rk["cpu"]["old_lp"]), advn, f(rk["cpu"]["ret"]),
                                   **dict(HP, normalize_advantage=False))
            ref = _ref_grad_flat(pol, grads_ref)
            err, scale = (grad.cpu() - ref).abs().max().item(), ref.abs().max().item()
            assert err <= 2e-5 * max(scale, 1.0) + 1e-6, (r, k, err, scale)
            # and the rank-local statistics give a measurably different gradient (the test would not notice a no-op otherwise)
            mb_local = _lib.Minibatch(_lib.ptr(rk["idx"]), 0, 0, k * B, B, 0, 0)
            g_local = torch.zeros(pol.n_trainable, device=dev)
            _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rk["rv"]), C.byref(mb_local), C.byref(hpar),
                                                _lib.ptr(g_local), _lib.ptr(rk["ws"]), _lib.stream_ptr()))
            assert (g_local.cpu() - ref).abs().max().item() > 1e-3 * scale
    with pytest.raises(ValueError):  # stats_count without a prepared epoch
        mb = _lib.Minibatch(_lib.ptr(ranks[0]["idx"]), 0, 0, 0, B, 0, 2 * B)
        _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(ranks[0]["rv"]), C.byref(mb), C.byref(hpar),
                                            _lib.ptr(torch.zeros(pol.n_trainable, device=dev)), _lib.ptr(ranks[0]["ws"]), _lib.stream_ptr()))


@pytest.mark.parametrize("task,hidden,mfma,batch", [("gridworld", 64, "f32", 256), ("gridworld", 64, "f32", 4096), ("ball3d", 256, "bf16", 2048), ("push", 64, "f32", 1024)])
def test_native_epoch_loop_matches_per_minibatch_calls(task, hidden, mfma, batch, monkeypatch):
    """tma_ppo_train_epoch_local (one call per epoch) against the same epoch issued minibatch by minibatch through tma_ppo_epoch_prepare +
    tma_ppo_minibatch_grad + tma_ppo_adam_step_local: bit-identical parameters, derived images and optimizer state; deterministic.
    (batch 256 / 1024 / 2048 on 64-wide nets also exercise the small-minibatch gradient kernel: 4-wave blocks, one tile per wave.)"""
    from three_mlagents_amd import _lib
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    if task == "push":
        monkeypatch.setenv("TMA_NO_PACKED", "1")  # one case gathers from the planes (no sample records, tma_rollout.packed = NULL)

    def build():
        env = make_vector_env(task, n_envs=64, seed=4)
        m = PPO("MlpPolicy", env, n_steps=64, batch_size=batch, n_epochs=2, seed=4, policy_kwargs={"net_arch": [hidden, hidden], "mfma_dtype": mfma})
        m.collect_rollouts()
        return env, m

    env_a, a = build()
    a.train()  # native epoch loop
    env_b, b = build()
    assert torch.equal(a.buf["advantages"], b.buf["advantages"]) and torch.equal(a.buf["obs"], b.buf["obs"])
    L, total = _lib.lib(), 64 * 64
    perm_seed = (b.seed * 2654435761 + 12345) & 0xFFFFFFFF
    step = 0
    if b._packed is not None:  # the view carries sample records: fill them, as train() does once per rollout
        _lib.check(L.tma_ppo_pack_samples(C.byref(b._rollout_view), C.byref(b.policy.dims), _lib.ptr(b._packed), b._stream()))
    for epoch in range(2):
        prepared = batch >= 256
        if prepared:
            ep = _lib.Minibatch(None, perm_seed, epoch, 0, total, 0)
            _lib.check(L.tma_ppo_epoch_prepare(C.byref(b._rollout_view), C.byref(ep), batch, C.byref(b.policy.dims), _lib.ptr(b.workspace), b._stream()))
        for start in range(0, total, batch):
            cnt = min(batch, total - start)
            mb = _lib.Minibatch(None, perm_seed, epoch, start, cnt, batch if prepared else 0)
            _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(b.policy.params), C.byref(b.policy.dims), C.byref(b._rollout_view), C.byref(mb), C.byref(b._hp),
                                                _lib.ptr(b.grad), _lib.ptr(b.workspace), b._stream()))
            step += 1
            _lib.check(L.tma_ppo_adam_step_local(_lib.ptr(b.policy.params), _lib.ptr(b.grad), _lib.ptr(b.exp_avg), _lib.ptr(b.exp_avg_sq),
                                                 C.byref(b.policy.dims), step, b.learning_rate, 0.9, 0.999, 1e-5, b.max_grad_norm, _lib.ptr(b.workspace),
                                                 b._stream(), cnt))
    pa, pb = a.policy.params.cpu(), b.policy.params.cpu()  # trainable region AND every derived copy / image
    assert a._adam_step == step and torch.isfinite(pa).all()
    assert torch.equal(pa, pb) and torch.equal(a.exp_avg.cpu(), b.exp_avg.cpu()) and torch.equal(a.exp_avg_sq.cpu(), b.exp_avg_sq.cpu())
    sa, sb = a.pop_train_stats(), b.pop_train_stats()
    assert sa["train/n_samples"] == sb["train/n_samples"] == 2 * total and abs(sa["train/approx_kl"] - sb["train/approx_kl"]) < 1e-6
    env_c, c = build()
    c.train()
    assert torch.equal(c.policy.params.cpu(), pa)  # run-to-run bitwise reproducible
    for e in (env_a, env_b, env_c):
        e.close()


@pytest.mark.parametrize("D,A", [(4, 5), (4, 12), (6, 5), (6, 14), (9, 7), (16, 16), (3, 2)])
def test_persistent_epoch_kernel_equals_per_minibatch_launches(D, A, monkeypatch):
    """The persistent epoch kernel (csrc/tma_h64p.hip: the reference's literal batch_size = 256 as one launch per epoch) against the same
    epochs issued as per-minibatch launches (TMA_NO_PERSIST=1), both through tma_ppo_train_epoch_local on synthetic rollouts whose widths
    select every instantiation of the kernel.  Gradient sums run in the same order in both paths, Adam is one shared routine and the clip
    norm differs only in its f64 summation order: parameters, derived images and moments must agree to the last bit or two; statistics and
    the reported norm to 1e-9 relative.  Then the first epoch against the torch restatement of SB3's loop (oracle), 1e-5."""
    from three_mlagents_amd import _lib

    T, N, B, H = 24, 64, 256, 64
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    res = []
    for persist in (True, False):
        if persist:
            monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
        else:
            monkeypatch.setenv("TMA_NO_PERSIST", "1")
        pol, sd = _policy(D, H, A, False)
        obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, False, T, N)
        d = {k: v.to(dev).contiguous() for k, v in dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret).items()}
        rv = _lib.Rollout(_lib.ptr(d["obs"]), _lib.ptr(d["actions"]), _lib.ptr(d["old_lp"]), _lib.ptr(d["adv"]), _lib.ptr(d["ret"]), T, N)
        hpar = _lib.PPOHParams(HP["clip_range"], HP["ent_coef"], HP["vf_coef"], 1)
        grad = torch.zeros(pol.n_trainable, device=dev)
        m, v = torch.zeros(pol.n_trainable, device=dev), torch.zeros(pol.n_trainable, device=dev)
        ws = torch.zeros(int(L.tma_ppo_workspace_bytes(C.byref(pol.dims))), dtype=torch.uint8, device=dev)
        n_mb, step, snaps = T * N // B, 1, []
        for epoch in range(3):
            _lib.check(L.tma_ppo_train_epoch_local(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rv), 77, epoch, B, C.byref(hpar), _lib.ptr(grad),
                                                   _lib.ptr(m), _lib.ptr(v), step, 3e-4, 0.9, 0.999, 1e-5, 0.5, _lib.ptr(ws), _lib.stream_ptr()))
            step += n_mb
            if epoch == 0:
                snaps.append({k: t.clone() for k, t in pol.state_dict().items()})
        out = (C.c_double * 8)()
        _lib.check(L.tma_ppo_pop_stats(_lib.ptr(ws), out, _lib.stream_ptr()))
        after = pol.params.clone()
        _lib.check(L.tma_policy_sync(_lib.ptr(pol.params), C.byref(pol.dims), _lib.stream_ptr()))
        assert torch.equal(after, pol.params)  # every derived copy / image is what a full refresh of the trainable region rebuilds
        assert float(grad.abs().max()) == 0.0
        res.append((after.cpu(), m.cpu(), v.cpu(), list(out), snaps[0], sd, (obs, actions, old_lp, adv, ret)))
    (p0, m0, v0, s0, snap0, sd0, roll), (p1, m1, v1, s1, _, _, _) = res
    assert torch.isfinite(p0).all() and s0[5] == s1[5] == 3 * T * N
    assert torch.allclose(p0, p1, rtol=0, atol=2e-7) and torch.allclose(m0, m1, rtol=1e-5, atol=1e-9) and torch.allclose(v0, v1, rtol=1e-5, atol=1e-12)
    for q in (0, 1, 2, 3, 4, 6, 7):
        assert abs(s0[q] - s1[q]) <= 1e-6 * max(1.0, abs(s1[q])), (q, s0[q], s1[q])
    # one epoch of SB3's loop on the CPU restatement (oracle/sb3_ref.RefTrainer: ppo_loss, clip_grad_norm_, torch.optim.Adam), minibatch by
    # minibatch in the order of the on-device permutation (tma_ppo_permutation)
    obs, actions, old_lp, adv, ret = roll
    idx_np = np.zeros(T * N, dtype=np.int64)
    _lib.check(L.tma_ppo_permutation(77, 0, T * N, idx_np.ctypes.data_as(C.c_void_p)))
    assert sorted(idx_np.tolist()) == list(range(T * N))
    perm = torch.from_numpy(idx_np)
    tr = sb3_ref.RefTrainer(sd0, lr=3e-4, max_grad_norm=0.5)
    flat = [_flatten_env_major(x, T, N) for x in (obs, actions, old_lp, adv, ret)]
    for start in range(0, T * N, B):
        rows = perm[start:start + B]
        tr.step(*[x[rows] for x in flat], **HP)
    for k in snap0:
        ref = tr.sd[k].detach()
        assert torch.allclose(snap0[k].cpu(), ref, rtol=0, atol=2e-5), (k, float((snap0[k].cpu() - ref).abs().max()))


def test_persistent_epoch_kernel_falls_back_to_launches_when_it_cannot_run(monkeypatch, capfd):
    """The persistent epoch kernel needs eight workgroups co-resident on one XCD.  When a launch cannot place / synchronise them it commits
    nothing; tma_ppo_train_epoch_local then runs THAT epoch through the per-minibatch launches and counts the event -- training goes on with
    the results of the launch path, bit for bit.  Failure is forced through the test hook TMA_PERSIST_FORCE_FAIL (the launch finds its abort
    word set)."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(mode):
        monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
        monkeypatch.delenv("TMA_PERSIST_FORCE_FAIL", raising=False)
        if mode == "launches":
            monkeypatch.setenv("TMA_NO_PERSIST", "1")
        elif mode == "forced_failure":
            monkeypatch.setenv("TMA_PERSIST_FORCE_FAIL", "1")
        elif mode == "failure_after_commit":  # the launch runs and commits EVERYTHING, then is declared failed: the worst case of a late abort
            monkeypatch.setenv("TMA_PERSIST_FORCE_FAIL", "late")
        env = make_vector_env("gridworld", n_envs=256, seed=4)
        m = PPO("MlpPolicy", env, n_steps=64, batch_size=256, n_epochs=3, seed=4, policy_kwargs={"net_arch": [64, 64]})
        m.collect_rollouts()
        m.train()
        st = m.pop_train_stats()  # (does not raise: the failure was handled where it happened)
        out = (m.policy.params.cpu(), m.exp_avg.cpu(), m.exp_avg_sq.cpu(), st, m._adam_step, getattr(m, "persist_fallbacks", 0))
        env.close()
        return out

    p_l, m_l, v_l, s_l, n_l, f_l = run("launches")
    p_f, m_f, v_f, s_f, n_f, f_f = run("forced_failure")
    p_p, m_p, v_p, s_p, n_p, f_p = run("persistent")
    p_c, m_c, v_c, s_c, n_c, f_c = run("failure_after_commit")
    #  a block may raise the abort word after another has committed: the epoch call snapshots parameters / moments / statistic slots
    # before the launch and restores them before the fallback, so even a fully committed launch that is declared failed changes nothing
    assert n_c == 3 * 64 and f_c == 3 and torch.equal(p_c, p_l) and torch.equal(m_c, m_l) and torch.equal(v_c, v_l)
    for k in ("train/policy_gradient_loss", "train/value_loss", "train/approx_kl", "train/n_samples"):
        assert s_c[k] == s_l[k], k
    assert n_l == n_f == n_p == 3 * 64 and (f_l, f_f, f_p) == (0, 3, 0)  # every one of the three epochs fell back, and was counted
    assert s_f["train/persist_fallbacks"] == 3.0 and "train/persist_fallbacks" not in s_p
    assert torch.equal(p_f, p_l) and torch.equal(m_f, m_l) and torch.equal(v_f, v_l)  # the fallback IS the launch path
    for k in ("train/policy_gradient_loss", "train/value_loss", "train/approx_kl", "train/n_samples"):
        assert s_f[k] == s_l[k], k
    assert torch.allclose(p_p, p_l, rtol=0, atol=1e-6)  # and the persistent kernel itself still runs when it can
    # (the library also says so on stderr, once per process: not asserted here -- an earlier fallback in the same process would have used it up)


@pytest.mark.parametrize("task,n_envs,n_steps,batch", [("gridworld", 256, 64, 1024), ("gridworld", 256, 64, 768), ("gridworld", 512, 128, 8192),
                                                     ("gridworld", 128, 64, 512), ("gridworld", 300, 10, 1000)])
def test_optimizer_step_folded_into_the_next_gradient_launch_is_bit_identical(monkeypatch, task, n_envs, n_steps, batch):
    """H = 64 fast path, minibatches of >= 256 samples: tma_ppo_train_epoch_local runs the clip + Adam step of minibatch k in the prologue of
    gradient launch k + 1 (every workgroup redoes it for its net and builds its LDS weight image from the results) instead of a launch of its
    own.  Same routine, same inputs: parameters (derived copies and images included), both Adam moments and the statistics equal the
    one-optimizer-launch-per-minibatch sequence (TMA_NO_ADAM_FOLD) bit for bit -- ragged last minibatches (768: 256 left over; 1000: 3 full
    + none left; both directions of the state ping-pong) included."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(fold):
        monkeypatch.setenv("TMA_NO_PERSIST", "1")
        if fold:
            monkeypatch.delenv("TMA_NO_ADAM_FOLD", raising=False)
        else:
            monkeypatch.setenv("TMA_NO_ADAM_FOLD", "1")
        env = make_vector_env(task, n_envs=n_envs, seed=11)
        m = PPO("MlpPolicy", env, n_steps=n_steps, batch_size=batch, n_epochs=3, seed=11, policy_kwargs={"net_arch": [64, 64]})
        for _ in range(2):
            m.collect_rollouts()
            m.train()
        st = m.pop_train_stats()
        out = (m.policy.params.cpu(), m.exp_avg.cpu(), m.exp_avg_sq.cpu(), st, m._adam_step, m.grad.cpu() if hasattr(m, "grad") else None)
        env.close()
        return out

    p0, m0, v0, s0, n0, g0 = run(False)
    p1, m1, v1, s1, n1, g1 = run(True)
    assert n0 == n1 == 2 * 3 * -(-(n_envs * n_steps) // batch)
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    if g0 is not None:
        assert torch.equal(g0, g1) and float(g1.abs().max()) == 0.0  # the gradient buffer is left zeroed either way
    for k in s0:
        assert s0[k] == s1[k], (k, s0[k], s1[k])


@pytest.mark.parametrize("task,hidden,n_envs,n_steps,batch", [("gridworld", 64, 256, 64, 1024), ("gridworld", 64, 256, 64, 768), ("gridworld", 64, 300, 10, 1000),
                                                            ("gridworld", 64, 64, 16, 256), ("ball3d", 256, 64, 32, 512)])
def test_native_data_parallel_epoch_equals_the_single_gpu_epoch(monkeypatch, task, hidden, n_envs, n_steps, batch):
    """tma_ppo_train_epoch_dp (the minibatch loop of a data-parallel rank: gradient, all-reduce callback, sum of squares + optimizer step on
    grad / world) at world size 1 (TMA_DP_PATH=1: the callback is a no-op) against the single-GPU epoch call, with and without the
    optimizer step folded into the next gradient launch there: the same bits from all three."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(mode):
        monkeypatch.setenv("TMA_NO_PERSIST", "1")
        for k in ("TMA_DP_PATH", "TMA_NO_ADAM_FOLD"):
            monkeypatch.delenv(k, raising=False)
        if mode in ("dp", "dp_unfolded"):
            monkeypatch.setenv("TMA_DP_PATH", "1")
        if mode == "dp_unfolded":
            monkeypatch.setenv("TMA_NO_ADAM_FOLD", "1")
        env = make_vector_env(task, n_envs=n_envs, seed=13)
        m = PPO("MlpPolicy", env, n_steps=n_steps, batch_size=batch, n_epochs=3, seed=13, policy_kwargs={"net_arch": [hidden, hidden]})
        for _ in range(2):
            m.collect_rollouts()
            m.train()
        st = m.pop_train_stats()
        out = (m.policy.params.cpu(), m.exp_avg.cpu(), m.exp_avg_sq.cpu(), st, m._adam_step, m.grad.cpu())
        env.close()
        return out

    ref = run("local")
    for mode in ("dp", "dp_unfolded"):
        got = run(mode)
        assert got[4] == ref[4] == 2 * 3 * -(-(n_envs * n_steps) // batch)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), mode
        assert float(got[5].abs().max()) == 0.0
        for k in ref[3]:
            assert got[3][k] == ref[3][k], (mode, k)


def test_data_parallel_epoch_reports_a_failing_collective(monkeypatch):
    """An exception raised by the all-reduce inside PPO.train's callback does not unwind through the native loop: the callback returns
    non-zero, tma_ppo_train_epoch_dp stops with an error, and train() re-raises the ORIGINAL exception."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    monkeypatch.setenv("TMA_DP_PATH", "1")
    env = make_vector_env("gridworld", n_envs=64, seed=3)
    m = PPO("MlpPolicy", env, n_steps=16, batch_size=256, n_epochs=1, seed=3, policy_kwargs={"net_arch": [64, 64]})
    m.collect_rollouts()
    m.world_size = 2  # (pretend: the callback then calls the collective)

    class Boom(RuntimeError):
        pass

    def failing(tensor, key):
        raise Boom("link down")

    monkeypatch.setattr(m, "_timed_all_reduce", failing)
    monkeypatch.setattr(m, "normalize_advantage", False)
    with pytest.raises(Boom, match="link down"):
        m.train()
    env.close()


def test_persistent_epoch_kernel_long_epoch_stays_with_the_launch_path(monkeypatch):
    """1024 optimizer steps in one persistent launch (GridWorld rollout of 1024 envs x 256 steps, the reference's batch_size = 256) against
    the same epoch as per-minibatch launches: the two paths differ only in the f64 summation order of the clip norm, so after a thousand
    dependent Adam steps the parameters still agree to 1e-6 and the loss statistics to 1e-6 relative; the persistent path is run-to-run
    bit-identical."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(persist):
        if persist:
            monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
        else:
            monkeypatch.setenv("TMA_NO_PERSIST", "1")
        env = make_vector_env("gridworld", n_envs=1024, seed=9)
        m = PPO("MlpPolicy", env, n_steps=256, batch_size=256, n_epochs=1, seed=9, policy_kwargs={"net_arch": [64, 64]})
        m.collect_rollouts()
        m.train()
        st = m.pop_train_stats()
        out = (m.policy.params.cpu(), m.exp_avg.cpu(), m.exp_avg_sq.cpu(), st, m._adam_step)
        env.close()
        return out

    p0, m0, v0, s0, n0 = run(True)
    p1, m1, v1, s1, n1 = run(False)
    p2, m2, v2, s2, n2 = run(True)
    assert n0 == n1 == 1024 and torch.isfinite(p0).all()
    assert torch.equal(p0, p2) and torch.equal(m0, m2) and torch.equal(v0, v2)
    assert torch.allclose(p0, p1, rtol=0, atol=1e-6), float((p0 - p1).abs().max())
    assert torch.allclose(m0, m1, rtol=1e-4, atol=1e-8) and torch.allclose(v0, v1, rtol=1e-4, atol=1e-10)
    for k in ("train/policy_gradient_loss", "train/value_loss", "train/entropy_loss", "train/approx_kl", "train/clip_fraction", "train/n_samples"):
        assert abs(s0[k] - s1[k]) <= 1e-6 * max(1.0, abs(s1[k])), (k, s0[k], s1[k])


@pytest.mark.parametrize("D,A,B", [(4, 5, 512), (6, 5, 1024), (7, 3, 512), (8, 9, 768), (3, 2, 256)])
def test_packed_sample_records_change_nothing_but_the_traffic(D, A, B, monkeypatch):
    """tma_rollout.packed (tma_ppo_pack_samples): the H = 64 gradient kernel reads one record per sample instead of gathering from five
    planes.  Same values, same summation orders: parameters, moments and statistics after three epochs must be IDENTICAL with and without the
    records (per-minibatch launches: the persistent batch-256 kernel does not read them), ragged last minibatch included."""
    from three_mlagents_amd import _lib

    monkeypatch.setenv("TMA_NO_PERSIST", "1")
    T, N, H = 25, 72, 64  # 1800 samples: not a multiple of any of the batch sizes
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    res = []
    for use_packed in (False, True):
        pol, sd = _policy(D, H, A, False)
        obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, False, T, N)
        d = {k: v.to(dev).contiguous() for k, v in dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret).items()}
        n_packed = int(L.tma_ppo_packed_floats(C.byref(pol.dims), T, N))
        assert n_packed == T * N * (((D + 3) // 4) * 4 + 4)
        packed = torch.full((n_packed,), float("nan"), device=dev) if use_packed else None
        rv = _lib.Rollout(_lib.ptr(d["obs"]), _lib.ptr(d["actions"]), _lib.ptr(d["old_lp"]), _lib.ptr(d["adv"]), _lib.ptr(d["ret"]), T, N, _lib.ptr(packed))
        if use_packed:
            _lib.check(L.tma_ppo_pack_samples(C.byref(rv), C.byref(pol.dims), _lib.ptr(packed), _lib.stream_ptr()))
            rec = packed.view(T * N, -1).cpu()
            xs = rec.shape[1] - 4
            assert torch.equal(rec[:, :D], obs.reshape(T * N, D)) and float(rec[:, D:xs].abs().sum()) == 0.0
            assert torch.equal(rec[:, xs], old_lp.reshape(-1)) and torch.equal(rec[:, xs + 1], adv.reshape(-1)) and torch.equal(rec[:, xs + 3], ret.reshape(-1))
            assert torch.equal(rec[:, xs + 2].view(torch.int32), actions.reshape(-1).to(torch.int32))
        hpar = _lib.PPOHParams(HP["clip_range"], HP["ent_coef"], HP["vf_coef"], 1)
        grad = torch.zeros(pol.n_trainable, device=dev)
        m, v = torch.zeros(pol.n_trainable, device=dev), torch.zeros(pol.n_trainable, device=dev)
        ws = torch.zeros(int(L.tma_ppo_workspace_bytes(C.byref(pol.dims))), dtype=torch.uint8, device=dev)
        n_mb, step = (T * N + B - 1) // B, 1
        for epoch in range(3):
            _lib.check(L.tma_ppo_train_epoch_local(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rv), 91, epoch, B, C.byref(hpar), _lib.ptr(grad),
                                                   _lib.ptr(m), _lib.ptr(v), step, 3e-4, 0.9, 0.999, 1e-5, 0.5, _lib.ptr(ws), _lib.stream_ptr()))
            step += n_mb
        out = (C.c_double * 8)()
        _lib.check(L.tma_ppo_pop_stats(_lib.ptr(ws), out, _lib.stream_ptr()))
        res.append((pol.params.clone().cpu(), m.cpu(), v.cpu(), list(out)))
    (p0, m0, v0, s0), (p1, m1, v1, s1) = res
    assert torch.isfinite(p0).all() and s0[5] == 3 * T * N
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1) and s0 == s1


def test_packed_records_are_refused_for_shapes_without_them():
    from three_mlagents_amd import _lib

    L = _lib.lib()
    for D, H, A in ((16, 64, 5), (4, 256, 5)):  # wide observations / the column-parallel kernels gather from the planes
        pol, _ = _policy(D, H, A, False)
        assert L.tma_ppo_packed_floats(C.byref(pol.dims), 8, 64) == 0
        rv = _lib.Rollout(None, None, None, None, None, 8, 64, None)
        buf = torch.zeros(16, device="cuda")
        assert L.tma_ppo_pack_samples(C.byref(rv), C.byref(pol.dims), _lib.ptr(buf), _lib.stream_ptr()) == _lib.TMA_ERR_INVALID


_DEFER_SCRIPT = r"""
import ctypes as C, sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import test_ppo_gpu as t
out = {{}}
# (>= 128 samples: below that the generic kernel with float atomics runs; 105 / 8 = the reference's `ant` task, 60: another width between 32 and 112 --
#  round 6: half groups with dW1 in registers, against the runtime-width kernel with dW1 accumulated in the slab)
for (D, A, cont, B) in [(6, 5, False, 256), (21, 3, False, 1000), (8, 2, True, 512), (4, 5, False, 200), (105, 8, True, 256), (105, 8, True, 700), (60, 4, False, 300)]:
    pol, sd = t._policy(D, 256, A, cont)
    T, N = 16, 80
    obs, actions, old_lp, adv, ret = t._rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    g1, st, _ = t._hip_grad(pol, bufs, T, N, perm, 11, B, t.HP)
    g2, _, _ = t._hip_grad(pol, bufs, T, N, perm, 11, B, t.HP)
    assert torch.equal(g1, g2)
    out[(D, A, cont, B)] = (g1.cpu(), st)
torch.save(out, sys.argv[1])
"""


def test_small_wide_minibatch_deferred_dw2_equals_the_slab_path(tmp_path):
    """f32 256-wide policies, minibatches <= 1024 samples: dW2 from the follow-up GEMM launch (wide_small_reduce_kernel, default) against
    per-block slabs + slab_reduce_kernel (TMA_NO_DEFER_W2=1) -- the same gradient up to the summation order of the sample dimension, the
    same loss statistics bit for bit, each path deterministic; Discrete and Box heads, ragged sizes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "defer.py"
    script.write_text(_DEFER_SCRIPT.format(root=root, tests=os.path.join(root, "tests")))
    res = []
    for env_extra in ({}, {"TMA_NO_DEFER_W2": "1"}):
        env = dict(os.environ, **env_extra)
        env.pop("TMA_NO_DEFER_W2", None) if not env_extra else None
        out = tmp_path / f"g{len(res)}.pt"
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(out))
    for key in res[0]:
        (ga, sa), (gb, sb) = res[0][key], res[1][key]
        scale = float(gb.abs().max())
        assert float((ga - gb).abs().max()) <= 2e-6 * max(scale, 1.0), key
        D, A, cont, B = key
        if D <= 32:
            assert sa == sb, key
        else:  # (eight waves against four: the head's split-K partial sums are added in another order, so log-probs differ in their last bits)
            assert all(abs(x - y) <= 1e-5 * max(1.0, abs(y)) for x, y in zip(sa, sb)), (key, sa, sb)
        w2 = slice(D * 256 + 256, D * 256 + 256 + 65536)
        assert float(ga[w2].abs().max()) > 0 and not torch.equal(ga[w2], gb[w2]) or B <= 16, key  # (the W2 block really took another route)
