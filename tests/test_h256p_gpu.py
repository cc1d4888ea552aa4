"""GPU tests of the persistent epoch kernel for the reference's default 256 x 256 policy at its literal batch_size = 256
(three-mlagents_amd/csrc/tma_h256p.hip; /root/reference/backend/mlagents/training.py:363-365,379).

The kernel is column-parallel over 2 x 32 workgroups and sums in its own (fixed) order, so it agrees with the per-minibatch launches to
rounding, not to the bit: tolerances are written next to each check.  Both paths are also held to the torch restatement of SB3's loop
(oracle/sb3_ref.RefTrainer, "parity unpinned" boundary -- DESIGN.md section 2)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sb3_ref
from test_ppo_gpu import HP, _flatten_env_major, _policy, _rollout

pytestmark = pytest.mark.gpu


def _epochs(D, A, T, N, n_epochs, persist, monkeypatch, seed=5):
    from three_mlagents_amd import _lib

    B, H = 256, 256
    if persist:
        monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
    else:
        monkeypatch.setenv("TMA_NO_PERSIST", "1")
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    pol, sd = _policy(D, H, A, False, seed=seed)
    roll = _rollout(pol, sd, D, A, False, T, N)
    obs, actions, old_lp, adv, ret = roll
    d = {k: v.to(dev).contiguous() for k, v in dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret).items()}
    rv = _lib.Rollout(_lib.ptr(d["obs"]), _lib.ptr(d["actions"]), _lib.ptr(d["old_lp"]), _lib.ptr(d["adv"]), _lib.ptr(d["ret"]), T, N)
    hpar = _lib.PPOHParams(HP["clip_range"], HP["ent_coef"], HP["vf_coef"], 1)
    grad = torch.zeros(pol.n_trainable, device=dev)
    m, v = torch.zeros(pol.n_trainable, device=dev), torch.zeros(pol.n_trainable, device=dev)
    ws = torch.zeros(int(L.tma_ppo_workspace_bytes(C.byref(pol.dims))), dtype=torch.uint8, device=dev)
    n_mb, step, snaps = T * N // B, 1, []
    for epoch in range(n_epochs):
        _lib.check(L.tma_ppo_train_epoch_local(_lib.ptr(pol.params), C.byref(pol.dims), C.byref(rv), 77, epoch, B, C.byref(hpar), _lib.ptr(grad),
                                               _lib.ptr(m), _lib.ptr(v), step, 3e-4, 0.9, 0.999, 1e-5, 0.5, _lib.ptr(ws), _lib.stream_ptr()))
        step += n_mb
        snaps.append({k: t.clone() for k, t in pol.state_dict().items()})
    out = (C.c_double * 8)()
    _lib.check(L.tma_ppo_pop_stats(_lib.ptr(ws), out, _lib.stream_ptr()))
    fallbacks = C.c_int64(0)
    _lib.check(L.tma_ppo_persist_fallbacks(_lib.ptr(ws), C.byref(fallbacks), _lib.stream_ptr()))
    after = pol.params.clone()
    _lib.check(L.tma_policy_sync(_lib.ptr(pol.params), C.byref(pol.dims), _lib.stream_ptr()))
    assert torch.equal(after, pol.params)  # every derived copy / image is what a full refresh of the trainable region rebuilds
    assert float(grad.abs().max()) == 0.0
    return dict(p=after[:pol.n_trainable].cpu(), m=m.cpu(), v=v.cpu(), stats=list(out), snaps=snaps, sd=sd, roll=roll, fallbacks=fallbacks.value,
                n_trainable=pol.n_trainable)


# (D, A): GridWorld / Push 4 x 5, Ball3D 6 x 5, Basic 21 x 3 (two small-tensor slots per thread, two dW1 k-tiles), the widest shape 32 x 16
@pytest.mark.parametrize("D,A", [(4, 5), (6, 5), (21, 3), (32, 16), (1, 2)])
def test_h256p_epoch_equals_per_minibatch_launches_and_the_sb3_restatement(D, A, monkeypatch):
    T, N = 16, 64  # 1024 samples: four optimizer steps per epoch
    a = _epochs(D, A, T, N, 3, True, monkeypatch)
    b = _epochs(D, A, T, N, 3, False, monkeypatch)
    assert a["fallbacks"] == 0 and b["fallbacks"] == 0  # the persistent kernel ran (and did not give up on a wait)
    assert torch.isfinite(a["p"]).all() and a["stats"][5] == b["stats"][5] == 3 * T * N
    # twelve dependent optimizer steps apart, sums in different orders: parameters to 2e-6 absolute (lr 3e-4: one step moves a parameter by <= 3e-4)
    dp = float((a["p"] - b["p"]).abs().max())
    assert dp <= 2e-6, dp
    assert torch.allclose(a["m"], b["m"], rtol=1e-3, atol=1e-7), float((a["m"] - b["m"]).abs().max())
    assert torch.allclose(a["v"], b["v"], rtol=1e-3, atol=1e-10), float((a["v"] - b["v"]).abs().max())
    for q in (0, 1, 2, 3, 4, 6, 7):
        assert abs(a["stats"][q] - b["stats"][q]) <= 1e-5 * max(1.0, abs(b["stats"][q])), (q, a["stats"][q], b["stats"][q])
    # the first epoch against SB3's loop on the CPU (ppo_loss, clip_grad_norm_, torch.optim.Adam), in the order of the on-device permutation
    from three_mlagents_amd import _lib

    obs, actions, old_lp, adv, ret = a["roll"]
    idx_np = np.zeros(T * N, dtype=np.int64)
    _lib.check(_lib.lib().tma_ppo_permutation(77, 0, T * N, idx_np.ctypes.data_as(C.c_void_p)))
    perm = torch.from_numpy(idx_np)
    tr = sb3_ref.RefTrainer(a["sd"], lr=3e-4, max_grad_norm=0.5)
    flat = [_flatten_env_major(x, T, N) for x in (obs, actions, old_lp, adv, ret)]
    for start in range(0, T * N, 256):
        rows = perm[start:start + 256]
        tr.step(*[x[rows] for x in flat], **HP)
    for k in a["snaps"][0]:
        ref = tr.sd[k].detach()
        assert torch.allclose(a["snaps"][0][k].cpu(), ref, rtol=0, atol=2e-5), (k, float((a["snaps"][0][k].cpu() - ref).abs().max()))


def test_h256p_single_step_gradient_shows_in_the_moments(monkeypatch):
    """After ONE epoch of two optimizer steps from zero moments, exp_avg = 0.1 g2 + 0.09 g1 (clip-scaled gradients): the first-moment vector
    of the persistent kernel against the launch path's is a direct read-out of the gradients both computed -- every tensor of both nets, 1e-5
    relative to the largest entry of the tensor (different summation orders), so a mis-routed column slice or quarter cannot hide."""
    from three_mlagents_amd import _lib

    D, A, T, N = 6, 5, 8, 64
    a = _epochs(D, A, T, N, 1, True, monkeypatch)
    b = _epochs(D, A, T, N, 1, False, monkeypatch)
    assert a["fallbacks"] == 0
    from three_mlagents_amd.ppo import HipActorCriticPolicy

    pol = HipActorCriticPolicy(D, A, False, 256, torch.device("cuda", 0), seed=5)
    offs = (C.c_int32 * 13)()
    _lib.check(_lib.lib().tma_policy_param_offsets(C.byref(pol.dims), offs))
    bounds = list(offs)[:12] + [a["n_trainable"]]
    names = ["pW1", "pb1", "pW2", "pb2", "pW3", "pb3", "vW1", "vb1", "vW2", "vb2", "vW3", "vb3"]
    for i, name in enumerate(names):
        lo, hi = bounds[i], bounds[i + 1]
        ma, mb = a["m"][lo:hi], b["m"][lo:hi]
        scale = float(mb.abs().max())
        assert scale > 0, name
        assert float((ma - mb).abs().max()) <= 1e-5 * scale, (name, float((ma - mb).abs().max()), scale)


def test_h256p_is_run_to_run_bit_identical_and_falls_back_when_it_cannot_run(monkeypatch):
    D, A, T, N = 4, 5, 16, 64
    a = _epochs(D, A, T, N, 2, True, monkeypatch)
    b = _epochs(D, A, T, N, 2, True, monkeypatch)
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"])
    launches = _epochs(D, A, T, N, 2, False, monkeypatch)
    monkeypatch.setenv("TMA_PERSIST_FORCE_FAIL", "1")  # the launch finds its abort word set: commits nothing, the epoch runs as launches
    f = _epochs(D, A, T, N, 2, True, monkeypatch)
    assert f["fallbacks"] == 2
    assert torch.equal(f["p"], launches["p"]) and torch.equal(f["m"], launches["m"]) and torch.equal(f["v"], launches["v"])
    monkeypatch.setenv("TMA_PERSIST_FORCE_FAIL", "late")  # the launch commits EVERYTHING and is then declared failed: the snapshot is restored first
    g = _epochs(D, A, T, N, 2, True, monkeypatch)
    assert g["fallbacks"] == 2
    assert torch.equal(g["p"], launches["p"]) and torch.equal(g["m"], launches["m"]) and torch.equal(g["v"], launches["v"])
    for q in (0, 1, 2, 3, 4, 5):
        assert g["stats"][q] == launches["stats"][q], q


def test_h256p_long_epoch_through_ppo(monkeypatch):
    """256 optimizer steps in one launch through PPO.train (GridWorld, 256 envs x 256 steps, the reference's net and batch size) against the
    per-minibatch launches: parameters to 2e-5 after 256 dependent Adam steps in different summation orders, statistics to 1e-4 relative."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(persist):
        if persist:
            monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
        else:
            monkeypatch.setenv("TMA_NO_PERSIST", "1")
        env = make_vector_env("gridworld", n_envs=256, seed=9)
        m = PPO("MlpPolicy", env, n_steps=256, batch_size=256, n_epochs=1, seed=9, policy_kwargs={"net_arch": [256, 256]})
        m.collect_rollouts()
        m.train()
        st = m.pop_train_stats()
        out = (m.policy.params.cpu(), m.exp_avg.cpu(), m.exp_avg_sq.cpu(), st, m._adam_step)
        env.close()
        return out

    p0, m0, v0, s0, n0 = run(True)
    p1, m1, v1, s1, n1 = run(False)
    assert n0 == n1 == 256 and torch.isfinite(p0).all() and "train/persist_fallbacks" not in s0
    assert torch.allclose(p0, p1, rtol=0, atol=2e-5), float((p0 - p1).abs().max())
    for k in ("train/policy_gradient_loss", "train/value_loss", "train/entropy_loss", "train/approx_kl", "train/n_samples"):
        assert abs(s0[k] - s1[k]) <= 1e-4 * max(1.0, abs(s1[k])), (k, s0[k], s1[k])


@pytest.mark.parametrize("hidden", [64, 256])
def test_all_epochs_in_one_persistent_launch_equal_a_launch_per_epoch(hidden, monkeypatch):
    """tma_ppo_train_epochs_local: where a persistent epoch kernel takes the shape and the epochs' sample offsets fit the workspace, the
    n_epochs of PPO.train run as ONE launch (the reference's own 8-env schedule: 32 optimizer steps an epoch) -- weights, moments and the
    step count stay on the chip between epochs.  Same kernel, same permutations, same Adam table: bit-identical to one launch per epoch
    (TMA_EPOCH_PER_CALL=1), for the 64 x 64 kernel (tma_h64p.hip) and the 256 x 256 one (tma_h256p.hip); and the forced failure of the
    one launch hands every epoch back (to per-epoch launches that fail the same way, then to the per-minibatch launches)."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    task = "basic" if hidden == 256 else "gridworld"  # (Basic's 21 observations are beyond the 64-wide persistent kernel's 16)

    def run(per_call, force_fail=False):
        monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
        monkeypatch.delenv("TMA_PERSIST_FORCE_FAIL", raising=False)
        if per_call:
            monkeypatch.setenv("TMA_EPOCH_PER_CALL", "1")
        else:
            monkeypatch.delenv("TMA_EPOCH_PER_CALL", raising=False)
        if force_fail:
            monkeypatch.setenv("TMA_PERSIST_FORCE_FAIL", "1")
        env = make_vector_env(task, n_envs=8, seed=3)
        m = PPO("MlpPolicy", env, n_steps=128, batch_size=256, n_epochs=5, seed=3, policy_kwargs={"net_arch": [hidden, hidden]})
        for _ in range(2):
            m.collect_rollouts()
            m.train()
        st = m.pop_train_stats()
        out = (m.policy.params.cpu(), m.exp_avg.cpu(), m.exp_avg_sq.cpu(), st, m._adam_step)
        env.close()
        return out

    p0, m0, v0, s0, n0 = run(False)
    p1, m1, v1, s1, n1 = run(True)
    assert n0 == n1 == 2 * 5 * 4 and "train/persist_fallbacks" not in s0 and "train/persist_fallbacks" not in s1
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    for k in ("train/policy_gradient_loss", "train/value_loss", "train/entropy_loss", "train/approx_kl", "train/n_samples"):
        assert abs(s0[k] - s1[k]) <= 1e-12 * max(1.0, abs(s1[k])), (k, s0[k], s1[k])  # (the same per-step sums, folded per launch instead of per epoch)
    p2, m2, v2, s2, n2 = run(False, force_fail=True)
    monkeypatch.setenv("TMA_NO_PERSIST", "1")
    monkeypatch.delenv("TMA_PERSIST_FORCE_FAIL", raising=False)
    env = make_vector_env(task, n_envs=8, seed=3)
    ref = PPO("MlpPolicy", env, n_steps=128, batch_size=256, n_epochs=5, seed=3, policy_kwargs={"net_arch": [hidden, hidden]})
    for _ in range(2):
        ref.collect_rollouts()
        ref.train()
    assert n2 == 40 and s2["train/persist_fallbacks"] == 10.0  # (two train() calls x five epochs, each epoch counted once)
    assert torch.equal(p2, ref.policy.params.cpu()) and torch.equal(m2, ref.exp_avg.cpu())
    env.close()


def test_h256p_takes_bf16x3_policies_at_the_literal_batch(monkeypatch):
    """mfma_dtype "bf16x3" (opt-in: the f32 256 x 256 update as a three-term bf16 split) only changes minibatches of >= 4 096 samples; at the
    reference's literal batch_size = 256 such a policy runs the exact-f32 update -- the persistent kernel when it can, and its three-plane weight
    images are rebuilt behind the launch like every other derived copy (tma_policy_sync afterwards changes nothing)."""
    import ctypes as C

    from three_mlagents_amd import _lib
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(persist):
        if persist:
            monkeypatch.delenv("TMA_NO_PERSIST", raising=False)
        else:
            monkeypatch.setenv("TMA_NO_PERSIST", "1")
        env = make_vector_env("gridworld", n_envs=64, seed=2)
        m = PPO("MlpPolicy", env, n_steps=32, batch_size=256, n_epochs=2, seed=2, policy_kwargs={"net_arch": [256, 256], "mfma_dtype": "bf16x3"})
        m.collect_rollouts()
        m.train()
        st = m.pop_train_stats()
        after = m.policy.params.clone()
        _lib.check(_lib.lib().tma_policy_sync(_lib.ptr(m.policy.params), C.byref(m.policy.dims), _lib.stream_ptr()))
        assert torch.equal(after, m.policy.params)
        out = (after[: m.policy.n_trainable].cpu(), st)
        env.close()
        return out

    p0, s0 = run(True)
    p1, s1 = run(False)
    assert "train/persist_fallbacks" not in s0 and torch.isfinite(p0).all()
    assert torch.allclose(p0, p1, rtol=0, atol=2e-6), float((p0 - p1).abs().max())
