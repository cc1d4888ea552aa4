"""mfma_dtype = "bf16x3" (round 5, opt-in): the f32 update of the reference's default 256 x 256 policy (backend/mlagents/training.py:363-365) on
the bf16 MFMA with every operand as three bf16 terms (csrc/tma_split3.h).  Held to the bound the exact-f32 kernels are held to against autograd
(2e-5 of the largest gradient entry: tests/test_ppo_gpu.py), and compared with the exact-f32 kernel on the same inputs; the weight planes the
optimizer maintains (scatter into three planes per Adam step) must equal a rebuild from the master weights; training must still learn."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import sb3_ref

pytestmark = pytest.mark.gpu

from tests.test_ppo_gpu import HP, _flatten_env_major, _hip_grad, _ref_grad_flat, _rollout  # noqa: E402


def _policy(D, H, A, dtype, seed=5):
    from three_mlagents_amd.ppo import HipActorCriticPolicy

    pol = HipActorCriticPolicy(D, A, False, H, torch.device("cuda", 0), seed=seed, mfma_dtype=dtype)
    sd = pol.state_dict()
    g = torch.Generator().manual_seed(seed)
    sd["action_net.weight"] = sd["action_net.weight"] * 40 + 0.05 * torch.randn(sd["action_net.weight"].shape, generator=g)
    sd["action_net.bias"] = 0.1 * torch.randn(sd["action_net.bias"].shape, generator=g)
    for k in list(sd):
        if k.endswith("bias") and k != "action_net.bias":
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    pol.load_state_dict(sd)
    return pol, sd


@pytest.mark.parametrize("D,A", [(6, 5), (4, 5), (21, 3)])
def test_split3_gradient_matches_autograd_and_the_exact_f32_kernel(D, A):
    H, T, N, B = 256, 64, 600, 33000
    pol, sd = _policy(D, H, A, "bf16x3")
    pol32, _ = _policy(D, H, A, "f32")
    obs, actions, old_lp, adv, ret = _rollout(pol32, sd, D, A, False, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    idx = perm[50:50 + B]
    f = lambda x: _flatten_env_major(x, T, N)[idx]  # noqa: E731
    stats_ref, grads_ref = sb3_ref.RefTrainer(sd).step(f(obs), f(actions), f(old_lp), f(adv), f(ret), **HP)
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    grad, st, _ = _hip_grad(pol, bufs, T, N, perm, 50, B, HP)
    grad2, _, _ = _hip_grad(pol, bufs, T, N, perm, 50, B, HP)
    assert torch.equal(grad, grad2)  # deterministic
    g32, st32, _ = _hip_grad(pol32, bufs, T, N, perm, 50, B, HP)
    ref = _ref_grad_flat(pol32, grads_ref)
    scale = ref.abs().max().item()
    err, err32 = (grad.cpu() - ref).abs().max().item(), (g32.cpu() - ref).abs().max().item()
    d = (grad - g32).abs().max().item()
    print(f"D={D}: split vs autograd {err / scale:.2e}, exact f32 vs autograd {err32 / scale:.2e}, split vs exact f32 {d / scale:.2e} (of max |g|)")
    assert err <= 2e-5 * max(scale, 1.0) + 1e-6, (err, scale)
    assert d <= 5e-6 * max(scale, 1.0), (d, scale)  # the split is an f32-class product: measured 3e-8 .. 2e-6
    assert st[5] == B and abs(st[0] / B - stats_ref["policy_loss"]) < 1e-5 and abs(st[1] / B - stats_ref["value_loss"]) < 1e-4
    assert abs(-st[2] / B - stats_ref["entropy_loss"]) < 1e-4 and abs(st[4] / B - stats_ref["clip_fraction"]) < 1e-6


def test_split3_small_minibatches_stay_on_the_exact_kernel():
    """Below 4096 samples the update runs the exact-f32 kernel (its half-group path is built for small minibatches): bit-identical to mfma_dtype f32."""
    D, A, H, T, N, B = 6, 5, 256, 16, 64, 512
    pol, sd = _policy(D, H, A, "bf16x3")
    pol32, _ = _policy(D, H, A, "f32")
    obs, actions, old_lp, adv, ret = _rollout(pol32, sd, D, A, False, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    bufs = dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret)
    g, _, _ = _hip_grad(pol, bufs, T, N, perm, 10, B, HP)
    g32, _, _ = _hip_grad(pol32, bufs, T, N, perm, 10, B, HP)
    assert torch.equal(g, g32)


def test_split3_planes_follow_the_optimizer_and_training_learns():
    from three_mlagents_amd import _lib
    from three_mlagents_amd.evaluation import evaluate_policy
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.vec_env import HipVecEnv

    env = HipVecEnv("gridworld", 4096, seed=1)
    model = PPO("MlpPolicy", env, n_steps=64, batch_size=32768, n_epochs=4, ent_coef=0.01, seed=1, policy_kwargs={"net_arch": [256, 256], "mfma_dtype": "bf16x3"})
    eval_env = HipVecEnv("gridworld", 16, seed=10_001)
    before, _ = evaluate_policy(model, eval_env, n_eval_episodes=32, deterministic=True)
    model.learn(4096 * 64 * 24)
    after, _ = evaluate_policy(model, eval_env, n_eval_episodes=32, deterministic=True)
    print(f"gridworld 256x256 bf16x3: deterministic eval {before:.3f} -> {after:.3f}; {model.logger_values}")
    assert after > 0.6 and after > before
    assert abs(model.logger_values["train/approx_kl"]) < 0.1
    # the three weight planes the optimizer scattered into, step after step, equal a rebuild from the master weights
    torch.cuda.synchronize()
    kept = model.policy.params.clone()
    _lib.check(_lib.lib().tma_policy_sync(_lib.ptr(model.policy.params), C.byref(model.policy.dims), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(kept.view(torch.int32), model.policy.params.view(torch.int32))
