"""Debug helper: per-segment error of the bf16 gradient vs the rounding-point emulation."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_bf16_gpu import _policies, _emulated_grad
from test_ppo_gpu import HP, _flatten_env_major, _hip_grad, _ref_grad_flat, _rollout

import ast
for (D, H, A, cont, B) in ast.literal_eval(sys.argv[1]):
    T, N = (64, 600) if B > 1000 else (16, 24)
    pol, sd = _policies(D, H, A, cont)
    obs, actions, old_lp, adv, ret = _rollout(pol, sd, D, A, cont, T, N)
    perm = torch.randperm(T * N, generator=torch.Generator().manual_seed(2))
    idx = perm[37:37 + B]
    f = lambda x: _flatten_env_major(x, T, N)[idx]
    ge, se = _emulated_grad(sd, f(obs), f(actions), f(old_lp), f(adv), f(ret), HP)
    grad, st, _ = _hip_grad(pol, dict(obs=obs, actions=actions, old_lp=old_lp, adv=adv, ret=ret), T, N, perm, 37, B, HP)
    emu, g = _ref_grad_flat(pol, ge), grad.cpu()
    print(D, H, A, cont, B, "total rel", float((g - emu).norm() / emu.norm()), "stats", [round(x / B, 5) for x in st[:5]], {k: round(v, 5) for k, v in se.items() if k != "log_prob"})
    segs = [(k, off, int(np.prod(shape))) for k, off, shape in pol._segments()]
    if cont:
        segs.append(("log_std", pol.offsets[12], pol.act_dim))
    for key, off, cnt in segs:
        r, x = emu[off:off + cnt], g[off:off + cnt]
        print("   %-40s rel %.4f  |ref| %.3e" % (key, float((x - r).norm() / max(r.norm().item(), 1e-9)), float(r.norm())))
