"""Debug helper: error of the bf16 forward vs the rounding-point emulation and vs the f32 reference, per config."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import sb3_ref
from test_bf16_gpu import _policies, _emulated_forward, BF_CONFIGS

for (D, H, A, cont) in BF_CONFIGS + [(32, 256, 5, False), (33, 256, 5, False), (64, 256, 5, False)]:
    pol, sd = _policies(D, H, A, cont)
    obs = torch.randn(133, D, generator=torch.Generator().manual_seed(1))
    out_emu, v_emu = _emulated_forward(sd, obs)
    out_ref, v_ref = sb3_ref.forward(sd, obs)
    a, v, lp = pol.act(obs.cuda(), deterministic=True)
    print(D, H, A, cont, "v-emu %.2e  v-f32 %.2e  emu-f32 %.2e" % ((v.cpu() - v_emu).abs().max(), (v.cpu() - v_ref).abs().max(), (v_emu - v_ref).abs().max()))
