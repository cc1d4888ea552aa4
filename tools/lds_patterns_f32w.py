#!/usr/bin/env python3
"""Measured LDS-array cycles of the access patterns of ppo_grad_wide_kernel (the exact-f32 256-wide gradient kernel): row-major images with ld = H + 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_patterns_h256p import show, r16, g
print("ppo_grad_wide_kernel (f32, ld = 258)")
for ld in (258, 260, 264, 272, 273, 257, 265):
    print(f" ld = {ld}")
    show("chain A operand  h[r16 ld + 4 i + g]", "r32", lambda l, ld=ld: r16(l) * ld + g(l))
    show("weight-gradient operand  h[(4 s + g) ld + 16 kt + r16]", "r32", lambda l, ld=ld: g(l) * ld + r16(l))
    show("C-layout store  h[(4 g + r) ld + r16]", "w32", lambda l, ld=ld: 4 * g(l) * ld + r16(l))
