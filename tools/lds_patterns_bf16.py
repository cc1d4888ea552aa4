#!/usr/bin/env python3
"""Measured LDS-array cycles of the access patterns of ppo_grad_wide_bf_kernel<false, 2, 4, 1, 1, 0, 8> (64-row groups, eight waves; tools/lds_pattern_probe.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_patterns_h256p import show, r16, g  # (prints that kernel's table first when imported: harmless)

print("ppo_grad_wide_bf_kernel, MT = 4 (words = 4 bytes; bf16 element offsets halved)")
LDA = 272 // 2  # A1 / A2 rows: H + 16 bf16
LDX = 48 // 2
swz = lambda r: r & 7
t_off_w = lambda r, m: (r * 64 + 8 * ((m >> 3) ^ swz(r)) + (m & 7)) / 2  # word offset of T[r][m] (64 bf16 per row)
show("P2 / dh1 A  a_frag(A, lda, r16, ks, g)", "r128", lambda l: r16(l) * LDA + 4 * g(l))
show("P1 A  a_frag(Xa, ldx, r16, 0, g)", "r128", lambda l: r16(l) * LDX + 4 * g(l))
for kk in (0, 1):
    show(f"dW2 / dW3 / P6  t_frag(T, r16, kk = {kk}, g)", "r128", lambda l, kk=kk: int(t_off_w(r16(l), 8 * (4 * kk + g(l)))))
for mt in (0, 1, 2, 3):
    show(f"epilogue quad  t_quad(T, n = r16, mt = {mt}, g)  store", "w64", lambda l, mt=mt: int(t_off_w(r16(l), 16 * mt + 4 * g(l))))
    show(f"delta quad read  t_quad(T, n = r16, mt = {mt}, g)", "r64", lambda l, mt=mt: int(t_off_w(r16(l), 16 * mt + 4 * g(l))))
show("row-major store  A[(4 g + e) lda + r16]  (ds_write_b16: timed as b32 on the word)", "w32", lambda l: (4 * g(l)) * LDA + r16(l) // 2)
show("Z3a read  (16 mt + r16) ldz + 8 g  (ldz = 48 bf16)", "r128", lambda l: r16(l) * 24 + 4 * g(l))
show("head W3 fragment from LDS  (lane 8 bf16)", "r128", lambda l: 4 * l)
