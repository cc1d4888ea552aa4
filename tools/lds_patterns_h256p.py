#!/usr/bin/env python3
"""LDS-array cycles of the access patterns of ppo_epoch_h256p_kernel, MEASURED (tools/lds_pattern_probe.hip) -- and of candidate layouts.
Build the probe first: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/bin/liblds_probe.so tools/lds_pattern_probe.hip"""
import ctypes as C, os, sys
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "liblds_probe.so"))
lib.lds_probe.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
KIND = {"r32": 0, "r64": 1, "r128": 2, "w32": 3, "w64": 4, "w128": 5}
IDEAL = {"r32": 2, "r64": 2, "r128": 4, "w32": 4, "w64": 6, "w128": 13}

def measure(kind, word_of_lane):
    offs = (C.c_int * 64)(*[4 * (word_of_lane(l) % 16384) for l in range(64)])
    out = C.c_double(0)
    assert lib.lds_probe(KIND[kind], offs, C.byref(out)) == 0
    return out.value

def show(name, kind, f):
    c = measure(kind, f)
    print(f"  {name:62s} {kind:5s} {c:6.2f} cycles  ({c / IDEAL[kind]:.2f} x conflict-free)")

H1_LD, W2S_LD, H2_LD, DZ2_LD, X_LD, DZ3_LD, W1_LD = 260, 80, 36, 48, 33, 17, 48
r16 = lambda l: l & 15
g = lambda l: l >> 4
print("reference streams")
show("contiguous b128", "r128", lambda l: 4 * l)
show("contiguous b64", "r64", lambda l: 2 * l)
show("contiguous b32", "r32", lambda l: l)
show("contiguous w32", "w32", lambda l: l)
show("contiguous w128", "w128", lambda l: 4 * l)
print("ppo_epoch_h256p_kernel as shipped (t = j = c = wave = 0 where they only shift the base)")
show("P1 x operand  Xs[(r16) X_LD + g]", "r32", lambda l: r16(l) * X_LD + g(l))
show("P1 w operand  W1s[g W1_LD + r16]", "r32", lambda l: g(l) * W1_LD + r16(l))
show("P1 h1 store   H1[(4 g + rr) H1_LD + r16]", "w32", lambda l: (4 * g(l)) * H1_LD + r16(l))
show("publish read  H1[(tid >> 3) H1_LD + 4 (tid & 7)]  (wave 0)", "r128", lambda l: (l >> 3) * H1_LD + 4 * (l & 7))
show("gather store  H1[(tid >> 3) H1_LD + 4 (tid & 7)]", "w128", lambda l: (l >> 3) * H1_LD + 4 * (l & 7))
show("P2 A  H1[r16 H1_LD + 4 g (+ 16 q)]", "r128", lambda l: r16(l) * H1_LD + 4 * g(l))
show("P2 B  W2s[g W2S_LD + 4 r16 (+ 8 u W2S_LD)]", "r128", lambda l: g(l) * W2S_LD + 4 * r16(l))
show("P2 h2 store  H2[(4 g + rr) H2_LD + r16]", "w32", lambda l: (4 * g(l)) * H2_LD + r16(l))
show("P3a A  H2[r16 H2_LD + 4 ks + g]", "r32", lambda l: r16(l) * H2_LD + g(l))
show("P4 dW3 A  H2[(4 sx + g) H2_LD + r16]", "r32", lambda l: g(l) * H2_LD + r16(l))
show("P4 dW3 B  DZ3[(4 sx + g) DZ3_LD + r16]", "r32", lambda l: g(l) * DZ3_LD + r16(l))
show("P4 dz2 A  DZ3[r16 DZ3_LD + 4 ns + g]", "r32", lambda l: r16(l) * DZ3_LD + g(l))
show("P4 dz2 store DZ2[(4 g + rr) DZ2_LD + r16]", "w32", lambda l: (4 * g(l)) * DZ2_LD + r16(l))
show("P5a A  H1[(4 sx + g) H1_LD + r16]", "r32", lambda l: g(l) * H1_LD + r16(l))
show("P5a B  DZ2[(4 sx + g) DZ2_LD + r16]", "r32", lambda l: g(l) * DZ2_LD + r16(l))
show("P5b A  DZ2[r16 DZ2_LD + 4 ks + g]", "r32", lambda l: r16(l) * DZ2_LD + g(l))
show("P5b B  W2s[(r16 >> 2) W2S_LD + 4 g + (r16 & 3)]", "r32", lambda l: (r16(l) >> 2) * W2S_LD + 4 * g(l) + (r16(l) & 3))
show("dz1 RMW read  H1[(4 g + rr) H1_LD + r16 (+ 16 nt)]", "r32", lambda l: (4 * g(l)) * H1_LD + r16(l))
show("P6 B  H1[(4 s + g) H1_LD + r16]", "r32", lambda l: g(l) * H1_LD + r16(l))
show("P6 A  Xs[(4 s + g) X_LD + r16]", "r32", lambda l: g(l) * X_LD + r16(l))
if len(sys.argv) > 1:
    print("candidates")
    for ld in (256, 260, 264, 272, 288):
        show(f"P2 A with H1_LD = {ld}", "r128", lambda l, ld=ld: r16(l) * ld + 4 * g(l))
        show(f"P5a A with H1_LD = {ld}", "r32", lambda l, ld=ld: g(l) * ld + r16(l))
        show(f"P1 store with H1_LD = {ld}", "w32", lambda l, ld=ld: 4 * g(l) * ld + r16(l))
    for ld in (64, 68, 72, 80, 96):
        show(f"P2 B with W2S_LD = {ld}", "r128", lambda l, ld=ld: g(l) * ld + 4 * r16(l))
        show(f"P5b B with W2S_LD = {ld}", "r32", lambda l, ld=ld: (r16(l) >> 2) * ld + 4 * g(l) + (r16(l) & 3))
