#!/usr/bin/env python3
"""Time tma_gae_flags at the BASELINE shape (T = 1024, N = 4096) with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
T, N = 1024, 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
r, v = torch.randn(T, N, device=dev, generator=g), torch.randn(T, N, device=dev, generator=g)
te = (torch.rand(T, N, device=dev, generator=g) < 0.02).to(torch.uint8)
tr = (torch.rand(T, N, device=dev, generator=g) < 0.01).to(torch.uint8)
lv = torch.randn(N, device=dev, generator=g)
adv, ret = torch.empty_like(r), torch.empty_like(r)
L = _lib.lib()
def run():
    _lib.check(L.tma_gae_flags(_lib.ptr(r), _lib.ptr(v), _lib.ptr(te), _lib.ptr(tr), _lib.ptr(lv), 0.99, 0.95, T, N, _lib.ptr(adv), _lib.ptr(ret), _lib.stream_ptr()))
for _ in range(3):
    run()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
byt = T * N * (4 + 4 + 2 + 8)  # rewards, values, two flag bytes in; advantages, returns out
print(f"gae_flags T={T} N={N}: {ts[len(ts)//2]:.1f} us  ({byt / ts[len(ts)//2] / 1e3:.0f} GB/s of {byt/1e6:.0f} MB)")
