#!/usr/bin/env python3
"""Fixed-overhead vs per-K-tile cost of sprc_gemm: time(K) for fixed M, N (MI355X only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32896
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
act = int(sys.argv[3]) if len(sys.argv) > 3 else 0
res = []
for K in (128, 704, 1408, 2816, 5632):
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn((N,), device="cuda")
    C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    for _ in range(3): E.gemm(A, W, bias=b, out_dtype=L.SPRC_BF16, act=act, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): E.gemm(A, W, bias=b, out_dtype=L.SPRC_BF16, act=act, out=C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res.append((K, ms))
    print(f"M={M} N={N} K={K:5d} act={act}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF")
(k0, t0), (k1, t1) = res[-2], res[-1]
slope = (t1 - t0) / (k1 - k0)
print(f"slope {slope*64*1e3:.3f} us per 64-wide K-tile -> asymptotic {2.0*M*N/slope/1e9:.0f} TF; intercept {(t0 - slope*k0)*1e3:.1f} us")
