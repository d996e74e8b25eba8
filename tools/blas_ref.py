#!/usr/bin/env python3
"""Measurement-only reference: what torch.matmul (hipBLASLt) reaches on the same shapes (not used by the product)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
def bench(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
DT = torch.float16 if "fp16" in sys.argv[1:] else torch.bfloat16
ODT = L.SPRC_F16 if DT == torch.float16 else L.SPRC_BF16
print("operands:", DT)
for M, N, K in [(32896, 4224, 1408), (32896, 6144, 1408), (32896, 1408, 6144), (32896, 1408, 1408), (4096, 4096, 4096), (8192, 8192, 8192), (14912, 768, 9216), (14912, 2304, 768), (59881, 9216, 1408)]:
    A = torch.randn((M, K), device="cuda").to(DT)
    W = (torch.randn((N, K), device="cuda") * 0.05).to(DT)
    C = torch.empty((M, N), dtype=DT, device="cuda")
    t_blas = bench(lambda: torch.matmul(A, W.t(), out=C))
    t_mine = bench(lambda: E.gemm(A, W, out_dtype=ODT, out=C))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: hipBLASLt {fl / t_blas / 1e9:7.1f} TF   sprc_gemm {fl / t_mine / 1e9:7.1f} TF")
