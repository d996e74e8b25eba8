#!/usr/bin/env python3
"""Time fp8-operand GEMM shapes (random e4m3 operands).  Usage: gemm_fp8_bench.py M,N,K[,f32|fp8] ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
for spec in sys.argv[1:]:
    f = spec.split(",")
    M, N, K = (int(x) for x in f[:3])
    A = torch.randn((M, K), device="cuda").clamp(-3, 3).to(torch.float8_e4m3fn)
    W = (torch.randn((N, K), device="cuda")).clamp(-3, 3).to(torch.float8_e4m3fn)
    ws = torch.ones(N, device="cuda")
    odt = L.SPRC_F32 if "f32" in f[3:] else L.SPRC_FP8 if "fp8" in f[3:] else L.SPRC_BF16
    C = torch.empty((M, N), dtype={L.SPRC_F32: torch.float32, L.SPRC_BF16: torch.bfloat16, L.SPRC_FP8: torch.float8_e4m3fn}[odt], device="cuda")
    kw = dict(out_dtype=odt, out=C, w_scale=ws, a_scale=1.0, out_scale=0.01)
    for _ in range(3): E.gemm(A, W, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): E.gemm(A, W, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"fp8 {spec:30s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TF", flush=True)
