#!/usr/bin/env python3
"""Split-precision product vs its alternatives on one shape: plain fp16 (K), fp16 + e4m3 correction segments (K, k8 = 2K: SPRC_F16X3),
and the three-fp16-segment form of ABI 3 (a plain fp16 product over 3K).  Usage: gemm_split_bench.py M,N,K ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E


def bench(f, it=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for spec in sys.argv[1:]:
    M, N, K = (int(x) for x in spec.split(","))
    x, w = torch.randn((M, K), device="cuda"), torch.randn((N, K), device="cuda") * 0.05
    A1, W1 = x.half(), w.half()
    A3, W3 = torch.cat([A1, A1, A1], 1).contiguous(), torch.cat([W1, W1, W1], 1).contiguous()
    As, Ws = E.split_rows(x), E.split_rows(w, weight=True)
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    t1 = bench(lambda: E.gemm(A1, W1, out_dtype=L.SPRC_F32, out=out))
    t3 = bench(lambda: E.gemm(A3, W3, out_dtype=L.SPRC_F32, out=out))
    ts = bench(lambda: E.gemm(As, Ws, out_dtype=L.SPRC_F32, out=out, K=K, k8=2 * K))
    fl = 2.0 * M * N * K
    print(f"{spec:22s} plain {t1:8.1f} us ({fl / t1 / 1e6:6.0f} TF)   3 x fp16 {t3:8.1f} us   fp16 + e4m3 {ts:8.1f} us   "
          f"(extra over plain: {t3 - t1:7.1f} vs {ts - t1:7.1f} us)", flush=True)
