#!/usr/bin/env python3
"""Micro-benchmark of sprc_gemm on the ViT-g / Q-Former shapes of the bench workload (MI355X only).
Usage: SPRC_GEMM_IMPL=<n> python tools/gemm_bench.py [--dtype bf16|fp32] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L  # noqa: E402
from sprc_amd import engine as E  # noqa: E402

SHAPES = [("vit qkv", 32896, 4224, 1408, L.ACT_NONE, "bf16"), ("vit proj+res", 32896, 1408, 1408, L.ACT_NONE, "f32r"),
          ("vit fc1+gelu", 32896, 6144, 1408, L.ACT_GELU, "bf16"), ("vit fc2+res", 32896, 1408, 6144, L.ACT_NONE, "f32r"),
          ("qf kv_all", 32896, 9216, 1408, L.ACT_NONE, "bf16"), ("qf qkv", 8192, 2304, 768, L.ACT_NONE, "bf16"),
          ("qf ffn in", 4096, 3072, 768, L.ACT_GELU, "bf16"), ("qf ffn out", 4096, 768, 3072, L.ACT_NONE, "f32r")]

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
dev = "cuda:0"
tot_f = tot_t = 0.0
for name, M, N, K, act, out in SHAPES:
    A = torch.randn((M, K), device=dev).to(tdt)
    W = (torch.randn((N, K), device=dev) * 0.05).to(tdt)
    b = torch.randn((N,), device=dev)
    resid = torch.randn((M, N), device=dev) if out == "f32r" else None
    odt = L.SPRC_F32 if out == "f32r" else (L.SPRC_BF16 if a.dtype == "bf16" else L.SPRC_F32)
    C_ = torch.empty((M, N), dtype=torch.float32 if odt == L.SPRC_F32 else torch.bfloat16, device=dev)
    for _ in range(3):
        E.gemm(A, W, bias=b, resid=resid, out_dtype=odt, act=act, out=C_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        E.gemm(A, W, bias=b, resid=resid, out_dtype=odt, act=act, out=C_)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = 2.0 * M * N * K
    tot_f += fl
    tot_t += ms
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:5d}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s")
print(f"impl={os.environ.get('SPRC_GEMM_IMPL', 'default')} dtype={a.dtype}: weighted {tot_f / tot_t / 1e9:.1f} TFLOP/s")
