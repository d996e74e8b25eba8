#!/usr/bin/env python3
"""Time arbitrary bf16 GEMM shapes.  Usage: gemm_shapes.py M,N,K[,f32|bf16[,gelu][,res]] ...   (env SPRC_GEMM_* honoured)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E

for spec in sys.argv[1:]:
    f = spec.split(",")
    M, N, K = (int(x) for x in f[:3])
    o32 = "f32" in f[3:]
    act = L.ACT_GELU if "gelu" in f[3:] else L.ACT_QUICKGELU if "qgelu" in f[3:] else L.ACT_NONE
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn((N,), device="cuda")
    C = torch.zeros((M, N), dtype=torch.float32 if o32 else torch.bfloat16, device="cuda")
    res = C if "res" in f[3:] else None
    kw = dict(bias=b, act=act, out_dtype=L.SPRC_F32 if o32 else L.SPRC_BF16, out=C, resid=res)
    for _ in range(3):
        E.gemm(A, W, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        E.gemm(A, W, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"{spec:36s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TF", flush=True)
