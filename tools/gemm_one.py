#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 counter passes). Usage: gemm_one.py M N K [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
M, N, K = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
b = torch.randn((N,), device="cuda")
C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
for _ in range(iters):
    E.gemm(A, W, bias=b, out_dtype=L.SPRC_BF16, out=C)
torch.cuda.synchronize()
