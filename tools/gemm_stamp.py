#!/usr/bin/env python3
"""Phase timeline of the 256x256 anti-phase GEMM (SPRC_GEMM_DEBUG=64 build path): s_memtime deltas of K-tile 8 for
wave 0 (group G0) and wave 4 (G1) of workgroup 0.  Usage: SPRC_GEMM_DEBUG=64 SPRC_GEMM_TILE=4 gemm_stamp.py M N K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
R = torch.zeros((M, N), dtype=torch.float32, device="cuda")
names = ["NC0 issue", "NC0 lgkm wait", "barrier", "C0 cluster", "barrier", "NC1 issue", "NC1 waits", "barrier", "C1 cluster",
         "vmcnt wait", "barrier"]
for rep in range(3):
    E.gemm(A, W, out_dtype=L.SPRC_BF16, out=C, resid=R)
    torch.cuda.synchronize()
    ts = R.view(-1)[:64].view(torch.int64).cpu().numpy()
    for g in range(2):
        t = ts[g * 16:g * 16 + 12]
        d = [int(t[i + 1] - t[i]) for i in range(11)]
        print(f"rep{rep} G{g} start+{int(t[0] - ts[0]):6d}  total {int(t[11] - t[0]):5d} :: " + "  ".join(f"{n}={v}" for n, v in zip(names, d)))
