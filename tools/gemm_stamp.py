#!/usr/bin/env python3
"""Phase timeline of the 256x256 anti-phase GEMM (SPRC_GEMM_DEBUG=64 build path): s_memtime stamps at the phase
boundaries of K-tile 8 for wave 0 (group G0) and wave 4 (G1) of workgroup 0.  Each stamp costs the wave a scalar-memory
round trip, so SPRC_GEMM_STAMP_MASK selects which of the 12 stamps are live: two live stamps measure one segment with
little perturbation.  Usage: SPRC_GEMM_DEBUG=64 SPRC_GEMM_TILE=4 [SPRC_GEMM_STAMP_MASK=0x5] gemm_stamp.py M N K
stamps: 0 loop top | 1 NC0 issued | 2 NC0 waited | 3 barrier | 4 C0 done | 5 barrier | 6 NC1 issued | 7 NC1 waited |
        8 barrier | 9 C1 done | 10 vmcnt waited | 11 barrier"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
M, N, K = (int(x) for x in sys.argv[1:4])
mask = int(os.environ.get("SPRC_GEMM_STAMP_MASK", "0xfff"), 0)
A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
R = torch.zeros((M, N), dtype=torch.float32, device="cuda")
live = [i for i in range(12) if (mask >> i) & 1]
for rep in range(3):
    E.gemm(A, W, out_dtype=L.SPRC_BF16, out=C, resid=R)
    torch.cuda.synchronize()
    ts = R.view(-1)[:64].view(torch.int64).cpu().numpy()
    for g in range(2):
        t = ts[g * 16:g * 16 + 12]
        d = "  ".join(f"{a}->{b}: {int(t[b] - t[a])}" for a, b in zip(live[:-1], live[1:]))
        print(f"mask {mask:#05x} rep{rep} G{g} :: {d}")
    pp = R.view(-1)[:128].view(torch.int64).cpu().numpy()[40:48]
    tt = R.view(-1)[:128].view(torch.int64).cpu().numpy()[32:40]
    for g in range(2):
        t = tt[g * 4:g * 4 + 4]
        print(f"   tile timeline (WG {os.environ.get('SPRC_GEMM_STAMP_WG', '552')}) G{g}: prologue {int(t[1] - t[0])}  K loop {int(t[2] - t[1])}  epilogue {int(t[3] - t[2])}  cycles  | prologue: setup {int(pp[g * 4] - t[0])}  issue {int(pp[g * 4 + 1] - pp[g * 4])}  land {int(pp[g * 4 + 2] - pp[g * 4 + 1])}  barrier {int(t[1] - pp[g * 4 + 2])}")
