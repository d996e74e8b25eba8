#!/usr/bin/env python3
"""Race screen for the staged GEMM pipelines: every bench-size shape, ITERS launches each on fresh random data, every launch
compared bit-for-bit with a second launch on the same data and (first iteration) with a dense reference; other GEMMs are
interleaved to perturb timing.  A staging race shows up as a sporadic mismatch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 25
M = 128 * 257
shapes = [("qkv", 4224, 1408, False, 0, False), ("proj", 1408, 1408, True, 0, True), ("fc1", 6144, 1408, False, 1, False),
          ("fc2", 1408, 6144, True, 0, True), ("kv_all", 9216, 1408, False, 0, False), ("qf_qkv", 2304, 768, False, 0, False)]
noise_a = torch.randn((4096, 768), device="cuda").to(torch.bfloat16)
noise_w = torch.randn((768, 768), device="cuda").to(torch.bfloat16)
bad = 0
for name, N, K, out32, act, res in shapes:
    m = 14912 if name.startswith("qf") else M
    scratch = torch.empty(8 * 128 * N, dtype=torch.float32, device="cuda")
    for it in range(ITERS):
        A = torch.randn((m, K), device="cuda").to(torch.bfloat16)
        W = (torch.randn((N, K), device="cuda") * 0.03).to(torch.bfloat16)
        r = torch.randn((m, N), device="cuda") if res else None
        kw = dict(resid=r, out_dtype=L.SPRC_F32 if out32 else L.SPRC_BF16, act=act, scratch=scratch)
        o1 = E.gemm(A, W, **kw)
        if it % 3 == 0:
            E.gemm(noise_a, noise_w)
        o2 = E.gemm(A, W, **kw)
        if not torch.equal(o1, o2):
            bad += 1
            print(f"MISMATCH {name} iter {it}: {(o1.float() - o2.float()).abs().max().item()}")
        if it == 0:
            dense = A.float() @ W.float().t()
            if act == 1:
                dense = torch.nn.functional.gelu(dense)
            if res:
                dense = dense + r
            err = (o1.float() - dense).abs().max().item()
            print(f"{name}: first launch max|diff vs dense| = {err:.3e}")
    torch.cuda.synchronize()
print("race screen:", "CLEAN" if bad == 0 else f"{bad} mismatches", f"({ITERS} x {len(shapes)} double launches)")
