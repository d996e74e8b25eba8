#!/usr/bin/env python3
"""Time the ViT LayerNorm shape (fp32 [M,1408] -> bf16). Usage: ln_one.py [M D iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E
a = [int(x) for x in sys.argv[1:]]
M, D, iters = (a + [32896, 1408, 50])[:3] if len(a) >= 2 else (32896, 1408, 50)
x = torch.randn((M, D), device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
y = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")          # flush the Infinity Cache between launches
def run(flush):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(iters):
        if flush: big.zero_()
        e0.record(); E.layernorm(x, g, b, 1e-6, L.SPRC_BF16, want32=False, y16=y); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3
run(False)
print(f"layernorm M={M} D={D}: warm {run(False):.1f} us, cache-flushed {run(True):.1f} us  ({M*D*6/1e6:.0f} MB moved)")
