#!/usr/bin/env python3
"""Run the ViT attention shape a few times (timing + rocprofv3 counter passes). Usage: attn_one.py [B H T dh iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import engine as E
a = [int(x) for x in sys.argv[1:]]
B, H, T, dh = (a + [128, 16, 257, 88])[:4] if len(a) >= 4 else (128, 16, 257, 88)
iters = a[4] if len(a) > 4 else 10
D = H * dh
qkv = torch.randn((B * T, 3 * D), device="cuda").to(torch.bfloat16)
out = torch.empty((B * T, D), dtype=torch.bfloat16, device="cuda")
f = lambda: E.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, T, T, dh, 3 * D, 3 * D, 3 * D, dh ** -0.5, out=out)
for _ in range(2): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attention B={B} H={H} T={T} dh={dh}: {ms*1e3:.1f} us  {4.0*B*H*T*T*dh/ms/1e9:.1f} TFLOP/s")
