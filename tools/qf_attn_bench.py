#!/usr/bin/env python3
"""Q-Former attention launches of one bench step, timed alone: tools/qf_attn_bench.py  (SPRC_ATTN_SMALL_NW=1|2|4 for A/B)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import engine as E
dev = "cuda:0"
def run(B, Tq, Tk, masked, x3=False, ldkv=None):
    H, dh = 12, 64
    ldkv = ldkv or 2 * H * dh                     # (the model's cross-attention reads a layer's K|V block out of [B * 257, 9216] rows)
    q = torch.randn((B * Tq, H * dh), device=dev).half(); k = torch.randn((B * Tk, ldkv), device=dev).half()
    mask = torch.zeros((B, Tk), device=dev) if masked else None
    f = lambda: E.attention(q, k, k[:, H * dh:], B, H, Tq, Tk, dh, H * dh, ldkv, ldkv, 0.125, key_mask=mask)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    mb = B * H * dh * 2 * (2 * Tq + 2 * Tk) / 1e6
    print(f"B {B:4d} Tq {Tq:3d} Tk {Tk:3d} mask {int(masked)}: {us:7.1f} us   ({mb:.0f} MB -> {mb / us / 1e3 * 1e3:.0f} GB/s)")
print("SPRC_ATTN_SMALL_NW =", os.environ.get("SPRC_ATTN_SMALL_NW", "4 (default)"), " SPRC_ATTN_CROSS =", os.environ.get("SPRC_ATTN_CROSS", "3 (default)"))
run(233, 64, 64, True); run(233, 32, 257, False); run(128, 32, 32, False); run(128, 32, 257, False)
run(233, 32, 257, False, ldkv=9216); run(128, 32, 257, False, ldkv=9216)
