#!/usr/bin/env python3
"""Time the Q-Former's GEMM launches as the model makes them (shape, split-precision or not, epilogue), under whatever SPRC_GEMM_*
environment is set (the switches are read once per process: one process per configuration -- tools/qf_tiles.sh drives the sweep).
Usage: qf_shapes.py [case-name-substring ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E

B, S, Lq = 233, 64, 32
R, Rq, Ri = B * S, B * Lq, 128 * Lq
# name, M, N, K, mix, out, act, resid, pair
CASES = [
    ("q.qkv          ", R, 2304, 768, 0, "f16", 0, 0, 0),
    ("q.attn_out     ", R, 768, 768, 1, "f32", 0, 1, 0),
    ("q.cross_q      ", Rq, 768, 768, 1, "f16", 0, 0, 0),
    ("q.cross_out    ", Rq, 768, 768, 1, "f32", 0, 1, 0),
    ("q.ffn_in  pair ", Rq, 3072, 768, 1, "x3", 1, 0, 1),
    ("q.ffn_out pair ", Rq, 768, 3072, 1, "f32", 0, 1, 1),
    ("q.ffn_in  one  ", R, 3072, 768, 1, "x3", 1, 0, 0),
    ("q.ffn_out one  ", R, 768, 3072, 1, "f32", 0, 1, 0),
    ("i.qkv          ", Ri, 2304, 768, 0, "f16", 0, 0, 0),
    ("i.attn_out     ", Ri, 768, 768, 1, "f32", 0, 1, 0),
    ("i.cross_q      ", Ri, 768, 768, 1, "f16", 0, 0, 0),
    ("i.ffn_in       ", Ri, 3072, 768, 1, "x3", 1, 0, 0),
    ("i.ffn_out      ", Ri, 768, 3072, 1, "f32", 0, 1, 0),
]


def bench(f, it=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


sel = sys.argv[1:]
tot = 0.0
for name, M, N, K, mix, out, act, res, pair in CASES:
    if sel and not any(s in name for s in sel):
        continue
    rows = M * 2 if pair else M
    x, w = torch.randn((rows, K), device="cuda"), torch.randn((N, K), device="cuda") * 0.05
    A = E.split_rows(x) if mix else x.half()
    W = E.split_rows(w, weight=True) if mix else w.half()
    W2 = W.clone()
    bias = torch.randn((N,), device="cuda")
    odt = {"f16": L.SPRC_F16, "f32": L.SPRC_F32, "x3": L.SPRC_F16X3}[out]
    C = torch.zeros((rows, 2 * N if out == "x3" else N), dtype=torch.float32 if out == "f32" else torch.float16, device="cuda")
    r = C if res else None
    kw = dict(out_dtype=odt, act=L.ACT_GELU if act else L.ACT_NONE, resid=r, out=C, K=K, k8=2 * K if mix else 0)
    if pair:
        qm, tm = E.rowmap(Lq, S, 0), E.rowmap(S - Lq, S, Lq)
        f = lambda: E.gemm_pair(A, W, W2, bias, bias, qm, tm, qm, tm, M, **kw)
    else:
        f = lambda: E.gemm(A, W, bias=bias, M=M, **kw)
    us = bench(f)
    tot += us
    fl = 2.0 * rows * N * K
    print(f"{name} {rows:6d} x {N:4d} x {K:4d} {'mix' if mix else '   '} {out:3s}  {us:8.1f} us  {fl / us / 1e6:7.0f} TF(alg)", flush=True)
print(f"sum {tot:8.1f} us")
