#!/usr/bin/env python3
"""Duo GEMM kernel (gemm_duo.hpp) on the GPU: results against the fp32 product of the SAME rounded operands (torch, measurement only) and
timing on the ViT-g shapes.  SPRC_GEMM_DUO / SPRC_DUO_SLEEP / SPRC_DUO_ORDER are read once per process: tools/duo_ab.sh loops over them.
Usage: duo_check.py [check] [time]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E

DT = torch.float16
tag = f"DUO={os.environ.get('SPRC_GEMM_DUO', '0')} SLEEP={os.environ.get('SPRC_DUO_SLEEP', 'auto')} ORDER={os.environ.get('SPRC_DUO_ORDER', '-')}"

def ref(A, W, b, act, res):
    y = A.float() @ W.float().t() + b
    if act == L.ACT_GELU:
        y = torch.nn.functional.gelu(y)
    if res is not None:
        y = y + res
    return y

if "check" in sys.argv:
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, K, o32, act, res) in [(256, 128, 64, False, L.ACT_NONE, False), (256, 128, 128, True, L.ACT_NONE, False), (300, 200, 192, True, L.ACT_NONE, True),
                                     (1000, 1408, 1408, False, L.ACT_NONE, False), (32896, 1408, 1408, True, L.ACT_NONE, True),
                                     (4099, 6144, 1408, False, L.ACT_GELU, False), (2056, 1408, 6144, True, L.ACT_NONE, True),
                                     (70000, 384, 192, False, L.ACT_NONE, False), (513, 132, 2048, True, L.ACT_NONE, False)]:
        A = torch.randn((M, K), device="cuda").to(DT)
        W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(DT)
        b = torch.randn((N,), device="cuda")
        r = torch.randn((M, N), device="cuda") if res else None
        C = E.gemm(A, W, bias=b, act=act, out_dtype=L.SPRC_F32 if o32 else L.SPRC_F16, resid=r)
        torch.cuda.synchronize()
        y = ref(A, W, b, act, r)
        err = (C.float() - y).abs().max().item()
        tol = 2e-4 if o32 else 6e-3
        worst = max(worst, err / tol)
        print(f"{tag} check M={M} N={N} K={K} f32={o32} act={act} res={res}: max err {err:.3e} {'OK' if err < tol else 'FAIL'}", flush=True)
    # run-to-run bit stability on a shape with many tiles per workgroup
    A = torch.randn((32896, 1408), device="cuda").to(DT); W = (torch.randn((4224, 1408), device="cuda") * 0.03).to(DT)
    c0 = E.gemm(A, W, out_dtype=L.SPRC_F16).clone()
    same = all(torch.equal(c0, E.gemm(A, W, out_dtype=L.SPRC_F16)) for _ in range(5))
    print(f"{tag} repeat-bit-identical: {same}   worst err/tol {worst:.2f}", flush=True)

if "time" in sys.argv:
    SH = [("qkv", 32896, 4224, 1408, False, L.ACT_NONE, False), ("fc1+gelu", 32896, 6144, 1408, False, L.ACT_GELU, False),
          ("proj+res", 32896, 1408, 1408, True, L.ACT_NONE, True), ("fc2+res", 32896, 1408, 6144, True, L.ACT_NONE, True),
          ("qf 14912x2304x768", 14912, 2304, 768, False, L.ACT_NONE, False), ("kv 59881x9216x1408", 59881, 9216, 1408, False, L.ACT_NONE, False),
          ("8192^3", 8192, 8192, 8192, False, L.ACT_NONE, False)]
    tot = 0.0
    for name, M, N, K, o32, act, res in SH:
        A = torch.randn((M, K), device="cuda").to(DT)
        W = (torch.randn((N, K), device="cuda") * 0.03).to(DT)
        b = torch.randn((N,), device="cuda")
        C = torch.zeros((M, N), dtype=torch.float32 if o32 else DT, device="cuda")
        scratch = torch.empty((8 * 128 * N,), dtype=torch.float32, device="cuda")
        kw = dict(bias=b, act=act, out_dtype=L.SPRC_F32 if o32 else L.SPRC_F16, out=C, resid=C if res else None, scratch=scratch)
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
        if name in ("qkv", "fc1+gelu", "proj+res", "fc2+res"):
            tot += us
        print(f"{tag} {name:20s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TF", flush=True)
    print(f"{tag} ViT layer (4 products): {tot:.1f} us", flush=True)
