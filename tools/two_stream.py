#!/usr/bin/env python3
"""Does running the two halves of a batch on two streams (offset by one GEMM) beat one stream over the whole batch?
One ViT-g layer's four GEMMs + LayerNorms as the unit of work."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E

D, F = 1408, 6144
def make(M):
    bf = torch.bfloat16
    t = dict(h=torch.randn(M, D, device="cuda").to(bf), qkv=torch.empty(M, 3 * D, device="cuda", dtype=bf),
             ctx=torch.randn(M, D, device="cuda").to(bf), x=torch.randn(M, D, device="cuda"),
             mlp=torch.empty(M, F, device="cuda", dtype=bf))
    return t
W = dict(qkv=(torch.randn(3 * D, D, device="cuda") * 0.03).to(torch.bfloat16), proj=(torch.randn(D, D, device="cuda") * 0.03).to(torch.bfloat16),
         fc1=(torch.randn(F, D, device="cuda") * 0.03).to(torch.bfloat16), fc2=(torch.randn(D, F, device="cuda") * 0.02).to(torch.bfloat16))
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")

def layer(t):
    E.layernorm(t["x"], g, b, 1e-6, L.SPRC_BF16, want32=False, y16=t["h"])
    E.gemm(t["h"], W["qkv"], out_dtype=L.SPRC_BF16, out=t["qkv"])
    M = t["x"].shape[0]
    q = t["qkv"]
    E.attention(q, q[:, D:], q[:, 2 * D:], M // 257, 16, 257, 257, 88, 3 * D, 3 * D, 3 * D, 88 ** -0.5, out=t["ctx"])
    E.gemm(t["ctx"], W["proj"], resid=t["x"], out_dtype=L.SPRC_F32, out=t["x"])
    E.layernorm(t["x"], g, b, 1e-6, L.SPRC_BF16, want32=False, y16=t["h"])
    E.gemm(t["h"], W["fc1"], out_dtype=L.SPRC_BF16, act=L.ACT_GELU, out=t["mlp"])
    E.gemm(t["mlp"], W["fc2"], resid=t["x"], out_dtype=L.SPRC_F32, out=t["x"])

full, ha, hb = make(128 * 257), make(64 * 257), make(64 * 257)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NL = 20
def run_one():
    for _ in range(NL):
        layer(full)
def run_two():
    with torch.cuda.stream(s1):
        for _ in range(NL):
            layer(ha)
    with torch.cuda.stream(s2):
        E.layernorm(hb["x"], g, b, 1e-6, L.SPRC_BF16, want32=False, y16=hb["h"])      # small offset
        for _ in range(NL):
            layer(hb)
def run_two_half():
    with torch.cuda.stream(s1):
        for _ in range(NL):
            layer(ha)
    with torch.cuda.stream(s2):
        E.gemm(hb["h"], W["fc1"], out_dtype=L.SPRC_BF16, act=L.ACT_GELU, out=hb["mlp"])      # half a layer of offset
        E.gemm(hb["mlp"], W["fc2"], resid=hb["x"], out_dtype=L.SPRC_F32, out=hb["x"])
        for _ in range(NL):
            layer(hb)
q1, q2, q3, q4 = (make(32 * 257) for _ in range(4))
s3, s4 = torch.cuda.Stream(), torch.cuda.Stream()
def run_four():
    for st, t in ((s1, q1), (s2, q2), (s3, q3), (s4, q4)):
        with torch.cuda.stream(st):
            for _ in range(NL):
                layer(t)
for name, fn in (("one stream, 128 images", run_one), ("two streams, 64 + 64", run_two), ("two streams, half-layer offset", run_two_half),
                 ("four streams, 4 x 32", run_four), ("one stream, 128 images", run_one), ("two streams, 64 + 64", run_two)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:28s} {dt / NL * 1e3:8.3f} ms per layer", flush=True)
