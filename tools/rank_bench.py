#!/usr/bin/env python3
"""Ranking stage at BASELINE.json config C5's per-GPU size: 1 M-image gallery / 8 GPUs = 125 000 images per shard (bf16 features,
2 GB) against 10 000 composed queries: max-over-32 cosine GEMM (sprc_sim_max) + top-51 (sprc_topk).
Usage: rank_bench.py [N_shard nq dtype]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import engine as E
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dt = torch.bfloat16 if (len(sys.argv) <= 3 or sys.argv[3] == "bf16") else torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
feats = torch.nn.functional.normalize(torch.randn((N, 32, 256), generator=g, device="cuda"), dim=-1).to(dt)
fusion = torch.nn.functional.normalize(torch.randn((nq, 256), generator=g, device="cuda"), dim=-1).to(dt)
QB = 2048                                      # queries per pass: sim tile [QB, N] fp32 = 1 GB
sim = torch.empty((QB, N), dtype=torch.float32, device="cuda")
def run():
    out = []
    for s in range(0, nq, QB):
        n = min(QB, nq - s)
        E.sim_max(fusion[s:s + n], feats, out=sim[:n])
        out.append(E.topk(sim[:n], 51))
    return out
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); it = 3
for _ in range(it): run()
torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / it
fl = 2.0 * nq * N * 32 * 256
print(f"rank {nq} queries x {N} images ({feats.dtype}): {dt_s*1e3:.1f} ms  {fl/dt_s/1e12:.1f} TFLOP/s  "
      f"{nq*N/dt_s/1e9:.2f} G pairs/s; gallery pass {N*32*256*feats.element_size()/1e9:.2f} GB x {(nq+QB-1)//QB}")
