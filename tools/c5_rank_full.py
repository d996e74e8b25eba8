#!/usr/bin/env python3
"""BASELINE config C5's ranking at its stated size on ONE GPU: 1 000 000 x 32 x 256 bf16 features in 8 logical shards x 10 000 queries
(dist.rank_logical_shards), timed (best of 3) -> gpurun_out/c5_rank_full.json (copied to profiles/r05_c5_rank_full.json).
The bit-identity checks live in tests/test_fullsize_gpu.py::test_c5_full_size_ranking_one_gpu_eight_logical_shards."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd.dist import ShardedRanker, rank_logical_shards

DEV = torch.device("cuda:0")
N, NQ, K = 1_000_000, 10_000, 51
g = torch.Generator(device=DEV).manual_seed(7)
feats = torch.empty((N, 32, 256), dtype=torch.bfloat16, device=DEV)
for s in range(0, N, 25_000):
    feats[s:s + 25_000] = torch.nn.functional.normalize(torch.randn((25_000, 32, 256), generator=g, device=DEV), dim=-1).to(torch.bfloat16)
fusion = torch.nn.functional.normalize(torch.randn((NQ, 256), generator=g, device=DEV), dim=-1).to(torch.bfloat16)

def best(fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), out
t_sh, (mv, mi) = best(lambda: rank_logical_shards(feats, fusion, K, 8))
t_gl, (gv, gi) = best(lambda: ShardedRanker(feats, 0, always_exchange=False).rank(fusion, K))
flop = 2.0 * 32 * 256 * N * NQ
out = {"workload": f"C5 ranking at full size on one GPU: {N} x 32 x 256 bf16 gallery features (16.4 GB) x {NQ} queries, top-{K}",
       "eight_logical_shards_ms": round(t_sh * 1e3, 1), "global_pass_ms": round(t_gl * 1e3, 1), "tflop": round(flop / 1e12, 2),
       "eight_logical_shards_tflops": round(flop / t_sh / 1e12, 1), "identical_bits": bool(torch.equal(mi, gi) and torch.equal(mv, gv)),
       "feature_bytes_read_GBs_eight_shards": round(feats.numel() * 2 * (NQ / (2 * 2**30 // (4 * 125_000))) / t_sh / 1e9, 1),
       "note": "per shard: blocks of 4294 query rows (2-GB score budget) -> max-over-32 GEMM + wavefront top-51; merge of 8 x 51 candidates per query"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/c5_rank_full.json", "w"), indent=1)
print(json.dumps(out))
