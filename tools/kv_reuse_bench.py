#!/usr/bin/env python3
"""Query side of a CIRR-val-sized evaluation (4181 queries over 2297 gallery images, ~2000 distinct reference images) with and without
the optional reference-K|V reuse (sprc_qformer_fuse_kv): time of the fusion passes, and of the one-off projection.  Not part of bench.py."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import engine as E, synth
from sprc_amd.config import get_config

dev = torch.device("cuda", 0)
cfg = get_config("pretrain", vit_depth=1)
sd = synth.make_state_dict(cfg, seed=0, device=str(dev))
eng = E.Engine(cfg, sd, dev, dtype="fp16", max_batch=233)
N, NQ = 2297, 4181
raw = torch.randn((N, 257, 1408), device=dev)
ids, mask, ref = synth.make_queries(NQ, N, seed=1)
ids, mask, ref = ids.to(dev), mask.to(dev), ref.to(dev)
uniq, inv = torch.unique(ref, return_inverse=True)


def plain():
    return torch.cat([eng.qformer_fuse(raw[ref[s:s + 233]], ids[s:s + 233], mask[s:s + 233])[0] for s in range(0, NQ, 233)])


def reuse():
    kv = torch.cat([eng.encode_kv(raw[uniq[s:s + 128]]) for s in range(0, len(uniq), 128)])
    return torch.cat([eng.qformer_fuse_kv(kv, inv[s:s + 233], ids[s:s + 233], mask[s:s + 233])[0] for s in range(0, NQ, 233)])


for name, f in (("per-query projection", plain), ("reference K|V reuse", reuse)):
    out = f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = f(); torch.cuda.synchronize()
    print(f"{name:22s}: {1e3 * (time.perf_counter() - t0):7.1f} ms for {NQ} queries ({len(uniq)} distinct reference images)")
    res = out if name.startswith("per") else res
    if not name.startswith("per"):
        print("bit-identical fusion vectors:", torch.equal(out, res))
