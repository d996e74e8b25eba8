#!/usr/bin/env python3
"""Time the Q-Former halves of a bench step (image pass for 128 images, fusion passes for 233 queries) and print the
per-class profile of each.  Under rocprofv3 --kernel-trace it also gives the per-kernel breakdown."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L, engine as E, synth
from sprc_amd.config import get_config
cfg = get_config("pretrain", vit_depth=1)
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(cfg, seed=0, device=str(dev))
eng = E.Engine(cfg, sd, dev, dtype="bf16", max_batch=233)
lib = L.load()
raw = torch.randn((128, 257, 1408), device=dev)
ids, mask, _ = synth.make_queries(233, 2297, seed=1)
ids, mask = ids.to(dev), mask.to(dev)
ref = raw[(7919 * torch.arange(233, device=dev)) % 128].contiguous()
def prof(fn, name, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    lib.sprc_prof_enable(1); fn(); lib.sprc_prof_enable(0); torch.cuda.synchronize()
    p = (L.ProfEntry * len(L.K_CLASSES))(); L.check(lib.sprc_prof_collect(p))
    print(f"{name}: {ms:.3f} ms  " + "  ".join(f"{k}: {p[i].ms:.2f} ms / {p[i].launches} launches / {p[i].flops / max(p[i].ms, 1e-9) / 1e9:.0f} TF" for i, k in enumerate(L.K_CLASSES) if p[i].launches))
prof(lambda: eng.qformer_image(raw), "qformer_image(128)")
prof(lambda: eng.qformer_fuse(ref, ids, mask), "qformer_fuse(233)")
