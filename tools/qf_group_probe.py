#!/usr/bin/env python3
"""What larger Q-Former batches buy: image pass at 128 / 256 / 512 images and fusion passes at 233 / 466 / 699 / 932 queries, per unit.
(fp16 engine with the default split-precision masks, full-depth Q-Former, depth-1 ViT: the ViT is not run.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import engine as E, synth
from sprc_amd.config import get_config
cfg = get_config("pretrain", vit_depth=1)
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(cfg, seed=0, device=str(dev))
eng = E.Engine(cfg, sd, dev, dtype="fp16", max_batch=932)
raw = torch.randn((512, 257, 1408), device=dev)

def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for b in (128, 256, 384, 512):
    ms = t(lambda: eng.qformer_image(raw[:b]))
    print(f"qformer_image({b:4d}): {ms:8.3f} ms   {ms / b * 128:7.3f} ms per 128 images", flush=True)
for q in (233, 466, 699, 932):
    ids, mask, _ = synth.make_queries(q, 2297, seed=1)
    ids, mask = ids.to(dev), mask.to(dev)
    ref = raw[(7919 * torch.arange(q, device=dev)) % 512].contiguous()
    ms = t(lambda: eng.qformer_fuse(ref, ids, mask))
    print(f"qformer_fuse({q:4d}): {ms:8.3f} ms   {ms / q * 233:7.3f} ms per 233 queries", flush=True)
