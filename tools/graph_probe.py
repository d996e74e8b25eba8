#!/usr/bin/env python3
"""Is a hipGraph replay of sprc_vit_forward (273 launches) faster than launching it eagerly?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import engine as E, synth
from sprc_amd.config import get_config
cfg = get_config("pretrain")
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(cfg, seed=0, device=str(dev))
eng = E.Engine(cfg, sd, dev, dtype="bf16", max_batch=128)
del sd
images = torch.randn((128, 3, 224, 224), device=dev)
raw = torch.empty((128, cfg.vit.tokens, cfg.vit.width), dtype=torch.float32, device=dev)
for _ in range(2):
    eng.vit_forward(images, out=raw)
torch.cuda.synchronize()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("eager  %.3f ms" % timeit(lambda: eng.vit_forward(images, out=raw)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.vit_forward(images, out=raw)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    eng.vit_forward(images, out=raw)
print("graph  %.3f ms" % timeit(g.replay))
print("eager  %.3f ms" % timeit(lambda: eng.vit_forward(images, out=raw)))
print("graph  %.3f ms" % timeit(g.replay))
