#!/usr/bin/env python3
"""CU-partition probe (round 6): does running two half-chip ViT forwards SIDE BY SIDE on disjoint CUs (sprc_stream_create_partition) beat
one full-chip forward?  Every kernel of the ViT is either matrix-bound (K loops) or memory-bound (epilogues, LayerNorm, attention), and
a kernel that owns all 256 CUs keeps them in lockstep; two partitions drift apart and one's memory bursts fall under the other's K loops.

    python tools/cumask_probe.py [--depth 39] [--iters 10]

Prints ms per 128 images for: (a) one stream, full chip, batch 128; (b) two partition streams, batch 64 each, concurrently;
(c) two partition streams, batch 128 each (ms per 128 images = half the pair's time); (d) two PLAIN streams, batch 64 each (no masks:
round 5's negative result, for reference); and checks that (b)'s outputs are bit-identical to (a)'s rows."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L
from sprc_amd import engine as E, synth
from sprc_amd.config import get_config

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=39)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--nparts", type=int, default=2)
a = ap.parse_args()

dev = torch.device("cuda", 0)
lib = L.load()
cfg = get_config("pretrain", vit_depth=a.depth)
sd = synth.make_state_dict(cfg, seed=0, device=str(dev))
eng = E.Engine(cfg, sd, dev, dtype=a.dtype, max_batch=128)
del sd
g = torch.Generator(device=dev).manual_seed(1)
images = torch.randn((128, 3, 224, 224), generator=g, device=dev)
NP = a.nparts
parts = []
for i in range(NP):
    h = C.c_void_p()
    L.check(lib.sprc_stream_create_partition(i, NP, C.byref(h)), "sprc_stream_create_partition")
    parts.append(torch.cuda.ExternalStream(h.value, device=dev))
    print(f"partition {i}/{NP}: stream {h.value:#x}, {lib.sprc_stream_cus(h)} CUs")
plain = [torch.cuda.Stream(device=dev) for _ in range(NP)]
# NOT the null stream: hipExtStreamCreateWithCUMask makes BLOCKING streams (no flags argument), and every operation on the legacy null
# stream -- an event record for wait_stream included -- orders itself after ALL earlier work of all blocking streams: with the null
# stream as the fork/join point the two partitions ran strictly one after the other (first version of this probe: 60.9 ms + 54.2 ms)
main = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
out_full = torch.empty((128, cfg.vit.tokens, cfg.vit.width), device=dev)
out_part = torch.empty((NP, 128, cfg.vit.tokens, cfg.vit.width), device=dev)


def full():
    eng.vit_forward(images, out=out_full)


def split(streams, per):
    for i, st in enumerate(streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            if per == 128:
                eng.vit_forward(images, out=out_part[i], slot=i + 1)
            else:
                eng.vit_forward(images[i * per:(i + 1) * per], out=out_part[0, i * per:(i + 1) * per], slot=i + 1)
    for st in streams:
        main.wait_stream(st)


def timeit(f, imgs_per_call):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters * 1e3 * 128 / imgs_per_call


per = 128 // NP
res = {}


def one_part(i, n):
    with torch.cuda.stream(parts[i]):
        eng.vit_forward(images[:n], out=out_part[0, :n], slot=i + 1)


# effective CU count of a partition: one large product, full chip vs partition 0 alone
A = torch.randn((32768, 1408), device=dev).to(torch.float16)
W = torch.randn((4096, 1408), device=dev).to(torch.float16)
Cc = torch.empty((32768, 4096), device=dev, dtype=torch.float16)


def gemm_on(st):
    with torch.cuda.stream(st):
        E.gemm(A, W, out=Cc) if hasattr(E, "gemm") else None


if hasattr(E, "gemm"):
    tf = timeit(lambda: gemm_on(main), 128)
    tp = timeit(lambda: gemm_on(parts[0]), 128)
    print(f"32768 x 4096 x 1408 product: full chip {tf:.3f} ms, partition 0 alone {tp:.3f} ms (ratio {tp / tf:.2f}; {NP}.00 = the partition has 1/{NP} of the CUs)")
    def gemm_both():
        for st in parts:
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for _ in range(8):
                    E.gemm(A, W, out=Cc if st is parts[0] else Cd)
        for st in parts:
            main.wait_stream(st)

    def gemm_one():
        parts[0].wait_stream(main)
        with torch.cuda.stream(parts[0]):
            for _ in range(8):
                E.gemm(A, W, out=Cc)
        main.wait_stream(parts[0])
    Cd = torch.empty_like(Cc)
    t1 = timeit(gemm_one, 128)
    t2 = timeit(gemm_both, 128)
    print(f"8 products on partition 0 alone {t1:.3f} ms; 8 products on EACH of the {NP} partitions at once {t2:.3f} ms "
          f"(= {t1:.2f}: side by side at the alone clock; {NP} x: serialised)")
    # do the two partitions' kernels overlap in time?  events on each stream around its share of (b)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in parts]
    base = torch.cuda.Event(enable_timing=True)
    base.record(main)
    for i, st in enumerate(parts):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            ev[i][0].record(st)
            eng.vit_forward(images[i * (128 // NP):(i + 1) * (128 // NP)], out=out_part[0, i * (128 // NP):(i + 1) * (128 // NP)], slot=i + 1)
            ev[i][1].record(st)
    torch.cuda.synchronize()
    for i in range(NP):
        print(f"partition {i}: its ViT forward ran from {base.elapsed_time(ev[i][0]):.2f} ms to {base.elapsed_time(ev[i][1]):.2f} ms after the common start")
res[f"(s) partition 0 ALONE, batch {per}"] = timeit(lambda: one_part(0, per), per)
res["(s') partition 1 ALONE, batch 128"] = timeit(lambda: one_part(1, 128), 128)
res["(a) one stream, full chip, batch 128"] = timeit(full, 128)
res[f"(b) {NP} CU-partition streams, batch {per} each"] = timeit(lambda: split(parts, per), 128)
res[f"(c) {NP} CU-partition streams, batch 128 each"] = timeit(lambda: split(parts, 128), 128 * NP)
res[f"(d) {NP} plain streams, batch {per} each"] = timeit(lambda: split(plain, per), 128)
res["(a') one stream, full chip, batch 128 (again)"] = timeit(full, 128)
for k, v in res.items():
    print(f"{k:55s} {v:8.2f} ms per 128 images")
full()
split(parts, per)
torch.cuda.synchronize()
same = torch.equal(out_full, out_part[0])
print("partitioned outputs bit-identical to the full-chip batch:", same, "" if same else f"(max |d| = {float((out_full - out_part[0]).abs().max()):.3e})")
