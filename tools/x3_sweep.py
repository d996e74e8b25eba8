#!/usr/bin/env python3
"""Which layer kinds of the Q-Former need split-precision (SPRC_F16X3) operands?  For each mask: max|dsim| of the fp16 engine against
the full-depth planted golden (the reference's scores) and the time of the Q-Former calls of one bench step (128 images, 233 queries).
    python tools/x3_sweep.py [mask ...]        (masks as integers, engine.X3_* bits; default: a one-out / one-in sweep)"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sprc_amd import engine as E, synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"
NAMES = ["qkv", "attn_out", "cross_q", "cross_out", "ffn_in", "ffn_out", "ckv", "heads"]
g = np.load(ROOT / "tests/golden" / os.environ.get("SPRC_GOLDEN", "planted_full_eva.npz"), allow_pickle=False)     # or planted_full_clip.npz (ViT-L)
cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
sd = synth.make_state_dict(cfg, seed=int(g["seed"]), planted=True)
images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]), planted=True)
ref = torch.from_numpy(g["ref_index"]).to(DEV)
ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
def _mask(a):
    i, _, f = a.partition(":")
    return (int(i, 0), int(f, 0) if f else int(i, 0))


# masks: "image[:query]" pairs
masks = [_mask(a) for a in sys.argv[1:]] or [(0, 0), (255, 255), (255, 0), (254, 0), (255, 142), (254, 142), (255, 174), (254, 238), (238, 238), (254, 254)]
raw16 = None
bq_ids, bq_mask, _ = synth.make_queries(233, 128, seed=3)
for m in masks:
    eng = E.Engine(cfg, sd, DEV, dtype="fp16", max_batch=233, qformer_x3=m)
    if raw16 is None:
        raw16 = torch.cat([eng.vit_forward(images[s:s + 32].to(DEV)) for s in range(0, images.shape[0], 32)])
        big = torch.randn((233, 257, cfg.vit.width), device=DEV)
    feats, _ = eng.qformer_image(raw16)
    fusion, _ = eng.qformer_fuse(raw16[ref], ids, mask)
    sim = E.sim_max(fusion, feats).cpu().numpy()
    err = np.abs(sim - g["sim"]).max()
    rms = np.sqrt(((sim - g["sim"]) ** 2).mean())
    for _ in range(2):
        eng.qformer_image(big[:128]); eng.qformer_fuse(big, bq_ids, bq_mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.qformer_image(big[:128]); eng.qformer_fuse(big, bq_ids, bq_mask)
    e1.record()
    torch.cuda.synchronize()
    on = " | ".join((",".join(n for i, n in enumerate(NAMES) if mm >> i & 1) or "-") for mm in m)
    print(f"mask {m[0]:3d}:{m[1]:3d} [{on}]: max|dsim| {err:.2e} rms {rms:.2e}   Q-Former per step {e0.elapsed_time(e1) / 20:.2f} ms", flush=True)
    del eng
    torch.cuda.empty_cache()
