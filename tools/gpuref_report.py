#!/usr/bin/env python3
"""The fp16 engine next to the reference's OWN 16-bit arithmetic.  For every planted golden that has a `_gpuref` twin
(oracle/gen_gpuref.py: the unmodified reference under fp16 autocast with fp16 trunk weights, i.e. what it computes on a GPU), print
the score error of (a) the reference's GPU arithmetic and (b) this engine -- both against the reference's CPU fp32 scores, the
north star's yardstick -- and (c) the distance between the two.
    python tools/gpuref_report.py [case ...] [--masks IMG:FUSE ...]
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sprc_amd import engine as E, synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"
GOLD = ROOT / "tests" / "golden"


def stats(d):
    a = np.abs(d)
    return f"max {a.max():.2e} rms {np.sqrt((d.astype(np.float64) ** 2).mean()):.2e} q99.9 {np.quantile(a, 0.999):.2e} >1e-3: {int((a > 1e-3).sum())}"


def run(case, masks):
    g = np.load(GOLD / f"{case}.npz", allow_pickle=False)
    gr = np.load(GOLD / f"{case}_gpuref.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    h16 = bool(int(g["trunk_fp16"])) if "trunk_fp16" in g.files else False
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]), planted=True, trunk_fp16=h16)
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]), planted=True)
    ref = torch.from_numpy(g["ref_index"]).to(DEV)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    print(f"== {case} ({g['sim'].size} scores{', fp16-valued trunk weights' if h16 else ''})\n   reference GPU arithmetic vs reference CPU fp32: {stats(gr['sim_gpuref'] - g['sim'])}", flush=True)
    for m in masks:
        eng = E.Engine(cfg, sd, DEV, dtype="fp16", max_batch=64, qformer_x3=m)
        raw = torch.cat([eng.vit_forward(images[s:s + 64].to(DEV)) for s in range(0, images.shape[0], 64)])
        feats, _ = eng.qformer_image(raw)
        fusion, _ = eng.qformer_fuse(raw[ref], ids, mask)
        sim = E.sim_max(fusion, feats).cpu().numpy()
        print(f"   engine fp16 masks {eng.x3_image}:{eng.x3_fuse} vs reference CPU fp32: {stats(sim - g['sim'])} | vs reference GPU arithmetic: {stats(sim - gr['sim_gpuref'])}", flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    args = sys.argv[1:]
    masks = [None]
    if "--masks" in args:
        i = args.index("--masks")
        masks = [None if m == "default" else tuple(int(v) for v in m.split(":")) for m in args[i + 1:]]
        args = args[:i]
    cases = args or sorted(p.name[:-len("_gpuref.npz")] for p in GOLD.glob("*_gpuref.npz"))
    for c in cases:
        run(c, masks)
