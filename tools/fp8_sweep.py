#!/usr/bin/env python3
"""Score error of the fp8 engine variants against the reference goldens: which 16-bit base (bf16 / fp16 + split-precision
Q-Former) and which ViT GEMMs on e4m3 operands (qkv + fc1 + fc2, or fc1 + fc2) hold which tolerance.
python tools/fp8_sweep.py            -> one line per (golden, variant): max|dsim|, rms"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sprc_amd import engine as E, synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"
GOLD = ROOT / "tests/golden"
VARIANTS = [("bf16", "all"), ("fp16", "all"), ("fp16", "mlp"), ("bf16", "mlp")]
MARGINS = [float(m) for m in sys.argv[1:]] or [1.0]


def run(eng, images, g, mb):
    raw = torch.cat([eng.vit_forward(images[s:s + mb].to(DEV)) for s in range(0, images.shape[0], mb)])
    feats, _ = eng.qformer_image(raw)
    ref = torch.from_numpy(g["ref_index"]).to(DEV)
    fusion, _ = eng.qformer_fuse(raw[ref], torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    d = E.sim_max(fusion, feats).cpu().numpy() - g["sim"]
    return float(np.abs(d).max()), float(np.sqrt((d ** 2).mean()))


for name in ("tiny_clip", "full_clip", "tiny_eva", "full_eva", "planted_eva", "planted_full_eva"):
    g = np.load(GOLD / f"{name}.npz", allow_pickle=False)
    planted = name.startswith("planted")
    depth = int(g["vit_depth"]) if "vit_depth" in g.files else None
    cfg = get_config(str(g["model_type"]) if "model_type" in g.files else "pretrain", **({"vit_depth": depth} if depth else {}))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]), planted=planted)
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]), planted=planted)
    mb = 32 if images.shape[0] > 32 else 8
    amax = {}
    for base in ("bf16", "fp16"):
        e = E.Engine(cfg, sd, DEV, dtype=base, max_batch=mb)
        amax[base] = torch.stack([e.calibrate_fp8(images[s:s + mb].to(DEV)) for s in range(0, images.shape[0], mb)]).amax(0)
        mx, rms = run(e, images, g, mb)
        print(f"{name:18s} {base:5s} (no fp8)          max|dsim| {mx:.2e}  rms {rms:.2e}", flush=True)
        del e
    for base, layers in VARIANTS:
        for margin in MARGINS:
            e = E.Engine(cfg, sd, DEV, dtype="fp8", max_batch=mb, fp8_amax=amax[base], fp8_margin=margin, fp8_base=base, fp8_layers=layers)
            mx, rms = run(e, images, g, mb)
            print(f"{name:18s} fp8/{base} {layers:3s} margin {margin:.2f}  max|dsim| {mx:.2e}  rms {rms:.2e}", flush=True)
            del e
    torch.cuda.empty_cache()
