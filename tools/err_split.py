#!/usr/bin/env python3
"""Where does a 16-bit engine's score error come from?  Full-depth planted case (the sizes of tests/golden/planted_full_eva.npz),
fp32 engine as the truth (it matches the reference to 3.5e-6): the ViT and the Q-Former of the 16-bit engine are swapped in one at a
time.  python tools/err_split.py [fp16|bf16]"""
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
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
g = np.load(ROOT / "tests/golden" / os.environ.get("SPRC_GOLDEN", "planted_full_eva.npz"), allow_pickle=False)     # or planted_full_clip.npz (ViT-L)
cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
sd = synth.make_state_dict(cfg, seed=int(g["seed"]), planted=True)
images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]), planted=True)
ref = torch.from_numpy(g["ref_index"]).to(DEV)
ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
e32 = E.Engine(cfg, sd, DEV, dtype="fp32", max_batch=32)
x3 = None if len(sys.argv) < 3 else int(sys.argv[2], 0)             # fp16: split-precision mask of the Q-Former (engine.X3_*; default: the engine's)
e16 = E.Engine(cfg, sd, DEV, dtype=dtype, max_batch=32, qformer_x3=x3)
print(f"16-bit engine: {dtype}, split-precision Q-Former: {e16.x3}")


def vit(eng):
    return torch.cat([eng.vit_forward(images[s:s + 32].to(DEV)) for s in range(0, images.shape[0], 32)])


def rel(a, b):
    return float((a - b).norm() / b.norm())


raw32, raw16 = vit(e32), vit(e16)
print(f"[{dtype}] raw: rel-rms err {rel(raw16, raw32):.2e}  max abs {float((raw16 - raw32).abs().max()):.2e}")
out = {}
for vname, raw in (("vit32", raw32), ("vit16", raw16)):
    for qname, eng in (("qf32", e32), ("qf16", e16)):
        feats, _ = eng.qformer_image(raw)
        fusion, _ = eng.qformer_fuse(raw[ref], ids, mask)
        out[(vname, qname)] = (feats, fusion, E.sim_max(fusion, feats))
f0, u0, s0 = out[("vit32", "qf32")]
print(f"fp32 engine vs the reference golden: max|dsim| = {np.abs(s0.cpu().numpy() - g['sim']).max():.2e}")
for k, (f, u, s) in out.items():
    # which side of the score carries the error: gallery features or the fused query
    s_f = E.sim_max(u0, f)
    s_u = E.sim_max(u, f0)
    print(f"{k}: max|dsim| {float((s - s0).abs().max()):.2e}  rms {float((s - s0).pow(2).mean().sqrt()):.2e} | feats rel {rel(f, f0):.2e} max {float((f - f0).abs().max()):.2e} "
          f"(dsim from feats only {float((s_f - s0).abs().max()):.2e}) | fusion rel {rel(u, u0):.2e} (dsim from fusion only {float((s_u - s0).abs().max()):.2e})")
