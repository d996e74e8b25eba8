#!/usr/bin/env python3
"""tests/golden/planted_c2_subset_eva.npz: the REFERENCE's CPU fp32 scores at config C2's size -- 2297 planted gallery images (CIRR-val's
gallery size), full-depth ViT-g, every 22nd of the 4181 composed queries (191 queries x 2297 images = 438 727 scores).

TEST INFRASTRUCTURE.  tests/test_configs_gpu.py compares the fp16 engine with the fp32 engine on all 9.6 M scores of that case (the fp32
engine is 5e-6 from the reference on every golden); this fixture lets the same test state the engine's error at that size against scores
the unmodified reference produced itself.  Weights, images and queries are drawn exactly as the test draws them
(seed 5 / per-batch image seeds 1000 + s / query seed 6).  ~35 min on 8 cores.

`--seed=S` (default 5) draws another case (weights S, images S, queries S + 1; file suffix `_s<S>`): `--h16 --seed=7` is the second
fp16-valued draw VERDICT r5 item 2(b) asks for, so that "one score in 438 727" is not a one-draw statement.

`--h16` writes planted_c2_subset_eva_h16.npz: the same case on a checkpoint whose trunk weights are fp16-VALUED (what a GPU-trained
reference checkpoint holds: eva_vit.py:410-425 converts the ViT to fp16 before training; blip2.py:36-44) -- the case on which the fp16
engine is held to a flat 1e-3 at the benchmarked size.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from oracle import ref_import  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

N, NQ, STEP = 2297, 4181, 22


def main():
    h16 = "--h16" in sys.argv                            # fp16-VALUED trunk weights: what a GPU-trained reference checkpoint holds
    torch.set_num_threads(int(next((a.split("=")[1] for a in sys.argv if a.startswith("--threads=")), 8)))
    seed = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--seed=")), 5))
    noise_seed = (lambda s: 1000 + s) if seed == 5 else (lambda s: seed * 100003 + s)          # == sprc_amd/planted.py:planted_images
    cfg = get_config("pretrain")
    sd = synth.make_state_dict(cfg, seed=seed, planted=True, trunk_fp16=h16)
    model = ref_import.build_reference_model(cfg, sd)
    g = torch.Generator().manual_seed(seed)
    basis = torch.randn((8, 3, 224, 224), generator=g)
    coef = torch.randn((N, 8), generator=g)
    ids, mask, ref = synth.make_queries(NQ, N, seed=seed + 1)
    qsel = torch.arange(0, NQ, STEP)
    need = {int(r) for r in ref[qsel]}
    feats, raws = [], {}
    t0 = time.time()
    with torch.no_grad():
        for s in range(0, N, 128):                       # the test's image draw: one generator per batch of 128
            gb = torch.Generator().manual_seed(noise_seed(s))
            noise = torch.randn((min(128, N - s), 3, 224, 224), generator=gb)
            img = torch.einsum("nk,kchw->nchw", coef[s:s + 128], basis) * 0.8 + noise * 0.4
            for b in range(0, img.shape[0], 32):
                f, r = model.extract_target_features(img[b:b + 32], mode="mean")
                feats.append(f)
                for j in range(f.shape[0]):
                    if s + b + j in need:
                        raws[s + b + j] = r[j].clone()
            print(f"  images {min(s + 128, N)}/{N}  {time.time() - t0:.0f}s", flush=True)
        feats = torch.cat(feats)
        sims = []
        for s in range(0, len(qsel), 24):
            q = qsel[s:s + 24]
            model.tokenizer.set_next(ids[q], mask[q])
            rr = torch.stack([raws[int(r)] for r in ref[q]])
            sims.append(model.inference(rr, feats, ["caption"] * len(q)))
    sim = torch.cat(sims).numpy().astype(np.float32)
    out = ROOT / "tests" / "golden" / ("planted_c2_subset_eva" + ("_h16" if h16 else "") + ("" if seed == 5 else f"_s{seed}") + ".npz")
    np.savez_compressed(out, model_type="pretrain", vit_depth=cfg.vit.depth, seed=seed, n_img=N, n_q=NQ, query_step=STEP, trunk_fp16=int(h16),
                        query_index=qsel.numpy(), ref_index=ref[qsel].numpy(), sim=sim, feats_probe=feats[::256, :2].numpy())
    print(f"wrote {out}: sim {sim.shape} range [{sim.min():.3f}, {sim.max():.3f}]  ({time.time() - t0:.0f}s)")


if __name__ == "__main__":
    main()
