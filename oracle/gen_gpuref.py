#!/usr/bin/env python3
"""The reference's GPU ARITHMETIC, run on the CPU, on the planted-structure cases: tests/golden/<case>_gpuref.npz.

TEST INFRASTRUCTURE.  north_star states the score tolerance (1e-3) against the reference's CPU fp32 path; what the reference
actually runs on a GPU is an fp16-autocast ViT + ln_vision (blip2.py:36-44, align_prompt.py:366-368) on fp16 trunk weights
(eva_vit.py:410-425) with the Q-Former in fp32.  This script evaluates the UNMODIFIED reference modules in exactly that
configuration (oracle/ref_import.build_reference_model(gpu_numerics=True): the reference's own convert_weights_to_fp16 + a CPU
autocast context where the reference asks for a CUDA one) on the weights / images / queries of an existing planted golden and
stores its scores next to the CPU-fp32 ones.  tests/test_fp16_gpu.py then holds the fp16 ENGINE to: no farther from the CPU-fp32
golden than the reference's own 16-bit path is.

CPU fp16 linears are ~9x slower than fp32 ones here: planted_full_eva (96 x 48) ~ 12 min, planted_big_eva (256 x 128) ~ 30 min on 8 cores.

    python oracle/gen_gpuref.py planted_full_eva planted_big_eva
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

GOLD = ROOT / "tests" / "golden"


def gpuref(case: str) -> None:
    g = np.load(GOLD / f"{case}.npz")
    model_type, seed, n_img, n_q = str(g["model_type"]), int(g["seed"]), int(g["n_img"]), int(g["n_q"])
    depth = int(g["vit_depth"])
    cfg = get_config(model_type, vit_depth=depth)
    sd = synth.make_state_dict(cfg, seed=seed, planted=True, trunk_fp16=bool(int(g["trunk_fp16"])) if "trunk_fp16" in g.files else False)
    images = synth.make_images(n_img, seed=seed, planted=True)
    ids, mask, ref = synth.make_queries(n_q, n_img, seed=seed + 1)
    assert np.array_equal(ids.numpy(), g["input_ids"]) and np.array_equal(ref.numpy(), g["ref_index"])
    model = ref_import.build_reference_model(cfg, sd, gpu_numerics=True)
    wdt = {p.dtype for p in model.visual_encoder.parameters() if p.dim() > 1 and p.shape[0] > 1 and p.dim() != 3}
    t0 = time.time()
    feats, raw = [], []
    with torch.no_grad():
        for s in range(0, n_img, 32):                     # the batch the fp32 golden was generated with (gen_golden.planted_goldens)
            f, r = model.extract_target_features(images[s:s + 32], mode="mean")
            assert f.dtype == torch.float32 and r.dtype == torch.float32           # align_prompt.py:368 `.float()`
            feats.append(f); raw.append(r)
            print(f"  {case}: images {s + len(f)}/{n_img}  {time.time() - t0:.0f}s", flush=True)
        feats, raw = torch.cat(feats), torch.cat(raw)
        sims = []
        for s in range(0, n_q, 24):
            model.tokenizer.set_next(ids[s:s + 24], mask[s:s + 24])
            sims.append(model.inference(raw[ref[s:s + 24]], feats, ["caption"] * len(ids[s:s + 24])))
    sim = torch.cat(sims).numpy().astype(np.float32)
    d = sim - g["sim"]
    err = np.abs(d)
    q = np.quantile(err, [0.5, 0.99, 0.999, 0.9999])
    np.savez_compressed(GOLD / f"{case}_gpuref.npz", case=case, model_type=model_type, vit_depth=depth, seed=seed, n_img=n_img, n_q=n_q,
                        sim_gpuref=sim, feats_head_gpuref=feats[:4].numpy(), max_err=np.float64(err.max()),
                        rms_err=np.float64(np.sqrt((d.astype(np.float64) ** 2).mean())), quantiles=q, n_over_1e3=int((err > 1e-3).sum()),
                        trunk_weight_dtypes=np.array(sorted(str(x) for x in wdt)))
    print(f"wrote {case}_gpuref.npz: reference fp16-autocast path vs its own CPU-fp32 path: max {err.max():.3e} rms "
          f"{np.sqrt((d.astype(np.float64) ** 2).mean()):.3e} q50/99/99.9/99.99 {q}  over 1e-3: {int((err > 1e-3).sum())}/{err.size} "
          f" ({time.time() - t0:.0f}s, trunk matrix dtypes {wdt})")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for c in (sys.argv[1:] or ["planted_full_eva"]):
        gpuref(c)
