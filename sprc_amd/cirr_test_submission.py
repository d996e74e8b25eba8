#!/usr/bin/env python3
"""`python -m sprc_amd.cirr_test_submission --blip-model-name blip2_cir_align_prompt --model-path X`

CIRR test1 submission files with the reference's flags and JSON layout (src/cirr_test_submission.py:16-58,
203-222).  --rerank (the reference's stage-2 branch, :88-112) re-scores every query's top-50 with `inference_rerank`
(blip2_qformer_cir_rerank.py:399-445: sprc_qformer_encode_kv + sprc_qformer_itm); single-process only.
"""
from __future__ import annotations

import json
from argparse import ArgumentParser

from .blip_validate import _load
from .harness import extract_index_blip_features, generate_cirr_test_dicts


def generate_cirr_test_submissions(file_name: str, blip_model, preprocess, txt_processors, rerank=False, num_workers: int = 2):
    from .data_utils import CIRRDataset, base_path
    import os
    classic = CIRRDataset("test1", "classic", preprocess)
    relative = CIRRDataset("test1", "relative", preprocess)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and rerank:
        # the stage-2 rerank is not sharded: every rank would encode the whole gallery, rerank every pair and write the same two
        # files at once.  Refuse instead of doing that silently.
        raise SystemExit("cirr_test_submission: --rerank runs on one process (launch without torchrun); the sharded path "
                         "covers the stage-1 submission only")
    if world > 1:                                                       # torchrun: gallery sharded over the ranks
        from .dist_eval import generate_cirr_test_dicts_sharded
        top, sub = generate_cirr_test_dicts_sharded(relative, classic, blip_model, txt_processors)
        if int(os.environ.get("RANK", "0")) != 0:
            return
    else:
        feats, names = extract_index_blip_features(classic, blip_model, num_workers=num_workers)
        top, sub = generate_cirr_test_dicts(relative, blip_model, feats, names, txt_processors, rerank)
    submission = {"version": "rc2", "metric": "recall", **top}
    group_submission = {"version": "rc2", "metric": "recall_subset", **sub}
    folder = base_path / "submission" / "CIRR"
    folder.mkdir(exist_ok=True, parents=True)
    print("Saving CIRR test predictions")
    with open(folder / f"recall_submission_{file_name}.json", "w+") as f:
        json.dump(submission, f, sort_keys=True)
    with open(folder / f"recall_subset_submission_{file_name}.json", "w+") as f:
        json.dump(group_submission, f, sort_keys=True)


def main(argv=None):
    from .blip_validate import _preprocess
    p = ArgumentParser()
    p.add_argument("--blip-model-name", default="blip2_cir_align_prompt", type=str)
    p.add_argument("--model-path", type=str)
    p.add_argument("--backbone", type=str, default="pretrain", help="pretrain for vit-g, pretrain_vitL for vit-l")
    p.add_argument("--rerank", type=lambda v: str(v).lower() in ("yes", "true", "t", "y", "1"), default=False)
    p.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--gpu-preprocess", action="store_true", help="image transform on the GPU (bit-identical to the PIL transform)")
    p.add_argument("--vit-depth", type=int, default=None, help="truncate the ViT to N blocks (entry-point smoke tests only)")
    a = p.parse_args(argv)
    model, txt = _load(a.blip_model_name, a.backbone, a.model_path, a.dtype, a.vit_depth)
    preprocess, workers = _preprocess(a.gpu_preprocess, model.device)
    generate_cirr_test_submissions(f"{a.blip_model_name}_2", model, preprocess, txt, a.rerank, workers)


if __name__ == "__main__":
    main()
