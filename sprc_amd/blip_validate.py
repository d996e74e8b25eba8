#!/usr/bin/env python3
"""`python -m sprc_amd.blip_validate --dataset CIRR --blip-model-name blip2_cir_align_prompt --model-path X`

Entry point with the reference's flags and printed JSON keys (src/blip_validate.py:103-155), running the
retrieval path on the MI355X HIP engine.  Extra flag: --dtype (bf16 | fp32).
"""
from __future__ import annotations

import json
import os
import sys
from argparse import ArgumentParser
from statistics import geometric_mean, harmonic_mean, mean

import torch

from .harness import compute_cirr_val_metrics, compute_fiq_val_metrics, extract_index_blip_features
from .index import load_index, save_index
from .model import load_model_and_preprocess


def _device():
    """cuda:LOCAL_RANK, and the process group when started under torchrun (`torchrun --nproc-per-node 8 -m
    sprc_amd.blip_validate ...`: one rank per GPU, RCCL)."""
    from .dist_eval import init_from_env
    dev, _, _ = init_from_env()
    return dev


def model_fingerprint(model) -> str:
    """The identity of the checkpoint that produced a feature store: sha256 over (name, shape, dtype, two 64-bit integer
    checksums of the BITS) of every tensor of the state dict.  Features of one checkpoint must never be ranked with the queries
    of another.  The checksums -- sum of the 32-bit words and sum of word x (position mod 65521 + 1), wrapping int64 -- are
    computed where the tensor lives (integer arithmetic: exact and order-independent, the same on CPU and GPU); hashing the
    4 GB of a ViT-g state dict on the host took ~4 s of every run that named a store."""
    import hashlib
    h = hashlib.sha256()
    for k, v in sorted(model.state_dict().items()):
        t = v.detach().contiguous()
        h.update(k.encode()); h.update(str(tuple(t.shape)).encode()); h.update(str(t.dtype).encode())
        if not t.numel():
            continue
        b = t.reshape(-1).view(torch.uint8)
        if b.numel() % 4:
            b = torch.cat([b, b.new_zeros(4 - b.numel() % 4)])
        w = b.view(torch.int32).to(torch.int64)
        pos = torch.arange(w.numel(), device=w.device, dtype=torch.int64).remainder_(65521).add_(1)
        h.update(str((int(w.sum()), int((w * pos).sum()))).encode())
    return h.hexdigest()


def _raw_store_dtype(model) -> torch.dtype:
    """The dtype a store may keep raw ViT embeddings in without changing a bit downstream: the 16-bit operand format the engine
    rounds them to when it reads them (K|V projection of the fusion pass / the rerank) -- unless that projection runs on split-
    precision operands of the fp32 values (engine.X3_CKV in the query-side mask) or the engine is fp32."""
    from . import _lib as L, engine as E
    eng = model.engine()
    if eng.dt == L.SPRC_F16 and not (eng.x3_fuse & E.X3_CKV):
        return torch.float16
    return torch.bfloat16 if eng.dt == L.SPRC_BF16 else torch.float32


def _gallery(dataset, model, cache, tag, backbone, dtype, num_workers: int = 2):
    """Encoded gallery `((feats, raw), names)`: from the feature store `<cache>/<tag>-...-<checkpoint fingerprint>.safetensors`
    when it exists AND was written by this checkpoint, else encoded now (and stored when a cache directory was given).
    No reference counterpart: utils.py:46-77 re-encodes on every run."""
    if not cache:
        return extract_index_blip_features(dataset, model, num_workers=num_workers)
    fp = model_fingerprint(model)
    path = os.path.join(cache, f"{tag}-{backbone}-{dtype}-{fp[:16]}.safetensors")
    if os.path.exists(path):
        (feats, raw), names, meta = load_index(path, device=model.device)
        if meta.get("checkpoint_sha256") == fp:
            print(f"loaded {len(names)} gallery rows from {path} ({meta})")
            return (feats, raw), names
        print(f"{path}: written by another checkpoint ({meta.get('checkpoint_sha256', '?')[:16]}), re-encoding")
    (feats, raw), names = extract_index_blip_features(dataset, model, num_workers=num_workers)
    save_index(path, feats, names, raw=raw, backbone=backbone, compute_dtype=dtype, checkpoint_sha256=fp, raw_dtype=_raw_store_dtype(model))
    return (feats, raw), names


def _load(blip_model_name, backbone, model_path, dtype, vit_depth=None):
    device = _device()
    kw = {}
    if vit_depth is not None:                      # truncated backbone (smoke tests of the entry points; not a reference flag)
        from .config import get_config
        kw["cfg"] = get_config(backbone, vit_depth=vit_depth)
    model, _, txt = load_model_and_preprocess(name=blip_model_name, model_type=backbone, is_eval=False, device=device,
                                              compute_dtype=dtype, **kw)
    ckpt = torch.load(model_path, map_location=device)
    msg = model.load_state_dict(ckpt[model.__class__.__name__], strict=False)     # blip_validate.py:107-109
    print("Missing keys {}".format(msg.missing_keys))
    return model, txt


def _geometric_mean(values) -> float:
    """statistics.geometric_mean raises when a recall is exactly 0 (the reference's script then dies after the whole
    evaluation, blip_validate.py:131); report 0.0 instead."""
    return geometric_mean(values) if all(v > 0 for v in values) else 0.0


def _sharded() -> bool:
    """Started under torchrun with more than one rank: the gallery is sharded over the ranks (sprc_amd/dist_eval.py)."""
    return int(os.environ.get("WORLD_SIZE", "1")) > 1


def _rank0() -> bool:
    return int(os.environ.get("RANK", "0")) == 0


def _preprocess(gpu: bool, device):
    """(transform, DataLoader workers): the reference's targetpad_transform(1.25, 224) on the host (PIL) or with the pixel
    work on the GPU (bit-identical; loader workers then only decode, the transform runs in the main process)."""
    from .data_utils import targetpad_transform, targetpad_transform_gpu
    return (targetpad_transform_gpu(1.25, 224, device), 2) if gpu else (targetpad_transform(1.25, 224), 2)


def _warn_index_cache_ignored(index_cache):
    if index_cache and _rank0():
        print(f"warning: --index-cache {index_cache} is ignored under torchrun: every rank encodes its own gallery shard",
              file=sys.stderr)


def blip_validate_cirr(blip_model_name, backbone, blip_model_path, dtype="fp16", index_cache=None, gpu_preprocess=False,
                       vit_depth=None, reuse_reference_kv=False):
    from .data_utils import CIRRDataset
    model, txt = _load(blip_model_name, backbone, blip_model_path, dtype, vit_depth)
    preprocess, workers = _preprocess(gpu_preprocess, model.device)
    relative_val = CIRRDataset("val", "relative", preprocess)
    classic_val = CIRRDataset("val", "classic", preprocess)
    if _sharded():
        _warn_index_cache_ignored(index_cache)
        from .dist_eval import compute_cirr_val_metrics_sharded
        r = compute_cirr_val_metrics_sharded(relative_val, classic_val, model, txt, num_workers=workers)
    else:
        feats, names = _gallery(classic_val, model, index_cache, "cirr-val", backbone, dtype, workers)
        r = compute_cirr_val_metrics(relative_val, model, feats, names, txt, reuse_reference_kv=reuse_reference_kv and dtype != "fp32")
    g1, g2, g3, r1, r5, r10, r50 = r
    out = {"group_recall_at1": g1, "group_recall_at2": g2, "group_recall_at3": g3, "recall_at1": r1, "recall_at5": r5,
           "recall_at10": r10, "recall_at50": r50, "mean(R@5+R_s@1)": (g1 + r5) / 2, "arithmetic_mean": mean(r),
           "harmonic_mean": harmonic_mean(r), "geometric_mean": _geometric_mean(r)}
    if _rank0():
        print(json.dumps(out, indent=4))
    return out


def blip_validate_fiq(val_dress_types, blip_model_name, backbone, model_path, dtype="fp16", index_cache=None, gpu_preprocess=False,
                      vit_depth=None):
    """FashionIQ evaluation (the reference calls this `clip_finetune_fiq`, blip_validate.py:26-98)."""
    from .data_utils import FashionIQDataset
    model, txt = _load(blip_model_name, backbone, model_path, dtype, vit_depth)
    model.eval()
    preprocess, workers = _preprocess(gpu_preprocess, model.device)
    r10s, r50s = [], []
    for d in val_dress_types:
        classic, relative = FashionIQDataset("val", [d], "classic", preprocess), FashionIQDataset("val", [d], "relative", preprocess)
        if _sharded():
            _warn_index_cache_ignored(index_cache)
            from .dist_eval import compute_fiq_val_metrics_sharded
            r10, r50 = compute_fiq_val_metrics_sharded(relative, classic, model, txt, num_workers=workers)
        else:
            feats, names = _gallery(classic, model, index_cache, f"fiq-val-{d}", backbone, dtype, workers)
            r10, r50 = compute_fiq_val_metrics(relative, model, feats, names, txt)
        r10s.append(r10)
        r50s.append(r50)
        torch.cuda.empty_cache()
    out = {}
    for d, a, b in zip(val_dress_types, r10s, r50s):
        out[f"{d}_recall_at10"], out[f"{d}_recall_at50"] = a, b
    out.update({"average_recall_at10": mean(r10s), "average_recall_at50": mean(r50s),
                "average_recall": (mean(r50s) + mean(r10s)) / 2})
    if _rank0():
        print(json.dumps(out, indent=4))
    return out


clip_finetune_fiq = blip_validate_fiq      # the reference's (misleading) name


def main(argv=None):
    p = ArgumentParser()
    p.add_argument("--dataset", type=str, required=True, help="should be either 'CIRR' or 'fashionIQ'")
    p.add_argument("--blip-model-name", default="blip2_cir_align_prompt", type=str)
    p.add_argument("--backbone", type=str, default="pretrain", help="pretrain for vit-g, pretrain_vitL for vit-l")
    p.add_argument("--model-path", type=str)
    p.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32"],
                   help="fp16 (default): the reference's GPU numerics -- fp16 ViT, Q-Former at ~fp32 product precision")
    p.add_argument("--index-cache", default=None, help="directory of gallery feature stores (sprc_amd/index.py): encode once, reuse")
    p.add_argument("--gpu-preprocess", action="store_true", help="pad / bicubic resize / crop / normalise on the GPU (bit-identical to the PIL transform)")
    p.add_argument("--reuse-reference-kv", action="store_true", help="CIRR: project each distinct reference image to the Q-Former's cross-attention "
                   "K|V once instead of once per query (same scores; single-process, 16-bit engines)")
    p.add_argument("--vit-depth", type=int, default=None, help="truncate the ViT to N blocks (entry-point smoke tests only)")
    a = p.parse_args(argv)
    if a.dataset.lower() not in ("fashioniq", "cirr"):
        raise ValueError("Dataset should be either 'CIRR' or 'FashionIQ")
    if a.dataset.lower() == "cirr":
        return blip_validate_cirr(a.blip_model_name, a.backbone, a.model_path, a.dtype, a.index_cache, a.gpu_preprocess, a.vit_depth,
                                  reuse_reference_kv=a.reuse_reference_kv)
    return blip_validate_fiq(["dress", "toptee", "shirt"], a.blip_model_name, a.backbone, a.model_path, a.dtype, a.index_cache,
                             a.gpu_preprocess, a.vit_depth)


if __name__ == "__main__":
    main()
