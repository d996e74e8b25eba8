"""Feature store for an encoded gallery ("feature-store format for feats ... + names").

The reference re-encodes the whole gallery on every run (`src/utils.py:46-77` returns tensors that nobody persists;
`save_memory` only moves them to the CPU, `:67-69`).  A gallery is encoded once here and kept as ONE safetensors file:

    feats      [N, 32, 256]  fp32   unit-norm Q-Former query features (what `inference` ranks against)
    raw        [N, 257, D]   fp32, or the engine's 16-bit operand dtype (`raw_dtype`)
                                    optional: ViT embeddings of the images that can be *reference* images of a query
                                    (CIRR/FashionIQ take references from the gallery itself); omitted for pure galleries.
                                    fp16 / bf16: the format the engine rounds them to anyway when it reads them (the K|V
                                    projection operand of the fusion pass and of the rerank): half the bytes, the same bits
                                    downstream (the rounding is idempotent); load_index hands them back as fp32
    metadata   names (JSON list, row order), backbone, dtype the features were computed in, format version,
               checkpoint_sha256 = fingerprint of the state dict that produced the features (blip_validate refuses a store
               written by another checkpoint)

Loading is a plain mmap + one host-to-device copy; ranking from a loaded store is bit-identical to ranking from the
tensors it was saved from (tests/test_host.py, tests/test_fullsize_gpu.py).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import torch

FORMAT = "sprc-index-1"


def save_index(path, feats: torch.Tensor, names: Sequence[str], raw: Optional[torch.Tensor] = None, backbone: str = "",
               compute_dtype: str = "", checkpoint_sha256: str = "", raw_dtype: torch.dtype = torch.float32) -> None:
    from safetensors.torch import save_file
    if feats.dim() != 3 or feats.shape[1] != 32:
        raise ValueError(f"feats must be [N,32,E], got {tuple(feats.shape)}")
    if len(names) != feats.shape[0] or (raw is not None and raw.shape[0] != feats.shape[0]):
        raise ValueError("names / raw must have one entry per gallery row")
    if len(set(names)) != len(names):
        raise ValueError("gallery names must be unique (they are the join key of the relative datasets)")
    tensors = {"feats": feats.detach().to("cpu", torch.float32).contiguous()}
    if raw is not None:
        if raw_dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise ValueError(f"raw_dtype {raw_dtype}")
        tensors["raw"] = raw.detach().to(raw_dtype).to("cpu").contiguous()          # rounded where the tensor lives (the GPU), then copied
    meta = {"format": FORMAT, "names": json.dumps(list(names)), "backbone": backbone, "compute_dtype": compute_dtype,
            "checkpoint_sha256": checkpoint_sha256}
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    save_file(tensors, str(path), metadata=meta)


def load_index(path, device="cpu", with_raw: bool = True) -> Tuple[Tuple[torch.Tensor, Optional[torch.Tensor]], List[str], dict]:
    """-> ((feats, raw or None), names, metadata) with the tensors on `device`: the `(index_features, index_names)` pair
    the reference's `compute_*_val_metrics` / `generate_*_predictions` take."""
    from safetensors import safe_open
    with safe_open(str(path), framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} file (format={meta.get('format')!r})")
        feats = f.get_tensor("feats")
        raw = f.get_tensor("raw") if with_raw and "raw" in f.keys() else None
    names = json.loads(meta["names"])
    if len(names) != feats.shape[0]:
        raise ValueError(f"{path}: {len(names)} names for {feats.shape[0]} rows")
    feats = feats.to(device)
    raw = raw.to(device).float() if raw is not None else None
    return (feats, raw), names, {k: v for k, v in meta.items() if k != "names"}
