"""Host-side datasets and preprocessing for the evaluation entry points (image decode is host I/O; SURVEY.md
section 8(f) N3 keeps GPU preprocessing as a "next" row).

Same on-disk layout, dataset modes and item tuples as the reference's src/data_utils.py (CIRR :203-286,
FashionIQ :108-200, TargetPad + CLIP normalisation :49-105) so the reference's data directory works as is;
the root is `SPRC_DATA_ROOT` (default: the parent of this repository, like the reference's `base_path`).
torchvision is not required: the transform is PIL + torch.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Callable, List

import numpy as np
import torch
from torch.utils.data import Dataset

base_path = Path(os.environ.get("SPRC_DATA_ROOT", Path(__file__).resolve().parents[2]))

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class TargetPad:
    """Zero-pad towards `target_ratio` when the aspect ratio exceeds it (data_utils.py:49-72)."""

    def __init__(self, target_ratio: float, size: int):
        self.target_ratio, self.size = target_ratio, size

    def __call__(self, image):
        from PIL import ImageOps
        w, h = image.size
        if max(w, h) / min(w, h) < self.target_ratio:
            return image
        scaled = max(w, h) / self.target_ratio
        hp, vp = max(int((scaled - w) / 2), 0), max(int((scaled - h) / 2), 0)
        return ImageOps.expand(image, border=(hp, vp, hp, vp), fill=0)


class HostTargetPad:
    """TargetPad -> bicubic resize (short side = dim) -> centre crop -> RGB -> [0,1] CHW -> CLIP normalise
    (data_utils.py:91-105) with PIL on the host.  A class (not a closure) so that loader workers started from a fork server /
    by spawn can unpickle it."""

    def __init__(self, target_ratio: float, dim: int):
        self.pad, self.dim = TargetPad(target_ratio, dim), dim

    def __call__(self, image):
        from PIL import Image
        dim = self.dim
        image = self.pad(image)
        w, h = image.size
        if w <= h:
            nw, nh = dim, int(dim * h / w)
        else:
            nw, nh = int(dim * w / h), dim
        image = image.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - dim) / 2.0)), int(round((nh - dim) / 2.0))
        image = image.crop((left, top, left + dim, top + dim)).convert("RGB")
        x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        mean, std = torch.tensor(CLIP_MEAN).view(3, 1, 1), torch.tensor(CLIP_STD).view(3, 1, 1)
        return (x - mean) / std


# Image modes for which "convert to RGB, then pad / resize / crop" (what the GPU transform does) equals the reference's order
# "pad / resize / crop in the image's own mode, convert last" (data_utils.py:49-105: TargetPad and Resize run before
# _convert_image_to_rgb): RGB trivially, L because grey -> RGB replicates the channel and commutes with a per-channel resampler.
# Palette ("P") and bilevel ("1") images are resized with NEAREST by PIL and padded with palette index 0; RGBA / LA are resampled
# with premultiplied alpha; 16-bit and float modes have their own paths -- those go through the PIL transform on the host.
GPU_EXACT_MODES = ("RGB", "L")


def targetpad_transform(target_ratio: float, dim: int) -> Callable:
    return HostTargetPad(target_ratio, dim)


class GpuTargetPad:
    """`targetpad_transform(target_ratio, dim)` with the pixel work on the GPU (sprc_preprocess_targetpad): the host only
    decodes (PIL) and hands over the uint8 RGB image; pad / bicubic resize / centre crop / normalise run as two HIP
    kernels and are bit-identical to the PIL path above (tests/test_preprocess.py).  Returns a [3, dim, dim] fp32 tensor
    on `device`.  Use with DataLoader(num_workers=0): the transform touches the GPU."""

    on_device = True        # items are CUDA tensors: loaders must not pin them, and must run in the main process

    def __init__(self, target_ratio: float, dim: int, device="cuda"):
        import ctypes as C
        from . import _lib as L
        self.L, self.C = L, C
        self.lib = L.load()
        self.ratio, self.dim, self.device = float(target_ratio), int(dim), torch.device(device)
        if self.device.type != "cuda":
            raise L.SprcError("GpuTargetPad needs a GPU device (the PIL transform `targetpad_transform` is the host path)")
        self.mean = (C.c_float * 3)(*CLIP_MEAN)
        self.std = (C.c_float * 3)(*CLIP_STD)
        self._ws = None
        self._host = HostTargetPad(self.ratio, self.dim)

    def __call__(self, image) -> torch.Tensor:
        if not isinstance(image, (np.ndarray, torch.Tensor)):
            if image.mode not in GPU_EXACT_MODES:                           # see GPU_EXACT_MODES: PIL's own per-mode behaviour
                return self._host(image).to(self.device, non_blocking=True)
            image = np.asarray(image.convert("RGB"), dtype=np.uint8)       # _convert_image_to_rgb (data_utils.py:74-75)
        src = torch.from_numpy(np.array(image, dtype=np.uint8, order="C")) if isinstance(image, np.ndarray) else image.contiguous()
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
            raise ValueError("expected a uint8 RGB image [H, W, 3]")
        h, w = int(src.shape[0]), int(src.shape[1])
        src = src.to(self.device, non_blocking=True)
        need = int(self.lib.sprc_preprocess_workspace_bytes(h, w, self.ratio, self.dim))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((3, self.dim, self.dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            self.L.check(self.lib.sprc_preprocess_targetpad(src.data_ptr(), h, w, w * 3, self.ratio, self.dim, self.mean, self.std,
                                                            out.data_ptr(), self._ws.data_ptr(), self._ws.numel(), st),
                         "sprc_preprocess_targetpad")
        self._keep = src                                                    # the kernels read it asynchronously
        return out


class DecodeRGB:
    """The host half of `GpuTargetPad`: PIL image -> uint8 RGB tensor [H, W, 3] (_convert_image_to_rgb, data_utils.py:74-75).
    What loader WORKERS run when the pixel work is on the GPU (sprc_amd/harness.py: extract_index_blip_features).
    Images in a mode the GPU transform does not reproduce (GPU_EXACT_MODES) are transformed here, on the host, and come out as
    the finished fp32 [3, dim, dim] tensor instead (the harness passes those through: `is_transformed`)."""

    def __init__(self, target_ratio: float = 1.25, dim: int = 224):
        self._host = HostTargetPad(target_ratio, dim)

    def __call__(self, image) -> torch.Tensor:
        if image.mode not in GPU_EXACT_MODES:
            return self._host(image)
        return torch.from_numpy(np.array(image.convert("RGB"), dtype=np.uint8, order="C"))


def is_transformed(item: torch.Tensor) -> bool:
    """True for a finished [3, dim, dim] fp32 item (host transform), False for a decoded uint8 [H, W, 3] image."""
    return item.dtype == torch.float32


def targetpad_transform_gpu(target_ratio: float, dim: int, device="cuda") -> Callable:
    return GpuTargetPad(target_ratio, dim, device)


class _Base(Dataset):
    def __getitem__(self, index):
        try:
            return self._get(index)
        except Exception as e:           # the reference swallows per-item errors and yields None (collate_fn drops it)
            print(f"Exception: {e}")
            return None


class CIRRDataset(_Base):
    """'classic': (name, image); 'relative': val -> (ref_name, target_name, caption, members),
    test1 -> (pair_id, ref_name, caption, members)."""

    def __init__(self, split: str, mode: str, preprocess: Callable):
        if split not in ("test1", "train", "val"):
            raise ValueError("split should be in ['test1', 'train', 'val']")
        if mode not in ("relative", "classic"):
            raise ValueError("mode should be in ['relative', 'classic']")
        self.split, self.mode, self.preprocess = split, mode, preprocess
        root = base_path / "cirr_dataset" / "cirr"
        self.triplets = json.loads((root / "captions" / f"cap.rc2.{split}.json").read_text())
        self.name_to_relpath = json.loads((root / "image_splits" / f"split.rc2.{split}.json").read_text())
        self.names = list(self.name_to_relpath.keys())
        print(f"CIRR {split} dataset in {mode} mode initialized")

    def _image(self, name):
        from PIL import Image
        return self.preprocess(Image.open(base_path / "cirr_dataset" / self.name_to_relpath[name]))

    def _get(self, index):
        if self.mode == "classic":
            name = self.names[index]
            return name, self._image(name)
        t = self.triplets[index]
        members, ref, cap = t["img_set"]["members"], t["reference"], t["caption"]
        if self.split == "val":
            return ref, t["target_hard"], cap, members
        if self.split == "test1":
            return t["pairid"], ref, cap, members
        return self._image(ref), self._image(t["target_hard"]), cap

    def __len__(self):
        return len(self.triplets) if self.mode == "relative" else len(self.names)


class FashionIQDataset(_Base):
    """'classic': (name, image); 'relative': val -> (ref_name, target_name, [cap1, cap2])."""

    def __init__(self, split: str, dress_types: List[str], mode: str, preprocess: Callable):
        if mode not in ("relative", "classic"):
            raise ValueError("mode should be in ['relative', 'classic']")
        if split not in ("test", "train", "val"):
            raise ValueError("split should be in ['test', 'train', 'val']")
        for d in dress_types:
            if d not in ("dress", "shirt", "toptee"):
                raise ValueError("dress_type should be in ['dress', 'shirt', 'toptee']")
        self.split, self.mode, self.dress_types, self.preprocess = split, mode, dress_types, preprocess
        root = base_path / "fashionIQ_dataset"
        self.triplets, self.image_names = [], []
        for d in dress_types:
            self.triplets += json.loads((root / "captions" / f"cap.{d}.{split}.json").read_text())
            self.image_names += json.loads((root / "image_splits" / f"split.{d}.{split}.json").read_text())
        print(f"FashionIQ {split} - {dress_types} dataset in {mode} mode initialized")

    def _image(self, name):
        from PIL import Image
        return self.preprocess(Image.open(base_path / "fashionIQ_dataset" / "images" / f"{name}.png"))

    def _get(self, index):
        if self.mode == "classic":
            name = self.image_names[index]
            return name, self._image(name)
        t = self.triplets[index]
        if self.split == "val":
            return t["candidate"], t["target"], t["captions"]
        if self.split == "test":
            return t["candidate"], self._image(t["candidate"]), t["captions"]
        return self._image(t["candidate"]), self._image(t["target"]), t["captions"]

    def __len__(self):
        return len(self.triplets) if self.mode == "relative" else len(self.image_names)
