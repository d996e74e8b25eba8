"""CPU restatement (TEST INFRASTRUCTURE) of the reference's image transform, `targetpad_transform(1.25, 224)`
(/root/reference/src/data_utils.py:49-72 TargetPad, :91-105 Compose): TargetPad -> Resize(dim, BICUBIC) ->
CenterCrop(dim) -> RGB -> ToTensor -> Normalize(CLIP mean/std).

torchvision's `Resize` / `CenterCrop` on a PIL image are PIL's own `Image.resize` / `Image.crop` (torchvision is not
installed here: SURVEY.md section 8(c)), so the arithmetic to reproduce is PIL's 8-bit resampler (libImaging/Resample.c,
Pillow 12.2.0 installed here; algorithm unchanged since Pillow 3.x):

  * per output coordinate: centre = (x + 0.5) * scale, support = 2 * max(scale, 1) (bicubic, a = -0.5), taps
    [xmin, xmax) = [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image, weights
    bicubic((x' + xmin - centre + 0.5) / max(scale, 1)) normalised to sum 1 -- all in double;
  * weights are converted to 22-bit fixed point (round half away from zero), a pixel is
    clip8((2^21 + sum_k pixel_k * w_k) >> 22);
  * two passes, HORIZONTAL first; the intermediate image is uint8 (rounded and clipped), then the vertical pass.

This module is pinned against PIL itself in tests/test_preprocess.py (bit-exact on random images of many shapes) and is
the checker of the HIP preprocessing kernel; it is never imported by the product.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (xmin[out], count[out], kk[out, ksize] int32 fixed point): Resample.c precompute_coeffs + normalize_coeffs_8bpc."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int32)
    cnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.array([_bicubic((x + lo - center + 0.5) * ss) for x in range(n)], dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        xmin[xx], cnt[xx] = lo, n
        kk[xx, :n] = fixed
    return xmin, cnt, kk


def _pass(img: np.ndarray, xmin, cnt, kk, axis: int) -> np.ndarray:
    """one resampling pass of a uint8 [H, W, C] image along `axis` (1 = horizontal, 0 = vertical)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(xmin),) + src.shape[1:], dtype=np.uint8)
    for i in range(len(xmin)):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[i, :cnt[i]].astype(np.int64), src[xmin[i]:xmin[i] + cnt[i]], axes=(0, 0))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL `Image.resize((out_w, out_h), BICUBIC)` on a uint8 [H, W, C] array (horizontal pass, then vertical)."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        out = _pass(out, *resample_coeffs(w, out_w), axis=1)
    if out_h != h:
        out = _pass(out, *resample_coeffs(h, out_h), axis=0)
    return out


def targetpad_geometry(w: int, h: int, target_ratio: float, dim: int):
    """-> (pad_x, pad_y, padded_w, padded_h, resized_w, resized_h, crop_left, crop_top): data_utils.py:62-72 (TargetPad),
    torchvision Resize(int) (short side -> dim, long side int(dim * long / short)), CenterCrop (int(round((s - dim) / 2)))."""
    hp = vp = 0
    if max(w, h) / min(w, h) >= target_ratio:
        scaled = max(w, h) / target_ratio
        hp, vp = max(int((scaled - w) / 2), 0), max(int((scaled - h) / 2), 0)
    pw, ph = w + 2 * hp, h + 2 * vp
    if pw <= ph:
        rw, rh = dim, int(dim * ph / pw)
    else:
        rw, rh = int(dim * pw / ph), dim
    return hp, vp, pw, ph, rw, rh, int(round((rw - dim) / 2.0)), int(round((rh - dim) / 2.0))


def targetpad_transform(img: np.ndarray, target_ratio: float = 1.25, dim: int = 224) -> np.ndarray:
    """uint8 RGB [H, W, 3] -> float32 [3, dim, dim], the reference's `targetpad_transform` (data_utils.py:91-105)."""
    h, w = img.shape[:2]
    hp, vp, pw, ph, rw, rh, left, top = targetpad_geometry(w, h, target_ratio, dim)
    padded = np.zeros((ph, pw, 3), dtype=np.uint8)
    padded[vp:vp + h, hp:hp + w] = img
    r = resize_bicubic_u8(padded, rw, rh)[top:top + dim, left:left + dim]
    x = r.astype(np.float32) / np.float32(255.0)                              # ToTensor
    mean, std = np.asarray(CLIP_MEAN, dtype=np.float32), np.asarray(CLIP_STD, dtype=np.float32)
    return ((x - mean) / std).transpose(2, 0, 1).copy()                      # Normalize, CHW
