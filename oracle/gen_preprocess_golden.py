#!/usr/bin/env python3
"""Generate tests/golden/preprocess.json: expected outputs of the reference's image transform targetpad_transform(1.25, 224)
(/root/reference/src/data_utils.py:91-105) on seeded synthetic images of many shapes, produced HERE with PIL (Pillow
12.2.0) -- torchvision's Resize / CenterCrop on a PIL image are PIL's own resize / crop, and torchvision is not installed
in this container.  TEST INFRASTRUCTURE; only data is written: per case the image recipe
(size, seed), the sha256 of the float32 [3,224,224] result and a strided sample of it."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
from PIL import Image, ImageOps

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CASES = [(500, 375), (375, 500), (300, 600), (1000, 200), (224, 224), (225, 224), (97, 301), (1024, 768), (333, 333), (640, 480),
         (200, 1000), (226, 181), (180, 224)]
MEAN = np.array((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
STD = np.array((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)


def synth_image(w: int, h: int, seed: int) -> np.ndarray:
    """smooth structure + noise, uint8 RGB [h, w, 3]"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 90 * np.sin(xx / (7.0 + c) + yy / (11.0 - c)) + rng.normal(0, 25, (h, w)) for c in range(3)], axis=-1)
    return np.clip(img, 0, 255).astype(np.uint8)


def reference_transform(img: np.ndarray, ratio: float = 1.25, dim: int = 224) -> np.ndarray:
    """the reference's Compose, written with the PIL calls torchvision makes (data_utils.py:62-72, :97-104)"""
    im = Image.fromarray(img)
    w, h = im.size
    if max(w, h) / min(w, h) >= ratio:
        scaled = max(w, h) / ratio
        hp, vp = max(int((scaled - w) / 2), 0), max(int((scaled - h) / 2), 0)
        im = ImageOps.expand(im, border=(hp, vp, hp, vp), fill=0)                       # F.pad(image, [hp, vp, hp, vp], 0, 'constant')
    w, h = im.size
    size = (dim, int(dim * h / w)) if w <= h else (int(dim * w / h), dim)               # Resize(dim): short side -> dim
    im = im.resize(size, Image.BICUBIC)
    left, top = int(round((size[0] - dim) / 2.0)), int(round((size[1] - dim) / 2.0))   # CenterCrop(dim)
    im = im.crop((left, top, left + dim, top + dim)).convert("RGB")
    x = np.asarray(im, dtype=np.uint8).astype(np.float32) / np.float32(255.0)          # ToTensor
    return ((x - MEAN) / STD).transpose(2, 0, 1).copy()                                 # Normalize


def main():
    out = []
    for i, (w, h) in enumerate(CASES):
        t = reference_transform(synth_image(w, h, 1000 + i))
        out.append({"w": w, "h": h, "seed": 1000 + i, "sha256": hashlib.sha256(t.tobytes()).hexdigest(),
                    "sample": t[:, ::37, ::41].round(6).tolist()})
    (ROOT / "tests" / "golden" / "preprocess.json").write_text(json.dumps({"pillow": Image.__version__, "cases": out}))
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    import PIL
    Image.__version__ = PIL.__version__
    main()
