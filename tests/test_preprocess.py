"""Image preprocessing: the reference's targetpad_transform(1.25, 224).
CPU: the oracle restatement of PIL's 8-bit bicubic resampler is pinned bit-exactly against PIL itself and against the
committed fixture (tests/golden/preprocess.json, generated with the PIL calls the reference's torchvision Compose makes);
the host transform of sprc_amd/data_utils.py equals it too.  GPU: the HIP kernels (sprc_preprocess_targetpad) reproduce the
same bits."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))

from oracle import preprocess_oracle as P  # noqa: E402
from gen_preprocess_golden import synth_image  # noqa: E402


def _cases(golden_dir):
    return json.loads((golden_dir / "preprocess.json").read_text())["cases"]


def test_oracle_resampler_is_pil_bit_for_bit():
    rng = np.random.default_rng(5)
    for (w, h, ow, oh) in [(500, 375, 298, 224), (97, 301, 224, 695), (1000, 250, 896, 224), (224, 224, 224, 224), (37, 41, 224, 248),
                           (640, 480, 100, 50)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        np.testing.assert_array_equal(P.resize_bicubic_u8(img, ow, oh), want)


def test_oracle_and_host_transform_match_the_fixture(golden_dir):
    from sprc_amd.data_utils import targetpad_transform
    tf = targetpad_transform(1.25, 224)
    for c in _cases(golden_dir):
        img = synth_image(c["w"], c["h"], c["seed"])
        got = P.targetpad_transform(img)
        assert got.shape == (3, 224, 224) and got.dtype == np.float32
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], (c["w"], c["h"])
        np.testing.assert_allclose(got[:, ::37, ::41], np.asarray(c["sample"], dtype=np.float32), atol=1e-6, rtol=0)
        host = tf(Image.fromarray(img)).numpy()
        assert np.array_equal(host, got), (c["w"], c["h"])


def test_geometry_matches_the_reference_rules():
    # no padding below the ratio; pad towards 1.25 above it; resize short side to 224; centre crop with Python rounding
    assert P.targetpad_geometry(500, 375, 1.25, 224)[:2] == (0, 12)             # 1.333 >= 1.25: 400 = 500 / 1.25 rows
    hp, vp, pw, ph, rw, rh, left, top = P.targetpad_geometry(1000, 200, 1.25, 224)
    assert (hp, vp) == (0, 300) and (pw, ph) == (1000, 800) and (rw, rh) == (280, 224) and (left, top) == (28, 0)
    assert P.targetpad_geometry(224, 224, 1.25, 224) == (0, 0, 224, 224, 224, 224, 0, 0)
    assert P.targetpad_geometry(300, 300 + 70, 1.25, 224)[:2] == (0, 0)          # ratio 1.233 < 1.25: untouched


@pytest.mark.gpu
def test_gpu_preprocessing_is_bit_identical(golden_dir):
    from sprc_amd.data_utils import GpuTargetPad
    tf = GpuTargetPad(1.25, 224, "cuda:0")
    for c in _cases(golden_dir):
        img = synth_image(c["w"], c["h"], c["seed"])
        got = tf(img).cpu().numpy()
        want = P.targetpad_transform(img)
        assert np.array_equal(got, want), f"{c['w']}x{c['h']}: max diff {np.abs(got - want).max()}"
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"]
    # PIL images and grayscale inputs go through _convert_image_to_rgb on the host
    gray = Image.fromarray(synth_image(260, 190, 7)[:, :, 0], mode="L")
    np.testing.assert_array_equal(tf(gray).cpu().numpy(), P.targetpad_transform(np.asarray(gray.convert("RGB"))))
    # palette / alpha / bilevel images: PIL pads and resamples them in their OWN mode (NEAREST for P and 1, premultiplied alpha for
    # RGBA / LA) before converting, which "convert first" does not reproduce -> routed through the host transform
    from sprc_amd.data_utils import HostTargetPad
    host = HostTargetPad(1.25, 224)
    base = Image.fromarray(synth_image(300, 180, 11))
    rgba = base.convert("RGBA")
    rgba.putalpha(Image.fromarray((synth_image(300, 180, 12)[:, :, 0] // 2 + 64).astype(np.uint8), mode="L"))
    for im in (base.convert("P", palette=Image.ADAPTIVE, colors=17), rgba, base.convert("LA"), base.convert("1")):
        np.testing.assert_array_equal(tf(im).cpu().numpy(), host(im).numpy(), err_msg=im.mode)
    rng = np.random.default_rng(9)
    for _ in range(12):                                   # random shapes incl. extreme aspect ratios and tiny images
        w, h = int(rng.integers(8, 900)), int(rng.integers(8, 900))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(tf(img).cpu().numpy(), P.targetpad_transform(img)), (w, h)
