"""Shared, session-lived test inputs for the GPU tier (the tier ran 870 s of a 1200-s limit).

Most of a full-depth parity test used to be `synth.make_state_dict` -- 1.17 G seeded parameters drawn on the CPU, ~8 s on the GPU box --
and the same (backbone, seed, planted, trunk_fp16) draw was repeated by every test and module that compares an engine with the same
golden (36 draws per tier for 16 distinct ones).  `state_dict` draws each FULL-DEPTH dict once per session and keeps it on the GPU
(the CPU draw moved over: bit-identical to what the golden's generator fed the reference; 4.7 GB per ViT-g dict, ~60 GB of the 288 GB in
total); shallow configurations (depth < 20: their tests also run the CPU oracle on the dict) are drawn fresh, on the CPU, as before.
`planted_case` adds the case's images.  `prefetch` (called by conftest.py once the GPU tier is collected) draws the tier's full-depth dicts
on two background threads, in the order the test files use them, so that the CPU draws run UNDER the GPU-side tests instead of between
them.  Nothing here is product code."""
from __future__ import annotations

import threading
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from sprc_amd import synth
from sprc_amd.config import get_config

_SD = {}
_IMG = {}
_LOCKS = defaultdict(threading.Lock)
_GUARD = threading.Lock()
_POOL = None
# the tier's full-depth draws in the order the test files (alphabetical) first use them: goldens by file name, then (seed, planted, h16)
# of the C2-size cases on the 39-block ViT-g
PREFETCH_GOLDENS = ["full_eva", "full_clip", "planted_full_eva", "planted_full_clip", "planted_full_eva_s1", "planted_full_clip_s1",
                    "planted_big_eva", "planted_big_eva_s3", "planted_full_eva_h16", "planted_full_clip_h16", "planted_big_eva_h16"]
PREFETCH_C2 = [(5, True, False), (5, True, True), (7, True, True)]


def state_dict(cfg, seed: int, planted: bool = False, trunk_fp16: bool = False):
    if cfg.vit.depth < 20 or not torch.cuda.is_available():
        return synth.make_state_dict(cfg, seed=seed, planted=planted, trunk_fp16=trunk_fp16)
    key = (cfg, int(seed), bool(planted), bool(trunk_fp16))
    with _GUARD:
        lock = _LOCKS[key]
    with lock:                                            # a test that needs a dict a prefetch thread is drawing waits for THAT draw
        if key not in _SD:
            sd = synth.make_state_dict(cfg, seed=seed, planted=planted, trunk_fp16=trunk_fp16)
            _SD[key] = {k: v.to("cuda:0") for k, v in sd.items()}
    return dict(_SD[key])


def prefetch(golden_dir) -> None:
    """start drawing the tier's full-depth state dicts in the background (idempotent; two threads: the draws are single-threaded torch.randn)"""
    global _POOL
    if _POOL is not None or not torch.cuda.is_available():
        return
    keys = []
    for name in PREFETCH_GOLDENS:
        path = golden_dir / f"{name}.npz"
        if path.exists():
            g = np.load(path, allow_pickle=False)
            cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
            keys.append((cfg, int(g["seed"]), name.startswith("planted"), bool(int(g["trunk_fp16"])) if "trunk_fp16" in g.files else False))
    keys += [(get_config("pretrain"), s, p, h) for s, p, h in PREFETCH_C2]
    _POOL = ThreadPoolExecutor(max_workers=2, thread_name_prefix="sd-prefetch")
    for k in keys:
        _POOL.submit(state_dict, *k)


def images(n: int, seed: int, planted: bool = False):
    key = (int(n), int(seed), bool(planted))
    if key not in _IMG:
        _IMG[key] = synth.make_images(n, seed=seed, planted=planted)
    return _IMG[key]


def planted_case(golden_dir, case: str):
    """-> (golden npz, cfg, state dict, images) of tests/golden/<case>.npz (a planted-structure case: oracle/gen_golden.py)"""
    g = np.load(golden_dir / (case if case.endswith(".npz") else f"{case}.npz"), allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    h16 = bool(int(g["trunk_fp16"])) if "trunk_fp16" in g.files else False
    sd = state_dict(cfg, int(g["seed"]), planted=True, trunk_fp16=h16)
    img = images(int(g["n_img"]), int(g["seed"]), planted=True)
    np.testing.assert_array_equal(img[:4, :, 0, :4].numpy(), g["image_probe"])
    return g, cfg, sd, img


def stop_prefetch() -> None:
    """drop the draws that have not started (session end: a subset run must not wait for dicts nobody will use)"""
    global _POOL
    if _POOL is not None:
        _POOL.shutdown(wait=False, cancel_futures=True)
        _POOL = None
