#!/usr/bin/env python3
"""Config C2 end to end through the ENTRY POINT: a synthetic CIRR-val-sized dataset on disk (2297 PNG images of assorted sizes,
4181 triplets, the reference's directory / JSON layout), a full-depth ViT-g checkpoint file with the reference's key layout, then
`python -m sprc_amd.blip_validate --dataset CIRR` as the reference's script is run (src/blip_validate.py:103-155): image
decoding, the transform, gallery encoding, query fusion, ranking, metrics.  Prints the wall-clock of each variant.
Usage (GPU box): tools/c2_e2e.py [n_images n_queries]      -- writes under $TMPDIR/sprc_c2"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

N_IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 2297
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 4181
def synth_image(w: int, h: int, seed: int) -> np.ndarray:
    """uint8 RGB [h, w, 3]: smooth gradients + blocks + noise (enough structure for the bicubic transform to matter)"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(x / (7 + seed % 5) + c) * np.cos(y / (11 + c)) for c in range(3)], axis=-1)
    img += rng.normal(0, 12, size=img.shape)
    bx, by = int(rng.integers(0, max(w - 40, 1))), int(rng.integers(0, max(h - 40, 1)))
    img[by:by + 40, bx:bx + 40] = rng.integers(0, 256, size=3)
    return np.clip(img, 0, 255).astype(np.uint8)


WORDS = ["make", "the", "dog", "bigger", "remove", "person", "add", "a", "red", "hat", "and", "two", "cats", "instead", "of", "one",
         "brighter", "background", "is", "more", "colour", "##ful", "##s", "##er", "left", "right"]


def build(root: Path):
    cirr = root / "cirr_dataset" / "cirr"
    (cirr / "captions").mkdir(parents=True, exist_ok=True)
    (cirr / "image_splits").mkdir(parents=True, exist_ok=True)
    (root / "cirr_dataset" / "img").mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(3)
    sizes = [(500, 375), (300, 600), (224, 224), (640, 200), (97, 301), (333, 333), (260, 190), (480, 640)]
    protos = [synth_image(w, h, 40 + i) for i, (w, h) in enumerate(sizes)]
    names = []
    for i in range(N_IMG):
        arr = np.roll(protos[i % len(protos)], i * 7, axis=1)          # distinct pixels per file, cheap to make
        name = f"val-{i:05d}"
        Image.fromarray(arr).save(root / "cirr_dataset" / "img" / f"{name}.png", compress_level=1)
        names.append(name)
    split = {n: f"img/{n}.png" for n in names}
    trip = []
    for q in range(NQ):
        ref = int(rng.integers(0, N_IMG))
        tgt = int((ref + 1 + rng.integers(0, N_IMG - 1)) % N_IMG)
        others = [int(i) for i in rng.choice(N_IMG, size=8, replace=False) if i not in (ref, tgt)][:4]
        members = [names[i] for i in rng.permutation([ref, tgt, *others])]
        cap = " ".join(rng.choice(WORDS[:20], size=int(rng.integers(2, 9))).tolist()).capitalize() + "."
        trip.append({"pairid": 100 + q, "reference": names[ref], "target_hard": names[tgt], "caption": cap, "img_set": {"members": members}})
    for sp in ("val", "test1"):                             # the same files serve as the test1 split (config C4's sizes are alike)
        (cirr / "captions" / f"cap.rc2.{sp}.json").write_text(json.dumps(trip))
        (cirr / "image_splits" / f"split.rc2.{sp}.json").write_text(json.dumps(split))
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + [".", ","]
    vocab += [f"tok{i}" for i in range(30522 - len(vocab))]
    (root / "vocab.txt").write_text("\n".join(vocab) + "\n")
    cfg = get_config("pretrain")
    sd = synth.make_state_dict(cfg, seed=17, planted=True)
    torch.save({"Blip2QformerCirAlignPrompt": sd, "epoch": 0}, root / "ckpt.pt")


def breakdown(root: Path):
    """Where a real-data gallery pass spends its time: decode alone (the loader workers, no GPU work), the whole pass, and the engine
    alone on resident tensors."""
    from torch.utils.data import DataLoader
    from sprc_amd import harness as H
    from sprc_amd.blip_validate import _load, _preprocess
    from sprc_amd.data_utils import CIRRDataset
    model, _ = _load("blip2_cir_align_prompt", "pretrain", str(root / "ckpt.pt"), "fp16", None)
    preprocess, workers = _preprocess(True, model.device)
    ds = CIRRDataset("val", "classic", preprocess)
    tf, dec = H._decode_only(ds)
    nw = max(2, min(12, H.usable_cores() - 2))
    t = time.perf_counter()
    n = 0
    for names, imgs in H._ThreadLoader(dec, 128, nw):
        n += len(names)
    t_dec = time.perf_counter() - t
    H.extract_index_blip_features(ds, model, num_workers=workers, keep_raw=False)           # warm-up: engine build, workspaces
    torch.cuda.synchronize()
    t = time.perf_counter()
    (feats, _), names = H.extract_index_blip_features(ds, model, num_workers=workers, keep_raw=False)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    x = torch.randn((128, 3, 224, 224), device=model.device)
    for _ in range(2):
        model.extract_target_features(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        model.extract_target_features(x)
    torch.cuda.synchronize()
    t_enc = (time.perf_counter() - t) / 10
    print(f"[c2_e2e breakdown] {H.usable_cores()} usable cores, {nw} decode threads: decode alone {n / t_dec:.0f} img/s "
          f"({t_dec:.2f} s for {n} PNG files); whole gallery pass {len(names) / t_all:.0f} img/s ({t_all:.2f} s); engine alone on resident "
          f"tensors {128 / t_enc:.0f} img/s", flush=True)


def main():
    root = Path(os.environ.get("TMPDIR", "/tmp")) / "sprc_c2"
    t0 = time.perf_counter()
    build(root)
    print(f"dataset + checkpoint written in {time.perf_counter() - t0:.1f} s: {N_IMG} images, {NQ} queries under {root}", flush=True)
    os.environ["SPRC_DATA_ROOT"], os.environ["SPRC_BERT_VOCAB"] = str(root), str(root / "vocab.txt")
    from sprc_amd import blip_validate as bv
    if os.environ.get("SPRC_C2_BREAKDOWN", "1") == "1":
        breakdown(root)
    base = ["--dataset", "CIRR", "--model-path", str(root / "ckpt.pt")]
    out = {}
    for tag, extra in (("PIL transform in loader workers", []), ("GPU transform", ["--gpu-preprocess"]),
                       ("GPU transform, gallery from the feature store", ["--gpu-preprocess", "--index-cache", str(root / "index")]),
                       ("GPU transform, gallery from the feature store (second run)", ["--gpu-preprocess", "--index-cache", str(root / "index")])):
        torch.cuda.synchronize()
        t = time.perf_counter()
        m = bv.main(base + extra)
        torch.cuda.synchronize()
        out[tag] = time.perf_counter() - t
        print(f"[c2_e2e] {tag}: {out[tag]:.1f} s   R@1 {m['recall_at1']:.2f} R@10 {m['recall_at10']:.2f} Rs@1 {m['group_recall_at1']:.2f}", flush=True)
    # the test-submission entry point (cirr_test_submission.py:203-222), without and with the stage-2 rerank of the top-50
    from sprc_amd import cirr_test_submission as cts
    for tag, extra in (("test submission", []), ("test submission + rerank", ["--rerank", "true"])):
        torch.cuda.synchronize()
        t = time.perf_counter()
        cts.main(["--model-path", str(root / "ckpt.pt"), "--gpu-preprocess", *extra])
        torch.cuda.synchronize()
        out[tag] = time.perf_counter() - t
        print(f"[c2_e2e] {tag}: {out[tag]:.1f} s", flush=True)
    print(json.dumps({"n_images": N_IMG, "n_queries": NQ, "seconds": out}))


if __name__ == "__main__":
    main()
