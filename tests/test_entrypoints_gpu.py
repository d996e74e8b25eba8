"""The command-line entry points end to end on a tiny synthetic dataset ON DISK: CIRR-layout
directories with PNG images of assorted sizes / modes, caption and split JSON files, a checkpoint file with the reference's
key layout, a synthetic WordPiece vocabulary -- `python -m sprc_amd.blip_validate` and `python -m sprc_amd.cirr_test_submission`
run as the reference's scripts do (src/blip_validate.py:103-155, src/cirr_test_submission.py:16-58, 203-222).
Checked: the metrics equal an independent evaluation (CPU oracle + PIL transform) of the same files; the GPU image
transform and the feature-store cache change nothing; a store written by another checkpoint is refused; the submission
files have the reference's layout, with and without --rerank."""
import importlib
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))

from oracle import sprc_oracle as O  # noqa: E402
from oracle import preprocess_oracle as P  # noqa: E402
from gen_preprocess_golden import synth_image  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

N_IMG, NQ, DEPTH = 14, 9, 1
WORDS = ["make", "the", "dog", "bigger", "remove", "person", "add", "a", "red", "hat", "and", "two", "cats", "instead", "of", "one",
         "brighter", "background", "is", "more", "colour", "##ful", "##s", "##er", "left", "right"]


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    root = tmp_path_factory.mktemp("sprc_data")
    cirr = root / "cirr_dataset" / "cirr"
    (cirr / "captions").mkdir(parents=True)
    (cirr / "image_splits").mkdir(parents=True)
    (root / "cirr_dataset" / "img").mkdir(parents=True)
    rng = np.random.default_rng(3)
    sizes = [(500, 375), (300, 600), (224, 224), (640, 200), (97, 301), (333, 333), (260, 190)]
    names, arrays = [], {}
    for i in range(N_IMG):
        w, h = sizes[i % len(sizes)]
        arr = synth_image(w, h, 40 + i)
        name = f"val-{i:03d}"
        if i == 4:                                        # one grayscale file: _convert_image_to_rgb
            Image.fromarray(arr[:, :, 0], mode="L").save(root / "cirr_dataset" / "img" / f"{name}.png")
            arr = np.stack([arr[:, :, 0]] * 3, axis=-1)
        else:
            Image.fromarray(arr).save(root / "cirr_dataset" / "img" / f"{name}.png")
        names.append(name)
        arrays[name] = arr
    split = {n: f"img/{n}.png" for n in names}
    trip = []
    for q in range(NQ):
        ref = int(rng.integers(0, N_IMG))
        tgt = int((ref + 1 + rng.integers(0, N_IMG - 1)) % N_IMG)
        others = [i for i in rng.permutation(N_IMG) if i not in (ref, tgt)][:4]
        members = [names[i] for i in rng.permutation([ref, tgt, *others])]
        cap = " ".join(rng.choice(WORDS[:20], size=int(rng.integers(2, 9))).tolist()).capitalize() + "."
        trip.append({"pairid": 100 + q, "reference": names[ref], "target_hard": names[tgt], "caption": cap, "img_set": {"members": members}})
    for sp in ("val", "test1"):
        (cirr / "captions" / f"cap.rc2.{sp}.json").write_text(json.dumps(trip))
        (cirr / "image_splits" / f"split.rc2.{sp}.json").write_text(json.dumps(split))
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + [".", ","]
    vocab += [f"tok{i}" for i in range(30522 - len(vocab))]
    (root / "vocab.txt").write_text("\n".join(vocab) + "\n")
    cfg = get_config("pretrain", vit_depth=DEPTH)
    sd = synth.make_state_dict(cfg, seed=17)
    torch.save({"Blip2QformerCirAlignPrompt": sd, "epoch": 0}, root / "ckpt.pt")
    sd2 = synth.make_state_dict(cfg, seed=18)
    torch.save({"Blip2QformerCirAlignPrompt": sd2, "epoch": 0}, root / "ckpt_other.pt")
    os.environ["SPRC_DATA_ROOT"], os.environ["SPRC_BERT_VOCAB"] = str(root), str(root / "vocab.txt")
    import sprc_amd.data_utils as du
    importlib.reload(du)                                   # base_path is read at import
    return dict(root=root, names=names, arrays=arrays, trip=trip, cfg=cfg, sd=sd)


def _oracle_metrics(world):
    """independent evaluation: PIL-exact transform (oracle restatement) + fp32 CPU oracle + the oracle's metric code"""
    from sprc_amd.processors import BlipCaptionProcessor
    from sprc_amd.tokenizer import BertWordPieceTokenizer
    cfg, sd, names = world["cfg"], world["sd"], world["names"]
    images = torch.from_numpy(np.stack([P.targetpad_transform(world["arrays"][n]) for n in names]))
    tok = BertWordPieceTokenizer()
    proc = BlipCaptionProcessor()
    t = tok([proc(x["caption"]) for x in world["trip"]], padding="max_length", truncation=True, max_length=32, return_tensors="pt")
    n2i = {n: i for i, n in enumerate(names)}
    ref = np.array([n2i[x["reference"]] for x in world["trip"]])
    tgt = np.array([n2i[x["target_hard"]] for x in world["trip"]])
    grp = np.array([[n2i[m] for m in x["img_set"]["members"]] for x in world["trip"]])
    with torch.no_grad():
        feats, raw = O.extract_target_features(sd, cfg, images)
        sim = O.inference(sd, cfg, raw[torch.from_numpy(ref)], feats, t.input_ids, t.attention_mask).numpy()
    return sim, ref, tgt, grp


def test_blip_validate_cirr_on_disk(world, tmp_path, capsys):
    from sprc_amd import blip_validate as bv
    args = ["--dataset", "CIRR", "--model-path", str(world["root"] / "ckpt.pt"), "--dtype", "fp32", "--vit-depth", str(DEPTH)]
    out = bv.main(args)
    sim, ref, tgt, grp = _oracle_metrics(world)
    want = O.cirr_metrics(sim, ref, tgt, grp)
    got = (out["group_recall_at1"], out["group_recall_at2"], out["group_recall_at3"], out["recall_at1"], out["recall_at5"],
           out["recall_at10"], out["recall_at50"])
    assert got == pytest.approx(want, abs=1e-4)
    assert json.loads(capsys.readouterr().out.split("Compute CIRR validation metrics")[-1].strip())["recall_at50"] == out["recall_at50"]
    # GPU image transform: bit-identical pixels -> identical metrics
    assert bv.main(args + ["--gpu-preprocess"]) == out
    # feature store: written once, reused, refused for another checkpoint
    cache = tmp_path / "index"
    assert bv.main(args + ["--index-cache", str(cache)]) == out
    stores = list(cache.glob("*.safetensors"))
    assert len(stores) == 1
    assert bv.main(args + ["--index-cache", str(cache)]) == out
    assert "loaded 14 gallery rows" in capsys.readouterr().out
    other = bv.main(["--dataset", "CIRR", "--model-path", str(world["root"] / "ckpt_other.pt"), "--dtype", "fp32", "--vit-depth",
                     str(DEPTH), "--index-cache", str(cache)])
    assert len(list(cache.glob("*.safetensors"))) == 2 and other != out        # its own store, its own numbers
    with pytest.raises(ValueError):
        bv.main(["--dataset", "imagenet"])


def test_cirr_test_submission_on_disk(world):
    from sprc_amd import cirr_test_submission as cts
    base = ["--model-path", str(world["root"] / "ckpt.pt"), "--dtype", "fp32", "--vit-depth", str(DEPTH)]
    cts.main(base)
    folder = world["root"] / "submission" / "CIRR"
    rec = json.loads((folder / "recall_submission_blip2_cir_align_prompt_2.json").read_text())
    sub = json.loads((folder / "recall_subset_submission_blip2_cir_align_prompt_2.json").read_text())
    assert rec["version"] == "rc2" and rec["metric"] == "recall" and sub["metric"] == "recall_subset"
    sim, ref, tgt, grp = _oracle_metrics(world)
    want_top, want_sub = O.cirr_test_dicts(sim, ref, grp, [x["pairid"] for x in world["trip"]], world["names"])
    for k, v in want_top.items():
        assert rec[k] == v and sub[k] == want_sub[k]
    # --rerank: stage 2 over the top-50 (here: the whole 14-image gallery) with the ITM head of the same checkpoint
    cts.main(base + ["--rerank", "true", "--gpu-preprocess"])
    rec2 = json.loads((folder / "recall_submission_blip2_cir_align_prompt_2.json").read_text())
    with torch.no_grad():
        images = torch.from_numpy(np.stack([P.targetpad_transform(world["arrays"][n]) for n in world["names"]]))
        raw = O.encode_image_tokens(world["sd"], world["cfg"], images)
    from sprc_amd.processors import BlipCaptionProcessor
    from sprc_amd.tokenizer import BertWordPieceTokenizer
    t = BertWordPieceTokenizer()([BlipCaptionProcessor()(x["caption"]) for x in world["trip"]], padding="max_length", truncation=True,
                                 max_length=32, return_tensors="pt")
    order = O.rank_stable(sim)
    with torch.no_grad():
        prob = torch.stack([O.inference_rerank(world["sd"], world["cfg"], raw[ref[q]:ref[q] + 1], raw[torch.from_numpy(order[q].copy())],
                                               t.input_ids[q:q + 1], t.attention_mask[q:q + 1]) for q in range(NQ)]).numpy()
    want_rr, _ = O.cirr_test_dicts(sim, ref, grp, [x["pairid"] for x in world["trip"]], world["names"], rerank_scores=prob)
    for k, v in want_rr.items():
        assert rec2[k] == v


def test_blip_validate_fashioniq_on_disk(world):
    """`--dataset fashionIQ` on a FashionIQ-layout directory (captions / image_splits / images per dress type): the printed
    Recall@10/50 equal an independent evaluation (PIL-exact transform + CPU oracle + the oracle's metric code); the PIL
    transform in fork-server workers, the GPU transform with decode-only workers and the feature store agree."""
    from sprc_amd import blip_validate as bv
    from sprc_amd.processors import BlipCaptionProcessor, fiq_compose_caption
    from sprc_amd.tokenizer import BertWordPieceTokenizer
    root, cfg, sd = world["root"], world["cfg"], world["sd"]
    fiq = root / "fashionIQ_dataset"
    for sub in ("captions", "image_splits", "images"):
        (fiq / sub).mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(11)
    want = {}
    for d, n_img, nq in (("dress", 9, 6), ("shirt", 7, 5)):
        names, arrays = [], {}
        for i in range(n_img):
            arr = synth_image(200 + 37 * i, 260 - 11 * i, 300 + i + len(d))
            name = f"{d[0]}{i:04d}"
            Image.fromarray(arr).save(fiq / "images" / f"{name}.png")
            names.append(name)
            arrays[name] = arr
        trip = []
        for q in range(nq):
            ref = int(rng.integers(0, n_img))
            tgt = int((ref + 1 + rng.integers(0, n_img - 1)) % n_img)
            caps = [" ".join(rng.choice(WORDS[:20], size=int(rng.integers(2, 6))).tolist()) for _ in range(2)]
            trip.append({"candidate": names[ref], "target": names[tgt], "captions": caps})
        (fiq / "captions" / f"cap.{d}.val.json").write_text(json.dumps(trip))
        (fiq / "image_splits" / f"split.{d}.val.json").write_text(json.dumps(names))
        # independent evaluation
        images = torch.from_numpy(np.stack([P.targetpad_transform(arrays[n]) for n in names]))
        proc, tok = BlipCaptionProcessor(), BertWordPieceTokenizer()
        caps = [proc(fiq_compose_caption(t["captions"][0], t["captions"][1])) for t in trip]
        t = tok(caps, padding="max_length", truncation=True, max_length=32, return_tensors="pt")
        n2i = {n: i for i, n in enumerate(names)}
        ref_idx = torch.tensor([n2i[x["candidate"]] for x in trip])
        tgt_idx = np.array([n2i[x["target"]] for x in trip])
        with torch.no_grad():
            feats, raw = O.extract_target_features(sd, cfg, images)
            sim = O.inference(sd, cfg, raw[ref_idx], feats, t.input_ids, t.attention_mask).numpy()
        want[d] = O.fiq_metrics(sim, tgt_idx)
    out = bv.blip_validate_fiq(["dress", "shirt"], "blip2_cir_align_prompt", "pretrain", str(root / "ckpt.pt"), "fp32", None, False, DEPTH)
    for d in ("dress", "shirt"):
        assert (out[f"{d}_recall_at10"], out[f"{d}_recall_at50"]) == pytest.approx(want[d], abs=1e-4)
    assert out["average_recall"] == pytest.approx((np.mean([want[d][0] for d in want]) + np.mean([want[d][1] for d in want])) / 2, abs=1e-4)
    assert bv.blip_validate_fiq(["dress", "shirt"], "blip2_cir_align_prompt", "pretrain", str(root / "ckpt.pt"), "fp32", None, True, DEPTH) == out
