"""CPU-side tests: host logic (caption processor, tokenizer, model protocol surface), and that the C-ABI
library loads and exports every symbol include/sprc.h declares (no compute calls without a GPU)."""
import json
import re
import tempfile
from pathlib import Path

import pytest
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from sprc_amd import _lib
    header = (ROOT / "include" / "sprc.h").read_text()
    declared = set(re.findall(r"\b(sprc_[a-z0-9_]+)\s*\(", header))
    declared -= {"sprc_stream"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), f"binding/header mismatch: {declared ^ set(_lib.SIGNATURES)}"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sprc_version() == _lib.ABI_VERSION == int(re.search(r"#define SPRC_ABI_VERSION (\d+)", header).group(1))
    assert isinstance(lib.sprc_last_error(), bytes)


def test_argument_validation_reports_errors_without_a_gpu():
    import ctypes as C
    from sprc_amd import _lib as L
    lib = L.load()
    g = L.GemmArgs()
    g.M, g.N, g.K, g.dtype, g.out_dtype = 4, 4, 48, L.SPRC_BF16, L.SPRC_F32
    assert lib.sprc_gemm(C.byref(g), None) == -1
    assert b"multiple" in lib.sprc_last_error()
    with pytest.raises(L.SprcError):
        L.check(lib.sprc_topk(None, 0, None, 0, 1, 1, 100, None, None, None), "sprc_topk")


def test_caption_processor_matches_reference(golden_dir):
    from sprc_amd.processors import BlipCaptionProcessor, fiq_compose_caption
    c = json.loads((golden_dir / "captions.json").read_text())
    proc = BlipCaptionProcessor()
    for raw, want in c["pre_caption"]:
        assert proc(raw) == want
    for c1, c2, composed, processed in c["fiq"]:
        assert fiq_compose_caption(c1, c2) == composed and proc(composed) == processed


def _vocab(tmp: Path) -> Path:
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += list("abcdefghijklmnopqrstuvwxyz0123456789") + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    toks += ["the", "dog", "is", "now", "stand", "##ing", "and", "by", "him", "##self", "make", "it", "more", "color",
             "##ful", "remove", "second", "person", "two", "dogs", "instead", "of", "one", ",", ".", "'", "-", "!", "?", "caf", "##e"]
    p = tmp / "vocab.txt"
    p.write_text("\n".join(toks) + "\n", encoding="utf-8")
    return p


def test_wordpiece_tokenizer_matches_transformers():
    """R9 pinned against the installed third-party implementation on a synthetic vocabulary (the real
    bert-base-uncased vocab is a network fetch; see sprc_amd/tokenizer.py)."""
    from transformers import BertTokenizer
    from sprc_amd.tokenizer import BertWordPieceTokenizer
    with tempfile.TemporaryDirectory() as d:
        vp = _vocab(Path(d))
        hf = BertTokenizer.from_pretrained(d, do_lower_case=True)       # transformers 5.x: vocab.txt in a directory
        hf.add_special_tokens({"bos_token": "[DEC]"})
        mine = BertWordPieceTokenizer(str(vp))
        assert len(mine) == len(hf)
        texts = ["the dog is now standing and by himself", "Make it more COLORFUL, remove the second person!",
                 "two dogs instead of one", "", "zzz qqq unknownword é café", "a" * 120,
                 " ".join(["dog"] * 60), "it's a two-dog thing?", "tab\tand\nnewline", "[SEP] the [MASK] dog"]
        a = hf(texts, padding="max_length", truncation=True, max_length=32, return_tensors="pt")
        b = mine(texts, padding="max_length", truncation=True, max_length=32, return_tensors="pt")
        assert torch.equal(a.input_ids, b.input_ids)
        assert torch.equal(a.attention_mask, b.attention_mask)


def test_model_surface_and_checkpoint_keys():
    from sprc_amd import synth
    from sprc_amd.config import get_config
    from sprc_amd.model import Blip2QformerCirAlignPrompt, load_model_and_preprocess
    cfg = get_config("pretrain", vit_depth=1, q_layers=2)
    m = Blip2QformerCirAlignPrompt(cfg=cfg)
    assert m.__class__.__name__ == "Blip2QformerCirAlignPrompt"            # the checkpoint key (utils.py:218-222)
    keys = set(m.state_dict())
    want = {n for n, _, _ in synth.param_specs(cfg)} | {"temp"}
    assert keys == want
    for k in ("visual_encoder.blocks.0.attn.qkv.weight", "visual_encoder.blocks.0.attn.q_bias", "ln_vision.weight",
              "Qformer.bert.encoder.layer.0.crossattention.self.key.weight", "Qformer.bert.encoder.layer.1.output_query.dense.weight",
              "query_tokens", "prompt_tokens", "vision_proj.weight", "text_proj.bias"):
        assert k in keys
    assert "Qformer.bert.encoder.layer.1.crossattention.self.key.weight" not in keys      # cross-attention on even layers only
    # the reference's save format: {epoch, ClassName: state_dict}; loaded with strict=False; extra keys tolerated
    sd = synth.make_state_dict(cfg, seed=5)
    sd["Qformer.cls.predictions.bias"] = torch.zeros(3)
    sd["visual_encoder.blocks.0.mlp.fc1.weight"] = sd["visual_encoder.blocks.0.mlp.fc1.weight"].half()   # fp16 ViT weights
    msg = m.load_state_dict({"epoch": 1, m.__class__.__name__: sd}[m.__class__.__name__], strict=False)
    assert msg.missing_keys == [] and msg.unexpected_keys == ["Qformer.cls.predictions.bias"]
    assert m.eval() is m and m.device.type == "cpu"
    from sprc_amd import _lib
    with pytest.raises(_lib.SprcError, match="no CPU fallback"):
        m.extract_target_features(torch.zeros(1, 3, 224, 224))
    with pytest.raises(_lib.SprcError, match="no CPU fallback"):           # forward (training losses) also runs on the HIP engine only
        m.tokenizer = lambda text, **kw: __import__("sprc_amd.tokenizer", fromlist=["TokenBatch"]).TokenBatch(
            torch.zeros((1, 32), dtype=torch.int64), torch.ones((1, 32), dtype=torch.int64))
        m({"image": torch.zeros(1, 3, 224, 224), "target": torch.zeros(1, 3, 224, 224), "text_input": ["x"]})
    with pytest.raises(KeyError):
        load_model_and_preprocess("blip2_cir_rerank_learn", "pretrain")


def test_engine_snapshots_are_dropped_when_the_weights_change():
    """the training step's frozen-trunk engine (`_tengine`) is a snapshot of the parameters exactly like the inference
    engine -- load_state_dict / .to() / init_synthetic must drop BOTH, or a later training step runs a stale trunk."""
    from sprc_amd.config import get_config
    from sprc_amd.model import Blip2QformerCirAlignPrompt, Blip2QformerCirRerank
    cfg = get_config("pretrain", vit_depth=1, q_layers=1)
    m = Blip2QformerCirAlignPrompt(cfg=cfg)
    for op in (lambda: m.load_state_dict(m.state_dict()), lambda: m.to(torch.float32), lambda: m.init_synthetic(1)):
        m._engine, m._tengine = object(), object()
        op()
        assert m._engine is None and m._tengine is None
    # the rerank class must not inherit align_prompt's differentiable forward (its reference objective is a different one)
    r = Blip2QformerCirRerank(cfg=cfg)
    with pytest.raises(NotImplementedError, match="stage-2 training objective"):
        r({"image": torch.zeros(1, 3, 224, 224), "target": torch.zeros(1, 3, 224, 224), "text_input": ["x"]})


def test_synthetic_inputs_follow_the_measurement_contract():
    from sprc_amd import synth
    ids, mask, ref = synth.make_queries(50, 97, seed=1)
    assert ids.shape == (50, 32) and (ids[:, 0] == 101).all()
    lens = mask.sum(1)
    assert lens.min() >= 4 and lens.max() <= 32
    assert (ids[torch.arange(50), lens - 1] == 102).all() and (ids * (1 - mask) == 0).all()
    assert torch.equal(ref, (7919 * torch.arange(50)) % 97)


def test_shard_bounds_and_owner():
    from sprc_amd.dist import owner_of, shard_bounds
    for n, w in [(2297, 8), (10, 3), (7, 8), (16, 4)]:
        cover = []
        for r in range(w):
            lo, hi = shard_bounds(n, w, r)
            cover += list(range(lo, hi))
            if hi > lo:
                assert (owner_of(torch.arange(lo, hi), n, w) == r).all()
        assert cover == list(range(n))


def test_feature_store_round_trip(tmp_path):
    """sprc_amd/index.py: a gallery saved and loaded back is the same bits, names keep their row order, and malformed
    inputs are refused (duplicate names would silently break the name -> row join of the relative datasets)."""
    import torch
    from sprc_amd.index import load_index, save_index
    g = torch.Generator().manual_seed(0)
    feats = torch.nn.functional.normalize(torch.randn((37, 32, 256), generator=g), dim=-1)
    raw = torch.randn((37, 257, 64), generator=g)
    names = [f"dev-{i:04d}-img{(i * 7) % 37}" for i in range(37)]
    p = tmp_path / "idx" / "cirr-val.safetensors"
    save_index(p, feats, names, raw=raw, backbone="pretrain", compute_dtype="bf16")
    (f2, r2), n2, meta = load_index(p)
    assert torch.equal(f2, feats) and torch.equal(r2, raw) and n2 == names
    assert meta["backbone"] == "pretrain" and meta["compute_dtype"] == "bf16" and meta["format"] == "sprc-index-1"
    (f3, r3), n3, _ = load_index(p, with_raw=False)
    assert r3 is None and torch.equal(f3, feats) and n3 == names
    save_index(tmp_path / "noraw.safetensors", feats, names)
    assert load_index(tmp_path / "noraw.safetensors")[0][1] is None
    # raw kept in the engine's 16-bit operand format: comes back as fp32 holding exactly the rounded values (idempotent rounding:
    # what the engine makes of it is what it makes of the fp32 original)
    for dt in (torch.float16, torch.bfloat16):
        save_index(tmp_path / "raw16.safetensors", feats, names, raw=raw, raw_dtype=dt)
        (f4, r4), _, _ = load_index(tmp_path / "raw16.safetensors")
        assert r4.dtype == torch.float32 and torch.equal(r4, raw.to(dt).float()) and torch.equal(r4.to(dt), raw.to(dt)) and torch.equal(f4, feats)
        assert (tmp_path / "raw16.safetensors").stat().st_size < 0.75 * p.stat().st_size
    with pytest.raises(ValueError, match="raw_dtype"):
        save_index(tmp_path / "bad.safetensors", feats, names, raw=raw, raw_dtype=torch.float64)
    with pytest.raises(ValueError, match="unique"):
        save_index(tmp_path / "dup.safetensors", feats, ["a"] * 37)
    with pytest.raises(ValueError, match="one entry per"):
        save_index(tmp_path / "short.safetensors", feats, names[:-1])
    with pytest.raises(ValueError, match=r"\[N,32,E\]"):
        save_index(tmp_path / "shape.safetensors", feats[:, :8], names)


def test_query_loaders_stay_in_the_main_process_and_gallery_workers_only_decode():
    """Loader policy of the evaluation harness (tools/c2_e2e.py measured 24-36 s per forked loader): relative splits load
    in-process whatever `num_workers` says; with an on-device transform the workers get a decode-only copy of the dataset."""
    import pickle
    from torch.utils.data import Subset
    from sprc_amd import harness as H
    from sprc_amd.data_utils import DecodeRGB, HostTargetPad, targetpad_transform

    rel = [("a", "b", "cap", ["a", "b"])] * 5
    loader = H._query_loader(rel, 2, 4, collate_fn=H.collate_fn)
    assert loader.num_workers == 0 and not loader.pin_memory and len(list(loader)) == 3

    class OnDevice:
        on_device = True

    class DS:
        def __init__(self):
            self.preprocess, self.names = OnDevice(), ["x", "y", "z"]

    ds = DS()
    tf, clone = H._decode_only(Subset(ds, [0, 2]))
    assert tf is ds.preprocess and isinstance(clone, Subset) and list(clone.indices) == [0, 2]
    assert isinstance(clone.dataset.preprocess, DecodeRGB) and isinstance(ds.preprocess, OnDevice)      # the original is untouched
    assert not H._pin(ds) and H._pin(clone)
    # what fork-server workers have to unpickle
    assert isinstance(pickle.loads(pickle.dumps(targetpad_transform(1.25, 224))), HostTargetPad)
    assert isinstance(pickle.loads(pickle.dumps(DecodeRGB())), DecodeRGB)
    # decode workers hand RGB / L images over as uint8 [H, W, 3]; modes the GPU transform does not reproduce (palette, alpha,
    # bilevel: PIL resamples those in their own mode before converting) come back already transformed by the PIL path
    from PIL import Image
    from sprc_amd.data_utils import is_transformed
    dec = DecodeRGB(1.25, 224)
    rgb = Image.fromarray((np.arange(60 * 40 * 3) % 251).astype(np.uint8).reshape(40, 60, 3))
    assert not is_transformed(dec(rgb)) and tuple(dec(rgb.convert("L")).shape) == (40, 60, 3)
    for im in (rgb.convert("P"), rgb.convert("RGBA"), rgb.convert("1")):
        out = dec(im)
        assert is_transformed(out) and torch.equal(out, HostTargetPad(1.25, 224)(im))
    # the thread-pool decode loader: dataset order, ragged last batch, unreadable items dropped
    class Items:
        def __len__(self):
            return 11

        def __getitem__(self, i):
            return None if i == 4 else (f"n{i}", torch.full((2 + i, 3, 3), i, dtype=torch.uint8))

    got = list(H._ThreadLoader(Items(), batch_size=4, threads=3))
    assert [b[0] for b in got] == [["n0", "n1", "n2", "n3"], ["n5", "n6", "n7"], ["n8", "n9", "n10"]]
    assert all(int(t[0, 0, 0]) == int(n[1:]) for b in got for n, t in zip(*b)) and len(H._ThreadLoader(Items(), 4, 3)) == 3
    # a short FIRST batch (the pipeline's fill): the first batch's remainder follows, later boundaries stay where they were
    ramp = H._ThreadLoader(Items(), batch_size=4, threads=2, first=1)
    assert [b[0] for b in ramp] == [["n0"], ["n1", "n2", "n3"], ["n5", "n6", "n7"], ["n8", "n9", "n10"]] and len(ramp) == 4
    assert [b[0] for b in H._ThreadLoader(Items(), batch_size=16, threads=2, first=32)] == [[f"n{i}" for i in range(11) if i != 4]]
    names, imgs = H._collate_ragged([("n0", torch.zeros(3, 4, 3, dtype=torch.uint8)), None, ("n1", torch.zeros(5, 2, 3, dtype=torch.uint8))])
    assert names == ["n0", "n1"] and [tuple(i.shape) for i in imgs] == [(3, 4, 3), (5, 2, 3)]


def test_model_fingerprint_sees_every_bit():
    """blip_validate.model_fingerprint (the key that ties a feature store to the checkpoint that produced it): integer
    checksums of the tensors' bits -- equal for equal state dicts, different after a one-ulp change, after swapping two elements
    (the position-weighted sum), after renaming a tensor or changing its dtype."""
    import torch
    from sprc_amd.blip_validate import model_fingerprint

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(1)
            self.a = torch.nn.Parameter(torch.randn((5, 7), generator=g))
            self.register_buffer("b", torch.randn((3,), generator=g).to(torch.bfloat16))      # 6 bytes: padded to whole words
            self.register_buffer("ids", torch.arange(4, dtype=torch.int64))

    m = M()
    f0 = model_fingerprint(m)
    assert f0 == model_fingerprint(M()) and len(f0) == 64
    with torch.no_grad():
        m.a[2, 3] = torch.nextafter(m.a[2, 3].detach(), torch.tensor(10.0))
    f1 = model_fingerprint(m)
    m2 = M()
    with torch.no_grad():
        x, y = m2.a[0, 0].clone(), m2.a[4, 6].clone()
        m2.a[0, 0], m2.a[4, 6] = y, x
    m3 = M()
    m3.b = m3.b.to(torch.float16)
    assert len({f0, f1, model_fingerprint(m2), model_fingerprint(m3)}) == 4
