"""GPU tests of the drop-in surface: the model class (`extract_target_features` / `inference`) and the
evaluation harness (names, arguments and return values of the reference's functions), against the oracle and
against the numbers the REFERENCE's own metric code produced (tests/golden/metrics.json)."""
import json

import numpy as np
import pytest
import torch
from torch.utils.data import Dataset

pytestmark = pytest.mark.gpu

from oracle import sprc_oracle as O  # noqa: E402
from sprc_amd import harness as H  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402
from sprc_amd.model import Blip2QformerCirAlignPrompt  # noqa: E402
from sprc_amd.tokenizer import TokenBatch  # noqa: E402

DEV = "cuda:0"


class FakeTokenizer:
    """Test double for the WordPiece tokenizer (the real vocabulary is a network fetch): caption "q<i>" -> row i."""

    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        rows = [int(t[1:]) for t in text]
        return TokenBatch(self.ids[rows], self.mask[rows])


@pytest.mark.parametrize("case", ["plain", "ties", "single_batch"])
def test_metric_functions_match_reference_numbers(golden_dir, case):
    c = json.loads((golden_dir / "metrics.json").read_text())[case]
    sim = torch.tensor(c["sim"], dtype=torch.float32, device=DEV)
    names = [f"img-{i:05d}" for i in range(c["N"])]
    cirr = H.cirr_metrics_from_sim(sim, c["ref"], c["tgt"], c["groups"])
    fiq = H.fiq_metrics_from_sim(sim, c["tgt"])
    top, sub = H.cirr_test_dicts_from_sim(sim, c["ref"], c["groups"], [1000 + i for i in range(c["nq"])], names)
    # always: identical to the oracle's stable-order contract (integer-exact)
    s = np.asarray(c["sim"], dtype=np.float32)
    assert cirr == O.cirr_metrics(s, np.asarray(c["ref"]), np.asarray(c["tgt"]), np.asarray(c["groups"]))
    assert fiq == O.fiq_metrics(s, np.asarray(c["tgt"]))
    o_top, o_sub = O.cirr_test_dicts(s, np.asarray(c["ref"]), np.asarray(c["groups"]), [1000 + i for i in range(c["nq"])], names)
    assert top == o_top and sub == o_sub
    if not c["ties"]:      # without ties the reference's unstable argsort has a unique answer: identical numbers
        np.testing.assert_allclose(cirr, c["cirr"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(fiq, c["fiq"], rtol=0, atol=1e-4)
        assert top == c["test_top50"] and sub == c["test_subset3"]


class _Gallery(Dataset):
    split = "val"

    def __init__(self, images):
        self.images = images

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return (f"img-{i:05d}", self.images[i]) if i != 3 else None      # one unreadable image: dropped by collate_fn


class _Relative(Dataset):
    def __init__(self, ref, tgt, groups):
        self.ref, self.tgt, self.groups = ref, tgt, groups

    def __len__(self):
        return len(self.ref)

    def __getitem__(self, i):
        return f"img-{self.ref[i]:05d}", f"img-{self.tgt[i]:05d}", f"q{i}", [f"img-{g:05d}" for g in self.groups[i]]


def test_model_protocol_and_cirr_loop_end_to_end():
    cfg = get_config("pretrain", vit_depth=2)
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32")
    sd = synth.make_state_dict(cfg, seed=11)
    msg = model.load_state_dict(sd, strict=False)
    assert not msg.missing_keys
    model = model.to(DEV)
    assert model.device.type == "cuda" and model.eval() is model
    n_img, nq = 13, 9
    images = synth.make_images(n_img, seed=12)
    (feats, raw), names = H.extract_index_blip_features(_Gallery(images), model, batch_size=5, num_workers=0)
    keep = [i for i in range(n_img) if i != 3]
    assert names == [f"img-{i:05d}" for i in keep] and feats.shape == (12, 32, 256) and raw.shape == (12, 257, 1408)
    with torch.no_grad():
        feats_o, raw_o = O.extract_target_features(sd, cfg, images[keep])
    np.testing.assert_allclose(feats.cpu().numpy(), feats_o.numpy(), atol=1e-4, rtol=0)

    ids, mask, _ = synth.make_queries(nq, 12, seed=13)
    model.tokenizer = FakeTokenizer(ids, mask)
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 12, nq)
    tgt = (ref + 1 + rng.integers(0, 11, nq)) % 12
    groups = np.stack([rng.permutation(np.array([ref[q], tgt[q], *[i for i in rng.permutation(12) if i not in (ref[q], tgt[q])][:4]]))
                       for q in range(nq)])
    gidx = [keep.index(k) if k in keep else 0 for k in range(13)]      # names -> positions in the kept gallery
    rel = _Relative([keep[r] for r in ref], [keep[t] for t in tgt], [[keep[g] for g in row] for row in groups])
    txt = {"eval": lambda c: c}
    got = H.compute_cirr_val_metrics(rel, model, (feats, raw), names, txt)
    with torch.no_grad():
        sim_o = O.inference(sd, cfg, raw_o[torch.from_numpy(ref)], feats_o, ids, mask).numpy()
    sim_d, *_ = H.generate_cirr_val_predictions(model, rel, names, (feats, raw), txt, num_workers=0)
    np.testing.assert_allclose(sim_d.cpu().numpy(), sim_o, atol=1e-4, rtol=0)
    want = O.cirr_metrics(sim_d.cpu().numpy(), ref, tgt, groups)       # integer-exact on the device scores
    assert got == want
    # single caption / single reference keeps a 2-D result
    one = model.inference(raw[:1], feats, ["q0"])
    assert one.shape == (1, 12)
    with pytest.raises(ValueError):
        model.inference(raw[:2], feats, ["q0"])


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_reference_kv_reuse_gives_the_same_scores(dtype):
    """Optional fast path (`compute_cirr_val_metrics(reuse_reference_kv=True)`, `blip_validate --reuse-reference-kv`; NOT used by bench.py):
    every distinct reference image is projected to the Q-Former's cross-attention K|V once (sprc_qformer_encode_kv) and the fusion pass
    of each query reads its row by index (sprc_qformer_fuse_kv) instead of projecting the image's 257 tokens again per query
    (Qformer.py:191-193 inside align_prompt.py:332-339).  Same scores -- the projection is the same product on the same rows -- same
    metrics; a query batch whose references repeat and arrive out of order exercises the index path of the attention kernel."""
    cfg = get_config("pretrain", vit_depth=2)
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype=dtype, max_batch=16)
    assert not model.load_state_dict(synth.make_state_dict(cfg, seed=21), strict=False).missing_keys
    model = model.to(DEV).eval()
    n_img, nq = 20, 37                                       # 37 queries over 7 distinct reference images, shuffled
    images = synth.make_images(n_img, seed=22)

    class Gal(Dataset):
        def __len__(self):
            return n_img

        def __getitem__(self, i):
            return f"img-{i:05d}", images[i]

    (feats, raw), names = H.extract_index_blip_features(Gal(), model, batch_size=8, num_workers=0)
    ids, mask, _ = synth.make_queries(nq, n_img, seed=23)
    model.tokenizer = FakeTokenizer(ids, mask)
    rng = np.random.default_rng(6)
    ref = rng.choice([1, 4, 5, 9, 12, 17, 19], nq)
    tgt = (ref + 1 + rng.integers(0, n_img - 1, nq)) % n_img
    groups = np.stack([rng.permutation(np.array([ref[q], tgt[q], *[i for i in rng.permutation(n_img) if i not in (ref[q], tgt[q])][:4]]))
                       for q in range(nq)])
    rel = _Relative(ref, tgt, groups)
    txt = {"eval": lambda c: c}
    sim_a, *_ = H.generate_cirr_val_predictions(model, rel, names, (feats, raw), txt, num_workers=0)
    rkv = H.build_reference_kv(model, names, (feats, raw), [f"img-{r:05d}" for r in ref])
    assert rkv.kv.shape[0] == 7
    sim_b, *_ = H.generate_cirr_val_predictions(model, rel, names, (feats, raw), txt, num_workers=0, reference_kv=rkv)
    d = float((sim_a - sim_b).abs().max())
    print(f"\n[reference K|V reuse, {dtype}] max |score difference| = {d:.1e} over {sim_a.numel()} scores; bit-identical: {torch.equal(sim_a, sim_b)}")
    assert d < 2e-6
    assert H.compute_cirr_val_metrics(rel, model, (feats, raw), names, txt, reuse_reference_kv=True) == \
        H.compute_cirr_val_metrics(rel, model, (feats, raw), names, txt)
    with pytest.raises(ValueError):                          # the exact-fp32 engine recomputes its projections: no kv path
        m32 = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=16)
        m32.load_state_dict(synth.make_state_dict(cfg, seed=21), strict=False)
        m32 = m32.to(DEV).eval()
        m32.engine().qformer_fuse_kv(rkv.kv.float(), torch.zeros(2, dtype=torch.int32), ids[:2], mask[:2])
