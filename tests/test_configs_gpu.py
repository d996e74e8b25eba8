"""BASELINE.json's single-GPU configurations at FULL size through the evaluation harness:

* C2 "CIRR-val full gallery (~2k), ViT-g, batch 128": 2297 synthetic images through `harness.extract_index_blip_features`
  (full-depth ViT-g, the engine's default 16-bit dtype, loader batch 128 -> 17 full batches + one of 121), then 4181 composed
  queries through `compute_cirr_val_metrics`;
* C3 "FashionIQ (dress+shirt+toptee) full gallery": every category at its real size (3817 x 2017, 6346 x 2038, 5373 x 1961) through
  `compute_fiq_val_metrics`.

Checked: (1) a sample of gallery images spread over the batches (first / middle / last ragged batch) re-encoded ALONE gives the same
feature bits as inside its batch of 128 -- a sample's rows do not depend on the batch it rides in (rows of the last half panel of a
batch go through the split-K remainder launch of fc2, another summation order: those agree to 16-bit noise and are counted);
(2) the metrics the harness returns equal the numpy oracle's on the SAME device scores (integer work: exact);
(3) raw embeddings are kept for reference images only (the `RawStore` policy), feature rows have unit norm, nothing is NaN.
"""
import numpy as np
import pytest
import torch
from torch.utils.data import Dataset

pytestmark = pytest.mark.gpu

from oracle import sprc_oracle as O  # noqa: E402
from sprc_amd import engine as E  # noqa: E402
from sprc_amd import harness as H  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402
from sprc_amd.model import Blip2QformerCirAlignPrompt  # noqa: E402
from sprc_amd.tokenizer import TokenBatch  # noqa: E402

DEV = "cuda:0"
TXT = {"eval": lambda c: c}


class _Tok:
    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        rows = [int(t.split()[0][1:]) for t in text]              # "q<i>" or the FashionIQ composition "Q<i> and x"
        return TokenBatch(self.ids[rows], self.mask[rows])


class _Gallery(Dataset):
    """N synthetic 224 x 224 images, drawn on demand (one generator per image: any subset reproduces the same pixels)."""
    split = "val"

    def __init__(self, n, seed):
        self.n, self.seed = n, seed
        self.names = [f"img-{i:05d}" for i in range(n)]

    def __len__(self):
        return self.n

    def image(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        return torch.randn((3, 224, 224), generator=g)

    def __getitem__(self, i):
        return self.names[i], self.image(i)


class _CirrRel(Dataset):
    def __init__(self, ref, tgt, groups):
        self.ref, self.tgt, self.groups = ref, tgt, groups

    def __len__(self):
        return len(self.ref)

    def __getitem__(self, i):
        return f"img-{self.ref[i]:05d}", f"img-{self.tgt[i]:05d}", f"q{i}", [f"img-{g:05d}" for g in self.groups[i]]


class _FiqRel(_CirrRel):
    dress_types = ["shirt"]

    def __getitem__(self, i):
        return f"img-{self.ref[i]:05d}", f"img-{self.tgt[i]:05d}", [f"q{i}", "x"]


@pytest.fixture(scope="module")
def model():
    cfg = get_config("pretrain")
    m = Blip2QformerCirAlignPrompt(cfg=cfg, max_batch=128).to(DEV).init_synthetic(seed=0)
    assert m.compute_dtype == "fp16" and cfg.vit.depth == 39
    return m


def _queries(nq, n, seed):
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, n, nq)
    tgt = (ref + 1 + rng.integers(0, n - 1, nq)) % n
    groups = np.zeros((nq, 6), dtype=np.int64)
    for q in range(nq):
        others = rng.choice(n, 8, replace=False)
        others = [int(o) for o in others if o not in (ref[q], tgt[q])][:4]
        groups[q] = rng.permutation(np.array([ref[q], tgt[q], *others]))
    ids, mask, _ = synth.make_queries(nq, n, seed=seed + 1)
    return ref, tgt, groups, ids, mask


def _encode(model, gal, refs):
    keep = {gal.names[i] for i in set(refs.tolist())}
    (feats, raw), names = H.extract_index_blip_features(gal, model, batch_size=128, num_workers=0, keep_raw=keep)
    assert names == gal.names and feats.shape == (gal.n, 32, 256) and torch.isfinite(feats).all()
    assert float((feats.norm(dim=-1) - 1).abs().max()) < 1e-5
    kept = sum(r is not None for r in raw)
    assert kept == len(keep) < gal.n                                  # raw embeddings for reference images only
    return feats, raw, names


def _batch_invariance(model, gal, feats, sample):
    """rows of sampled images encoded alone (batch of 5) vs inside their loader batch of 128"""
    exact, noisy = 0, []
    for s in range(0, len(sample), 5):
        idx = sample[s:s + 5]
        f_alone, _ = model.extract_target_features(torch.stack([gal.image(i) for i in idx]).to(DEV))
        for j, i in enumerate(idx):
            if torch.equal(f_alone[j], feats[i]):
                exact += 1
            else:
                noisy.append((i, float((f_alone[j] - feats[i]).abs().max())))
    return exact, noisy


def test_c2_cirr_val_full_size_through_the_harness(model):
    n, nq = 2297, 4181
    gal = _Gallery(n, seed=1)
    ref, tgt, groups, ids, mask = _queries(nq, n, seed=2)
    model.tokenizer = _Tok(ids, mask)
    feats, raw, names = _encode(model, gal, ref)
    # batches: 17 x 128 + 121.  In a batch of B images the rows [B * 257 // 256 * 256, B * 257) -- the tail of the LAST image -- take
    # the remainder launches: sample first / interior / last-of-batch images of the first, a middle and the ragged last batch
    sample = [0, 1, 64, 126, 127, 128, 1150, 1151, 1279, 2175, 2176, 2200, 2296]
    exact, noisy = _batch_invariance(model, gal, feats, sample)
    print(f"\n[C2] {exact} of {len(sample)} sampled images bit-identical to their stand-alone encoding; others (last image of a batch: "
          f"split-K remainder rows): {[(i, f'{d:.1e}') for i, d in noisy]}")
    last_of_batch = lambda i: (i + 1) % 128 == 0 or i == n - 1                      # noqa: E731
    assert exact >= len(sample) - 5 and all(last_of_batch(i) and d < 1e-3 for i, d in noisy)
    assert not any(last_of_batch(i) for i in set(sample) - {i for i, _ in noisy}) or exact >= 8
    rel = _CirrRel(ref, tgt, groups)
    got = H.compute_cirr_val_metrics(rel, model, (feats, raw), names, TXT)
    sim, *_ = H.generate_cirr_val_predictions(model, rel, names, (feats, raw), TXT, num_workers=0)
    assert sim.shape == (nq, n) and torch.isfinite(sim).all()
    want = O.cirr_metrics(sim.cpu().numpy(), ref, tgt, groups)
    assert got == want
    print(f"[C2] CIRR-val sizes {n} x {nq}: metrics == oracle on the device scores: {[round(x, 3) for x in got]}")


@pytest.mark.parametrize("category,n,nq", [("dress", 3817, 2017), ("shirt", 6346, 2038), ("toptee", 5373, 1961)])
def test_c3_fashioniq_every_category_full_size(model, category, n, nq):
    """BASELINE config C3 "FashionIQ (dress+shirt+toptee) full gallery": all THREE categories at their real sizes (SURVEY.md section 8:
    3 817 / 6 346 / 5 373 gallery images, 2 017 / 2 038 / 1 961 queries; validate_blip.py:149-207), full-depth ViT-g, the engine's
    default dtype, through `compute_fiq_val_metrics`."""
    seed = {"dress": 13, "shirt": 3, "toptee": 23}[category]
    gal = _Gallery(n, seed=seed)
    ref, tgt, groups, ids, mask = _queries(nq, n, seed=seed + 1)
    model.tokenizer = _Tok(ids, mask)
    feats, raw, names = _encode(model, gal, ref)
    last = (n // 128) * 128                                              # first image of the ragged last batch
    sample = [0, 700, min(3000, last - 2), last - 1, last, n - 1]
    exact, noisy = _batch_invariance(model, gal, feats, sample)
    assert exact >= 4 and all(((i + 1) % 128 == 0 or i == n - 1) and d < 1e-3 for i, d in noisy)
    rel = _FiqRel(ref, tgt, groups)
    rel.dress_types = [category]
    fiq_txt = {"eval": lambda c: c}
    got = H.compute_fiq_val_metrics(rel, model, (feats, raw), names, fiq_txt)
    sim, *_ = H.generate_fiq_val_predictions(model, rel, names, (feats, raw), fiq_txt, num_workers=0)
    assert sim.shape == (nq, n)
    want = O.fiq_metrics(sim.cpu().numpy(), tgt)
    assert tuple(got) == tuple(want)
    print(f"\n[C3] FashionIQ '{category}' sizes {n} x {nq}: R@10, R@50 == oracle on the device scores: {got}")
