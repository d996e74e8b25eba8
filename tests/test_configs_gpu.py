"""BASELINE.json's single-GPU configurations at FULL size through the evaluation harness:

* C2 "CIRR-val full gallery (~2k), ViT-g, batch 128": 2297 synthetic images through `harness.extract_index_blip_features`
  (full-depth ViT-g, the engine's default 16-bit dtype, loader batch 128 -> 17 full batches + one of 121), then 4181 composed
  queries through `compute_cirr_val_metrics`;
* C3 "FashionIQ full gallery": the largest category's 6346-image gallery and its 2038 queries through
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


def test_c3_fashioniq_largest_category_full_size(model):
    n, nq = 6346, 2038
    gal = _Gallery(n, seed=3)
    ref, tgt, groups, ids, mask = _queries(nq, n, seed=4)
    model.tokenizer = _Tok(ids, mask)
    feats, raw, names = _encode(model, gal, ref)
    exact, noisy = _batch_invariance(model, gal, feats, [0, 700, 3000, 6271, 6272, 6345])
    assert exact >= 4 and all(((i + 1) % 128 == 0 or i == n - 1) and d < 1e-3 for i, d in noisy)
    rel = _FiqRel(ref, tgt, groups)
    fiq_txt = {"eval": lambda c: c}
    got = H.compute_fiq_val_metrics(rel, model, (feats, raw), names, fiq_txt)
    sim, *_ = H.generate_fiq_val_predictions(model, rel, names, (feats, raw), fiq_txt, num_workers=0)
    assert sim.shape == (nq, n)
    want = O.fiq_metrics(sim.cpu().numpy(), tgt)
    assert tuple(got) == tuple(want)
    print(f"\n[C3] FashionIQ 'shirt' sizes {n} x {nq}: R@10, R@50 == oracle on the device scores: {got}")


def test_c2_size_recall_of_the_fp16_engine_equals_the_fp32_engine(model):
    """North star: "Recall@1/5/10 equal to reference on CIRR-val".  CIRR-val SIZES (2297 gallery images, 4181 composed queries), planted-
    structure weights and images (scores spread over ~1.0), full depth: the fp32 engine stands in for the reference (it matches the
    reference's scores to 5e-6 on every reference-generated golden) and the fp16 engine -- the dtype bench.py headlines -- must give
    the same CIRR subset recalls and Recall@1/5/10 (and Recall@50 within 0.3 points: see the last assertion) on targets placed at planned
    ranks of the fp32 ordering, K boundaries kept 5e-3 away from near-ties where such a position exists within 40 ranks.  Printed: the score-error
    distribution over all 9.6 M scores (what a 16-bit ViT does at scale: DESIGN.md section 4.3)."""
    n, nq = 2297, 4181
    cfg = get_config("pretrain")
    sd = synth.make_state_dict(cfg, seed=5, planted=True)
    g = torch.Generator().manual_seed(5)
    basis = torch.randn((8, 3, 224, 224), generator=g)
    coef = torch.randn((n, 8), generator=g)
    ids, mask, ref = synth.make_queries(nq, n, seed=6)
    ref = ref.numpy()
    sims = {}
    for dtype in ("fp32", "fp16"):
        eng = E.Engine(cfg, sd, DEV, dtype=dtype, max_batch=233)
        feats, raws = [], []
        for s in range(0, n, 128):                     # planted images, drawn batch by batch (the same for both engines)
            gb = torch.Generator().manual_seed(1000 + s)
            noise = torch.randn((min(128, n - s), 3, 224, 224), generator=gb)
            img = (torch.einsum("nk,kchw->nchw", coef[s:s + 128], basis) * 0.8 + noise * 0.4).to(DEV)
            raw = eng.vit_forward(img)
            feats.append(eng.qformer_image(raw)[0])
            raws.append(raw.to(torch.float16) if dtype == "fp16" else raw)            # (the fp16 engine rounds them to fp16 anyway)
        feats, raws = torch.cat(feats), torch.cat(raws)
        fus = []
        for s in range(0, nq, 233):
            r = raws[torch.from_numpy(ref[s:s + 233]).to(DEV)].float()
            fus.append(eng.qformer_fuse(r, ids[s:s + 233], mask[s:s + 233])[0])
        sims[dtype] = E.sim_max(torch.cat(fus), feats)
        del eng, feats, raws, fus
        torch.cuda.empty_cache()
    s32, s16 = sims["fp32"], sims["fp16"]
    d = (s16 - s32).abs()
    q = torch.quantile(d.flatten()[::7].float(), torch.tensor([0.5, 0.99, 0.999, 0.9999], device=DEV)).tolist()
    # targets at planned ranks of the fp32 ordering (reference image removed), moved to the nearest position with 5e-3 of room
    s = s32.cpu().numpy().copy()
    s[np.arange(nq), ref] = -np.inf
    order = np.argsort(-s, axis=1, kind="stable")
    plan = [0, 0, 1, 2, 3, 4, 5, 8, 9, 10, 15, 30, 48, 49, 50, 51, 75, 120]
    rng = np.random.default_rng(9)
    tgt = np.zeros(nq, dtype=np.int64)
    groups = np.zeros((nq, 6), dtype=np.int64)
    for qi in range(nq):
        sc = s[qi][order[qi]]
        gaps = sc[:-1] - sc[1:]                                                          # gap below position p
        want = plan[qi % len(plan)]
        ok = [p for p in range(max(1, want - 40), want + 40) if gaps[p - 1] > 5e-3 and gaps[p] > 5e-3] or [want]
        pos = 0 if (want == 0 and gaps[0] > 5e-3) else min(ok, key=lambda p: (abs(p - want), p))
        tgt[qi] = order[qi][pos]
        others = [int(o) for o in rng.choice(n, 8, replace=False) if o not in (ref[qi], tgt[qi])][:4]
        groups[qi] = rng.permutation(np.array([ref[qi], tgt[qi], *others]))
    m32 = H.cirr_metrics_from_sim(s32, ref, tgt, groups)
    m16 = H.cirr_metrics_from_sim(s16, ref, tgt, groups)
    f32, f16 = H.fiq_metrics_from_sim(s32, tgt), H.fiq_metrics_from_sim(s16, tgt)
    top1 = float((s16.argmax(1) == s32.argmax(1)).float().mean())
    print(f"\n[C2 sizes, planted, fp16 vs fp32 engine] max|dsim|={float(d.max()):.2e} rms={float(d.pow(2).mean().sqrt()):.2e} quantiles 50 / 99 / 99.9 / 99.99 %: "
          f"{q[0]:.1e} / {q[1]:.1e} / {q[2]:.1e} / {q[3]:.1e} over {d.numel()} scores; top-1 image equal for {100 * top1:.2f} % of the queries; "
          f"CIRR metrics fp32 {[round(x, 2) for x in m32]} fp16 {[round(x, 2) for x in m16]}; FashionIQ {f32} / {f16}")
    assert float(d.pow(2).mean().sqrt()) < 4e-4 and q[1] < 1.2e-3
    # subset recalls and Recall@1/5/10: equal.  Recall@50: around position 50 of 2297 the reference's own score gaps (median 1e-4) are below
    # ANY 16-bit engine's error, a 5e-3 margin does not exist there and the planned target keeps its place without one: a handful
    # of the 4181 queries cross K = 50 (measured: 6 = 0.14 points), which is what "equal Recall" can mean at this gallery size
    np.testing.assert_allclose(m16[:6], m32[:6], rtol=0, atol=1e-9)
    assert abs(m16[6] - m32[6]) < 0.3 and f16[0] == f32[0] and abs(f16[1] - f32[1]) < 0.3
