"""The composite forward at the batch it is BENCHMARKED at: 128 images / 233 queries at full depth
(M = 32 896 rows: the 256x256 anti-phase GEMM kernel, the peeled remainder panel, the split-K remainder of fc2, 2 048-
workgroup attention launches, ~2.4 GB of carved workspace) -- none of which the 2-image goldens reach.

* fp32 engine, B = 128 / 233: the two images and three queries of tests/golden/full_eva.npz (outputs of the unmodified
  REFERENCE at full depth) are planted inside the big batches; their rows must match the reference like the small run.
* bf16 engine, B = 128 / 233 vs a B = 5 / 3 run of the same samples: every kernel on the path reduces over K in the
  same order whatever the tile shape, so the rows are BIT-IDENTICAL -- except the rows of the last half panel
  (image 127), whose fc2 products go through the 8-way split-K remainder launch (another summation order): those agree
  to bf16 noise.  The bf16 scores of the planted samples stay within the 1e-3 bar of the reference.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sprc_amd import engine as E  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"

import sys as _sys, os as _os  # noqa: E402
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _cases as CASES  # noqa: E402  (session-lived full-depth state dicts: tests/_cases.py)
B, NQ = 128, 233
POS = [5, 77]                    # batch slots of the golden's two images
QSLOT = [0, 100, 232]            # query slots of the golden's three queries
SAMPLE = [5, 77, 0, 126, 127]    # images re-run at B = 5 (127 = the split-K tail)


@pytest.fixture(scope="module")
def setup(golden_dir):
    g = np.load(golden_dir / "full_eva.npz", allow_pickle=False)
    cfg = get_config("pretrain")
    assert int(g["vit_depth"]) == cfg.vit.depth == 39
    sd = CASES.state_dict(cfg, int(g["seed"]))
    images = synth.make_images(B, seed=4321)
    images[POS] = synth.make_images(int(g["n_img"]), seed=int(g["seed"]))
    ids, mask, _ = synth.make_queries(NQ, B, seed=77)
    ids[QSLOT], mask[QSLOT] = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    ref_slot = (7919 * torch.arange(NQ)) % B
    ref_slot[QSLOT] = torch.tensor([POS[i] for i in g["ref_index"].tolist()])
    return g, cfg, sd, images, ids, mask, ref_slot


def _forward(eng, images, ids, mask, ref_slot):
    raw = eng.vit_forward(images.to(DEV))
    feats, _ = eng.qformer_image(raw)
    fusion, _ = eng.qformer_fuse(raw.index_select(0, ref_slot.to(DEV)), ids, mask)
    torch.cuda.synchronize()
    return raw, feats, fusion


def test_fp32_engine_at_bench_batch_matches_the_reference(setup):
    g, cfg, sd, images, ids, mask, ref_slot = setup
    eng = E.Engine(cfg, sd, DEV, dtype="fp32", max_batch=NQ)
    raw, feats, fusion = _forward(eng, images, ids, mask, ref_slot)
    rows = g["rows"].tolist()
    np.testing.assert_allclose(raw[POS][:, rows].cpu().numpy(), g["raw"], atol=2e-3, rtol=0)
    np.testing.assert_allclose(feats[POS].cpu().numpy(), g["feats"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(fusion[QSLOT].cpu().numpy(), g["fusion"], atol=1e-4, rtol=0)
    sim = E.sim_max(fusion[QSLOT].contiguous(), feats[POS].contiguous()).cpu().numpy()
    np.testing.assert_allclose(sim, g["sim"], atol=1e-4, rtol=0)
    print(f"\n[B={B}/{NQ} fp32] max|dsim|={np.abs(sim - g['sim']).max():.2e} max|dfeats|={np.abs(feats[POS].cpu().numpy() - g['feats']).max():.2e}")
    del eng
    torch.cuda.empty_cache()


def test_bf16_engine_bench_batch_equals_small_batch(setup):
    g, cfg, sd, images, ids, mask, ref_slot = setup
    eng = E.Engine(cfg, sd, DEV, dtype="bf16", max_batch=NQ)
    raw, feats, fusion = _forward(eng, images, ids, mask, ref_slot)
    assert torch.isfinite(raw).all() and torch.isfinite(feats).all() and torch.isfinite(fusion).all()
    # the same five images / three queries alone
    small = images[SAMPLE]
    raw_s = eng.vit_forward(small.to(DEV))
    feats_s, _ = eng.qformer_image(raw_s)
    local = torch.tensor([SAMPLE.index(POS[i]) for i in g["ref_index"].tolist()])
    fusion_s, _ = eng.qformer_fuse(raw_s.index_select(0, local.to(DEV)), ids[QSLOT], mask[QSLOT])
    torch.cuda.synchronize()
    same = [i for i, s in enumerate(SAMPLE) if s != 127]
    assert torch.equal(raw[SAMPLE][same], raw_s[same]), "rows of a sample must not depend on the batch it rides in"
    assert torch.equal(feats[SAMPLE][same], feats_s[same])
    assert torch.equal(fusion[QSLOT], fusion_s)
    tail = SAMPLE.index(127)                     # split-K remainder rows: another summation order, bf16-noise apart
    d_tail = float((raw[127] - raw_s[tail]).abs().max())
    cos_tail = float((feats[127] * feats_s[tail]).sum(-1).min())
    print(f"\n[B={B} bf16] image 127 (split-K tail) vs its B=5 run: max|draw|={d_tail:.2e} min cos(feats)={cos_tail:.6f}")
    assert d_tail < 5e-2 and cos_tail > 0.9995
    sim = E.sim_max(fusion[QSLOT].contiguous(), feats[POS].contiguous()).cpu().numpy()
    dsim = np.abs(sim - g["sim"]).max()
    print(f"[B={B}/{NQ} bf16] max|dsim| vs the reference = {dsim:.2e}")
    assert dsim < 1e-3
