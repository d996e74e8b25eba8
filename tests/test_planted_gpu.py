"""Ordering parity on a case where order MEANS something: planted-structure weights
(sprc_amd/synth.py: plant_structure), a depth-4 ViT-g + the full Q-Former, 160 gallery images, 72 composed queries,
golden scores / targets / metrics produced by running the unmodified REFERENCE (oracle/gen_golden.py: planted_goldens).

fp32 engine: top-51 indices equal the reference's stable order UNCONDITIONALLY at every position whose reference score
differs from both neighbours by more than 1e-5 (the excluded positions are counted, printed, and must stay under 1 %);
CIRR / FashionIQ metrics and the submission dicts equal the numbers the reference's own metric code produced.
bf16 engine: the same metrics must come out equal (targets sit at planned ranks with >= 5e-3 margins to their neighbours);
top-10 agreement with the reference order is reported.
"""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sprc_oracle as O  # noqa: E402
from sprc_amd import engine as E  # noqa: E402
from sprc_amd import harness as H  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"

import sys as _sys, os as _os  # noqa: E402
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _cases as CASES  # noqa: E402  (session-lived full-depth state dicts: tests/_cases.py)
K = 51


@pytest.fixture(scope="module")
def case(golden_dir):
    return CASES.planted_case(golden_dir, "planted_eva")


def _run(case, dtype):
    g, cfg, sd, images = case
    eng = E.Engine(cfg, sd, DEV, dtype=dtype, max_batch=64)
    raw = eng.vit_forward(images.to(DEV))
    feats, _ = eng.qformer_image(raw)
    ref = torch.from_numpy(g["ref_index"]).to(DEV)
    fusion, _ = eng.qformer_fuse(raw[ref], torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    sim = E.sim_max(fusion, feats)
    torch.cuda.synchronize()
    return feats, fusion, sim


def _metrics(g, sim):
    cirr = H.cirr_metrics_from_sim(sim, g["ref_index"], g["tgt_index"], g["groups"])
    fiq = H.fiq_metrics_from_sim(sim, g["tgt_index"])
    names = [f"img-{i:05d}" for i in range(int(g["n_img"]))]
    top, sub = H.cirr_test_dicts_from_sim(sim, g["ref_index"], g["groups"], [1000 + i for i in range(int(g["n_q"]))], names)
    return cirr, fiq, top, sub


def test_fp32_engine_reproduces_the_reference_order(case):
    g = case[0]
    feats, fusion, sim = _run(case, "fp32")
    s = sim.cpu().numpy()
    np.testing.assert_allclose(feats[:4].cpu().numpy(), g["feats_head"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(fusion.cpu().numpy(), g["fusion"], atol=1e-4, rtol=0)
    dsim = np.abs(s - g["sim"]).max()
    assert dsim < 1e-4                                                     # bar: 1e-3
    _, idx = E.topk(sim, K)
    idx = idx.cpu().numpy().astype(np.int64)
    ref_order = O.rank_stable(g["sim"])[:, :K + 1]
    ref_sorted = np.take_along_axis(g["sim"], ref_order, axis=1)
    gap = np.abs(np.diff(ref_sorted, axis=1))                              # gap[:, p] between reference positions p and p+1
    solid = np.ones((s.shape[0], K), dtype=bool)
    solid[:, 1:] &= gap[:, :K - 1] > 1e-5
    solid &= gap[:, :K] > 1e-5
    excluded = int((~solid).sum())
    print(f"\n[planted fp32] max|dsim|={dsim:.2e}; {excluded} of {solid.size} top-{K} positions sit within 1e-5 of a neighbour "
          f"in the REFERENCE's scores and are excluded ({100.0 * excluded / solid.size:.3f} %)")
    assert excluded <= 0.01 * solid.size
    np.testing.assert_array_equal(idx[solid], ref_order[:, :K][solid])     # unconditional: no `if gap` escape
    cirr, fiq, top, sub = _metrics(g, sim)
    np.testing.assert_allclose(cirr, g["cirr"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(fiq, g["fiq"], rtol=0, atol=1e-4)
    assert 0.0 < cirr[3] < cirr[5] < cirr[6] < 100.0                       # Recall@1 < @10 < @50: a non-trivial case
    dicts = json.loads(str(g["test_dicts"]))
    assert top == dicts["top"] and sub == dicts["sub"]


def test_bf16_engine_keeps_the_metrics(case):
    g = case[0]
    _, _, sim = _run(case, "bf16")
    s = sim.cpu().numpy()
    dsim = np.abs(s - g["sim"]).max()
    cirr, fiq, top, sub = _metrics(g, sim)
    ref_order = O.rank_stable(g["sim"])
    _, idx = E.topk(sim, K)
    idx = idx.cpu().numpy().astype(np.int64)
    agree10 = float((idx[:, :10] == ref_order[:, :10]).mean())
    set10 = float(np.mean([len(set(a[:10]) & set(b[:10])) / 10.0 for a, b in zip(idx, ref_order)]))
    print(f"\n[planted bf16] max|dsim|={dsim:.2e} (rank-8 heads); top-10 positions equal to the "
          f"reference order: {100 * agree10:.1f} %, top-10 set overlap {100 * set10:.1f} %; cirr={[round(x, 2) for x in cirr]}")
    np.testing.assert_allclose(cirr, g["cirr"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(fiq, g["fiq"], rtol=0, atol=1e-4)
    assert set10 > 0.9
    assert dsim < 3e-2      # rank-8 heads: a cosine in an 8-dim subspace magnifies the bf16 feature noise ~10x (measured 1.3e-2);
                            # the 1e-3 bar on full-rank heads is asserted in tests/test_e2e_gpu.py and tests/test_benchshape_gpu.py
