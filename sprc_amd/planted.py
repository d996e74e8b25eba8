"""Planted-structure retrieval case at CIRR-val's sizes (2297 gallery images x 4181 composed queries): the workload behind the
`recall` object of the bench line (`bench.py --recall`) and tests/test_configs_gpu.py's C2-size Recall test.

There is no checkpoint and no dataset offline, so "Recall@K on CIRR-val" (BASELINE.json's metric) is measured on seeded synthetic
weights with planted structure (synth.plant_structure: scores spread over ~1.0) and planted images, with each query's target placed at
a planned rank of a REFERENCE ordering (the exact-fp32 engine's, which is 5e-6 from the reference's scores on every reference-generated
golden; tests/golden/planted_c2_subset_eva.npz holds the unmodified reference's own scores for every 22nd query of this very case, and
`reference_subset_report` -- the `recall` object of the bench line -- compares the engine with THOSE).
Everything here runs on the HIP engine; nothing imports the oracle.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from . import engine as E
from . import harness as H
from . import synth
from .config import SprcConfig

N_GALLERY, N_QUERIES = 2297, 4181
PLAN = [0, 0, 1, 2, 3, 4, 5, 8, 9, 10, 15, 30, 48, 49, 50, 51, 75, 120]     # planned rank of a query's target (reference image removed)


def planted_images(n: int = N_GALLERY, seed: int = 5):
    """-> iterator of (start, images[<=128, 3, 224, 224]) -- 0.8 x a mixture of 8 basis images + 0.4 x noise, one generator per batch
    (noise seeds 1000 + start for the original draw, seed 5; seed * 100003 + start for any other draw)"""
    g = torch.Generator().manual_seed(seed)
    basis = torch.randn((8, 3, 224, 224), generator=g)
    coef = torch.randn((n, 8), generator=g)
    for s in range(0, n, 128):
        gb = torch.Generator().manual_seed(1000 + s if seed == 5 else seed * 100003 + s)
        noise = torch.randn((min(128, n - s), 3, 224, 224), generator=gb)
        yield s, torch.einsum("nk,kchw->nchw", coef[s:s + 128], basis) * 0.8 + noise * 0.4


def planted_scores(cfg: SprcConfig, sd: Dict[str, torch.Tensor], device, dtype: str, n: int = N_GALLERY, nq: int = N_QUERIES,
                   seed: int = 5, query_index=None, images=None) -> Tuple[torch.Tensor, np.ndarray]:
    """sim[nq, n] of the `dtype` engine on the planted case (weights `sd` = synth.make_state_dict(cfg, seed, planted=True)), and the
    queries' reference indices.  query_index: only these of the nq queries (rows of sim / entries of ref in that order);
    images: a materialised list(planted_images(n, seed)) to share between several engines."""
    ids, mask, ref = synth.make_queries(nq, n, seed=seed + 1)
    ref = ref.numpy()
    if query_index is not None:
        qi = np.asarray(query_index, dtype=np.int64)
        ids, mask, ref = ids[torch.from_numpy(qi)], mask[torch.from_numpy(qi)], ref[qi]
        nq = len(qi)
    eng = E.Engine(cfg, sd, device, dtype=dtype, max_batch=233)
    feats, raws = [], []
    for s, img in (images if images is not None else planted_images(n, seed)):
        raw = eng.vit_forward(img.to(device))
        feats.append(eng.qformer_image(raw)[0])
        raws.append(raw.to(torch.float16) if dtype == "fp16" else raw)            # (the fp16 engine rounds them to fp16 anyway)
    feats, raws = torch.cat(feats), torch.cat(raws)
    fus = []
    for s in range(0, nq, 233):
        r = raws[torch.from_numpy(ref[s:s + 233]).to(device)].float()
        fus.append(eng.qformer_fuse(r, ids[s:s + 233], mask[s:s + 233])[0])
    sim = E.sim_max(torch.cat(fus), feats)
    del eng, feats, raws, fus
    torch.cuda.empty_cache()
    return sim, ref


def planned_targets(sim_ref: torch.Tensor, ref: np.ndarray, seed: int = 9, margin: float = 5e-3):
    """Targets / 6-member subset groups: query q's target is the image the REFERENCE ordering ranks at PLAN[q % len] after removing the
    reference image (validate_blip.py:258-261), moved to the nearest position within 40 ranks whose score is `margin` away from both
    neighbours where such a position exists (a K boundary then does not sit on a near-tie)."""
    nq, n = sim_ref.shape
    s = sim_ref.cpu().numpy().copy()
    s[np.arange(nq), ref] = -np.inf
    order = np.argsort(-s, axis=1, kind="stable")
    rng = np.random.default_rng(seed)
    tgt = np.zeros(nq, dtype=np.int64)
    groups = np.zeros((nq, 6), dtype=np.int64)
    for qi in range(nq):
        sc = s[qi][order[qi]]
        gaps = sc[:-1] - sc[1:]                                                          # gap below position p
        want = PLAN[qi % len(PLAN)]
        ok = [p for p in range(max(1, want - 40), want + 40) if gaps[p - 1] > margin and gaps[p] > margin] or [want]
        pos = 0 if (want == 0 and gaps[0] > margin) else min(ok, key=lambda p: (abs(p - want), p))
        tgt[qi] = order[qi][pos]
        others = [int(o) for o in rng.choice(n, 8, replace=False) if o not in (ref[qi], tgt[qi])][:4]
        groups[qi] = rng.permutation(np.array([ref[qi], tgt[qi], *others]))
    return tgt, groups


def recall_report(sim_ref: torch.Tensor, sim_eng: torch.Tensor, ref: np.ndarray) -> dict:
    """CIRR metrics (validate_blip.py:232-285: Recall@1/5/10/50, subset Recall@1/2/3) of both score matrices on targets planned on
    `sim_ref`, + the score-error distribution of `sim_eng` against `sim_ref`."""
    tgt, groups = planned_targets(sim_ref, ref)
    m_ref = H.cirr_metrics_from_sim(sim_ref, ref, tgt, groups)
    m_eng = H.cirr_metrics_from_sim(sim_eng, ref, tgt, groups)
    d = (sim_eng - sim_ref).abs()
    q = torch.quantile(d.flatten()[::7].float(), torch.tensor([0.5, 0.99, 0.999, 0.9999], device=d.device)).tolist()
    names = ["recall_at_1", "recall_at_5", "recall_at_10", "recall_at_50", "subset_recall_at_1", "subset_recall_at_2", "subset_recall_at_3"]
    # harness order: (group R@1, R@2, R@3, R@1, R@5, R@10, R@50)  (validate_blip.py:285)
    idx = [3, 4, 5, 6, 0, 1, 2]
    return {"engine": {k: round(float(m_eng[i]), 4) for k, i in zip(names, idx)},
            "reference_order": {k: round(float(m_ref[i]), 4) for k, i in zip(names, idx)},
            "top1_image_equal_pct": round(100.0 * float((sim_eng.argmax(1) == sim_ref.argmax(1)).float().mean()), 3),
            "max_abs_dsim": float(d.max()), "rms_dsim": float(d.pow(2).mean().sqrt()),
            "dsim_quantiles_50_99_99.9_99.99": [float(x) for x in q],
            "metrics_ref": [float(x) for x in m_ref], "metrics_eng": [float(x) for x in m_eng], "tgt": tgt, "groups": groups}


def reference_subset_report(cfg: SprcConfig, device, dtype: str, golden_path, images=None, sd=None, keep_scores: bool = False) -> dict:
    """The bench line's `recall` object (BASELINE.json's metric: "... + Recall@1/5/10, CIRR-val"; validate_blip.py:255-285): the engine
    against scores the UNMODIFIED REFERENCE produced on its CPU fp32 path for every 22nd query of the planted CIRR-val-sized case
    (191 queries x 2297 images; the fixture is data generated by oracle/gen_c2_subset.py, read here like any dataset file).  Targets are
    planned on the REFERENCE's ordering; both score matrices go through the same metric code (harness.cirr_metrics_from_sim, integer-exact
    against the oracle in tests/)."""
    g = np.load(golden_path, allow_pickle=False)
    n, nq, qi = int(g["n_img"]), int(g["n_q"]), g["query_index"]
    h16 = bool(int(g["trunk_fp16"])) if "trunk_fp16" in g.files else False
    if sd is None:                                      # (callers that hold the case's weights already pass them)
        sd = synth.make_state_dict(cfg, seed=int(g["seed"]), planted=True, trunk_fp16=h16)
    s_eng, ref = planted_scores(cfg, sd, device, dtype, n=n, nq=nq, seed=int(g["seed"]), query_index=qi, images=images)
    assert np.array_equal(ref, g["ref_index"]), "fixture and engine disagree on the queries' reference images"
    s_ref = torch.from_numpy(g["sim"]).to(device)
    rep = recall_report(s_ref, s_eng, ref)
    o_ref, o_eng = E.topk(s_ref.contiguous(), 10)[1], E.topk(s_eng.contiguous(), 10)[1]      # the library's stable top-k (rank.hip)
    keys = ("recall_at_1", "recall_at_5", "recall_at_10")
    d = (s_eng - s_ref).abs().flatten().float()
    q = torch.quantile(d[::2], torch.tensor([0.99, 0.999, 0.9999], device=d.device)).tolist()   # (torch.quantile takes < 2^24 elements)
    out = {"engine": rep["engine"], "reference": rep["reference_order"],
            "equal_recall_at_1_5_10": all(rep["engine"][k] == rep["reference_order"][k] for k in keys),
            "equal_subset_recalls": all(rep["engine"][k] == rep["reference_order"][k] for k in ("subset_recall_at_1", "subset_recall_at_2", "subset_recall_at_3")),
            "top1_image_equal_pct": rep["top1_image_equal_pct"],
            "top10_order_equal_pct": round(100.0 * float((o_ref == o_eng).all(1).float().mean()), 3),
            "max_abs_dsim": rep["max_abs_dsim"], "rms_dsim": rep["rms_dsim"], "scores_over_1e-3": int(((s_eng - s_ref).abs() > 1e-3).sum()),
            "dsim_quantiles_99_99.9_99.99": [float(x) for x in q],
            "scores": int(s_ref.numel()), "trunk_weights": "fp16-valued (a GPU-trained reference checkpoint: eva_vit.py:410-425)" if h16 else "fp32-valued synthetic"}
    if keep_scores:                                    # (tests: the same engine scores against a second yardstick, without a second encode)
        out["_s_eng"], out["_s_ref"] = s_eng, s_ref
    return out
