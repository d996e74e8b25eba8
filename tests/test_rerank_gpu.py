"""Stage-2 rerank on the GPU: `inference_rerank` against the REFERENCE's
Blip2QformerCirRerank.inference_rerank (tests/golden/rerank_eva.npz, oracle/gen_golden.py: rerank_goldens) and against the
oracle on a larger ragged case; the --rerank branch of generate_cirr_test_dicts against the dicts the reference's own
cirr_test_submission.generate_cirr_test_dicts(rerank=True) produced (tests/golden/metrics.json)."""
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
from sprc_amd.model import Blip2QformerCirRerank, get_model_class  # noqa: E402
from sprc_amd.tokenizer import TokenBatch  # noqa: E402

DEV = "cuda:0"


class _Tok:
    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        rows = [int(t[1:]) for t in text]
        return TokenBatch(self.ids[rows], self.mask[rows])


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 5e-3)])
def test_inference_rerank_matches_reference(golden_dir, dtype, tol):
    g = np.load(golden_dir / "rerank_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    assert get_model_class("blip2_cir_rerank") is Blip2QformerCirRerank
    model = Blip2QformerCirRerank(cfg=cfg, compute_dtype=dtype, max_batch=8)
    assert not model.load_state_dict(sd, strict=False).missing_keys
    model = model.to(DEV)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    model.tokenizer = _Tok(ids, mask)
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]))
    _, raw = model.extract_target_features(images.to(DEV))
    ref, cand = torch.from_numpy(g["ref_index"]).to(DEV), torch.from_numpy(g["cand_index"]).to(DEV)
    prob = model.inference_rerank(raw[ref], raw[cand.reshape(-1)], [f"q{i}" for i in range(int(g["n_q"]))])
    one = model.inference_rerank(raw[ref[:1]], raw[cand[0]], ["q0"])
    torch.cuda.synchronize()
    print(f"\n[rerank {dtype}] max|dprob| = {np.abs(prob.cpu().numpy() - g['prob']).max():.2e}")
    np.testing.assert_allclose(prob.cpu().numpy(), g["prob"], atol=tol, rtol=0)
    np.testing.assert_allclose(one.cpu().numpy(), g["prob_one"], atol=tol, rtol=0)
    # the cached form (K|V once per image, pairs named by index) gives the same bits
    eng = model.engine()
    kv = eng.encode_kv(raw)
    again = model.rerank_pairs(kv, ref, kv, cand, [f"q{i}" for i in range(int(g["n_q"]))])
    assert torch.equal(again.reshape(-1), prob)


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-6), ("fp16", 2e-4), ("bf16", 2e-3)])
def test_rerank_class_stage1_inference_is_text_only(golden_dir, dtype, tol):
    """`Blip2QformerCirRerank.inference` (blip2_qformer_cir_rerank.py:373-397) ignores the reference image: the Q-Former encodes the
    caption alone (no query rows, text FFN) and text_proj of its [CLS] row is matched against the gallery features.  Golden: the
    REFERENCE rerank class's `inference` on its own gallery features (the alias class used to inherit align_prompt's
    two-pass fusion, so `--rerank` started from different top-50 candidates than the reference)."""
    g = np.load(golden_dir / "rerank_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    model = Blip2QformerCirRerank(cfg=cfg, compute_dtype=dtype, max_batch=8)
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV)
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    model.tokenizer = _Tok(ids, mask)
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]))
    feats, raw = model.extract_target_features(images.to(DEV))
    ref = torch.from_numpy(g["ref_index"]).to(DEV)
    caps = [f"q{i}" for i in range(int(g["n_q"]))]
    sim = model.inference(raw[ref], feats, caps)
    other = model.inference(raw[(ref + 1) % raw.shape[0]], feats, caps)            # the reference image does not enter
    assert torch.equal(sim, other)
    err = np.abs(sim.cpu().numpy() - g["sim_stage1"]).max()
    print(f"\n[rerank class stage 1, {dtype}] max|dsim| = {err:.2e}")
    assert err < tol
    if dtype == "fp32":
        np.testing.assert_allclose(feats.cpu().numpy(), g["feats"], atol=1e-5, rtol=0)
        # ... and it is NOT the align_prompt score
        from sprc_amd.model import Blip2QformerCirAlignPrompt
        ap = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=8)
        ap.load_state_dict(sd, strict=False)
        ap = ap.to(DEV)
        ap.tokenizer = _Tok(ids, mask)
        assert np.abs(ap.inference(raw[ref], feats, caps).cpu().numpy() - g["sim_stage1"]).max() > 1e-3


def test_rerank_ragged_pairs_against_the_oracle():
    """13 pairs (not a multiple of anything), more pairs than max_batch, shared candidates, captions of every length."""
    cfg = get_config("pretrain_vitL", vit_depth=1)
    sd = synth.make_state_dict(cfg, seed=31)
    eng = E.Engine(cfg, sd, DEV, dtype="fp32", max_batch=5)
    n_img, nq, T = 7, 13, 1
    images = synth.make_images(n_img, seed=32)
    ids, mask, ref = synth.make_queries(nq, n_img, seed=33)
    cand = (ref * 3 + 1) % n_img
    with torch.no_grad():
        raw_o = O.encode_image_tokens(sd, cfg, images)
        want = O.inference_rerank(sd, cfg, raw_o[ref], raw_o[cand], ids, mask).numpy()       # B == BT: one candidate per query
    raw = eng.vit_forward(images.to(DEV))
    kv = eng.encode_kv(raw)
    got = eng.itm(kv, ref, kv, cand, ids, mask).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
    with pytest.raises(IndexError):
        eng.itm(kv, ref + 100, kv, cand, ids, mask)


def test_rerank_branch_of_the_submission_dicts_matches_the_reference(golden_dir):
    c = json.loads((golden_dir / "metrics.json").read_text())["plain"]
    sim = torch.tensor(c["sim"], dtype=torch.float32, device=DEV)
    table = torch.tensor(c["rerank_table"], dtype=torch.float32)
    names = [f"img-{i:05d}" for i in range(c["N"])]

    def rerank_fn(rows, cand):
        return table[torch.tensor(rows)[:, None], cand]

    top, sub = H.cirr_test_dicts_from_sim(sim, c["ref"], c["groups"], [1000 + i for i in range(c["nq"])], names, rerank_fn=rerank_fn)
    assert top == c["rerank_top50"] and sub == c["rerank_subset3"]
    assert top != c["test_top50"]                                     # the second stage really reorders
    s = np.asarray(c["sim"], dtype=np.float32)
    order = O.rank_stable(s)[:, :50]
    o_top, o_sub = O.cirr_test_dicts(s, np.asarray(c["ref"]), np.asarray(c["groups"]), [1000 + i for i in range(c["nq"])], names,
                                     rerank_scores=np.take_along_axis(np.asarray(c["rerank_table"], dtype=np.float32), order, axis=1))
    assert (top, sub) == (o_top, o_sub)
