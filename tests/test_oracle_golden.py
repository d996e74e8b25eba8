"""The CPU oracle (oracle/sprc_oracle.py) against outputs of the REFERENCE itself.

tests/golden/*.npz|json were produced by oracle/gen_golden.py, which imports and runs the
unmodified reference modules in the build container (R1-R8, R10).  These tests pin the oracle.
"""
import json

import numpy as np
import pytest
import torch

from oracle import sprc_oracle as O
from sprc_amd import synth
from sprc_amd.config import get_config

TOL = 2e-5     # fp32 round-off between two CPU evaluations of the same graph


def _load(golden_dir, name):
    g = np.load(golden_dir / name, allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]))
    # the synthetic generators must reproduce what the golden script saw
    np.testing.assert_array_equal(images[:, :, 0, :4].numpy(), g["image_probe"])
    probe = np.concatenate([v.flatten()[:8].numpy() for v in list(sd.values())[:3]])
    np.testing.assert_array_equal(probe, g["weight_probe"])
    return g, cfg, sd, images


@pytest.mark.parametrize("name", ["tiny_eva.npz", "tiny_clip.npz"])
def test_model_stages_match_reference(golden_dir, name):
    g, cfg, sd, images = _load(golden_dir, name)
    rows = g["rows"].tolist()
    taps = {}
    with torch.no_grad():
        feats, raw = O.extract_target_features(sd, cfg, images, taps=taps)
        vit_out = O.vit_forward(sd, cfg, images)
        img_q = O.qformer_forward(sd, cfg, sd["query_tokens"].expand(images.shape[0], -1, -1),
                                  encoder_hidden_states=raw)
    np.testing.assert_allclose(taps["patch_embed"][:, rows].numpy(), g["patch_embed"], atol=TOL, rtol=0)
    np.testing.assert_allclose(taps["block0"][:, rows].numpy(), g["block0"], atol=TOL, rtol=0)
    np.testing.assert_allclose(vit_out[:, rows].numpy(), g["vit_out"], atol=TOL, rtol=0)
    np.testing.assert_allclose(raw[:, rows].numpy(), g["raw"], atol=TOL, rtol=0)
    np.testing.assert_allclose(raw[0].numpy(), g["raw_full0"], atol=TOL, rtol=0)
    np.testing.assert_allclose(img_q.numpy(), g["img_q"], atol=TOL, rtol=0)
    np.testing.assert_allclose(feats.numpy(), g["feats"], atol=TOL, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(feats.numpy(), axis=-1), 1.0, atol=1e-5)

    ids = torch.from_numpy(g["input_ids"])
    mask = torch.from_numpy(g["attention_mask"])
    ref = torch.from_numpy(g["ref_index"])
    t2 = {}
    with torch.no_grad():
        sim = O.inference(sd, cfg, raw[ref], feats, ids, mask, taps=t2)
    np.testing.assert_allclose(t2["pass1"].numpy(), g["pass1"], atol=TOL, rtol=0)
    np.testing.assert_allclose(t2["pass2"].numpy(), g["pass2"], atol=TOL, rtol=0)
    np.testing.assert_allclose(t2["fusion"].numpy(), g["fusion"], atol=TOL, rtol=0)
    np.testing.assert_allclose(sim.numpy(), g["sim"], atol=TOL, rtol=0)


@pytest.mark.slow
@pytest.mark.parametrize("name", ["full_eva.npz", "full_clip.npz"])
def test_full_depth_matches_reference(golden_dir, name):
    path = golden_dir / name
    if not path.exists():
        pytest.skip("full-depth golden not generated")
    g, cfg, sd, images = _load(golden_dir, name)
    with torch.no_grad():
        feats, raw = O.extract_target_features(sd, cfg, images)
        sim = O.inference(sd, cfg, raw[torch.from_numpy(g["ref_index"])], feats,
                          torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    np.testing.assert_allclose(feats.numpy(), g["feats"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(raw[:, g["rows"].tolist()].numpy(), g["raw"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(sim.numpy(), g["sim"], atol=5e-5, rtol=0)


def _valid_order(sim_row, order):
    """`order` sorts fl32(1-sim) ascending (any tie order)."""
    d = O.distances(sim_row[None])[0]
    return np.all(np.diff(d[order]) >= 0)


@pytest.mark.parametrize("case", ["plain", "ties", "single_batch"])
def test_metrics_match_reference(golden_dir, case):
    c = json.loads((golden_dir / "metrics.json").read_text())[case]
    sim = np.asarray(c["sim"], dtype=np.float32)
    ref, tgt, groups = np.asarray(c["ref"]), np.asarray(c["tgt"]), np.asarray(c["groups"])
    names = [f"img-{i:05d}" for i in range(c["N"])]
    got = O.cirr_metrics(sim, ref, tgt, groups)
    fiq = O.fiq_metrics(sim, tgt)
    top, sub = O.cirr_test_dicts(sim, ref, groups, [1000 + i for i in range(c["nq"])], names)
    if not c["ties"]:
        np.testing.assert_allclose(got, c["cirr"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(fiq, c["fiq"], rtol=0, atol=1e-4)
        assert top == c["test_top50"] and sub == c["test_subset3"]
    else:
        # torch.argsort in the reference is unstable: with engineered ties the reference's order is
        # ONE valid order, the contract's stable order another.  Both must sort the same distances,
        # so the sorted distance sequences agree position by position.
        name_to_i = {n: i for i, n in enumerate(names)}
        d = O.distances(sim)
        for q, (pid, ref_top) in enumerate(sorted(c["test_top50"].items(), key=lambda kv: int(kv[0]))):
            mine = top[pid]
            a = d[q, [name_to_i[n] for n in ref_top]]
            b = d[q, [name_to_i[n] for n in mine]]
            np.testing.assert_array_equal(a, b)
        # recall differs from the reference's only through ties straddling K or involving the target
        assert np.all(np.abs(np.asarray(got) - np.asarray(c["cirr"])) <= 100.0 * 8 / c["nq"])


def test_rank_of_is_position_in_stable_order():
    rng = np.random.default_rng(3)
    sim = (np.round(rng.uniform(0, 1, (9, 40)) * 8) / 8).astype(np.float32)
    order = O.rank_stable(sim)
    listed = rng.integers(0, 40, (9, 5))
    r = O.rank_of(sim, listed)
    for q in range(9):
        for l in range(5):
            assert order[q, r[q, l]] == listed[q, l]


def test_caption_processing_matches_reference(golden_dir):
    c = json.loads((golden_dir / "captions.json").read_text())
    for raw, want in c["pre_caption"]:
        assert O.pre_caption(raw) == want
    for c1, c2, composed, processed in c["fiq"]:
        assert O.fiq_caption(c1, c2) == composed
        assert O.pre_caption(composed) == processed


def test_planted_structure_case_matches_reference(golden_dir):
    """Planted-structure weights (synth.plant_structure): scores spread over > 1.0, targets at planned ranks.  The oracle
    reproduces the reference's scores, its stable order wherever the reference's own neighbouring scores differ by more
    than 1e-5, and the numbers the reference's metric code printed for them."""
    g = np.load(golden_dir / "planted_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    seed, n_img = int(g["seed"]), int(g["n_img"])
    sd = synth.make_state_dict(cfg, seed=seed, planted=True)
    images = synth.make_images(n_img, seed=seed, planted=True)
    np.testing.assert_array_equal(images[:4, :, 0, :4].numpy(), g["image_probe"])
    ids, mask, ref = (torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "ref_index"))
    with torch.no_grad():
        feats, raw = O.extract_target_features(sd, cfg, images)
        fusion = O.fuse_queries(sd, cfg, raw[ref], ids, mask)
        sim = O.similarity(fusion, feats).numpy()
    np.testing.assert_allclose(feats[:4].numpy(), g["feats_head"], atol=TOL, rtol=0)
    np.testing.assert_allclose(fusion.numpy(), g["fusion"], atol=TOL, rtol=0)
    np.testing.assert_allclose(sim, g["sim"], atol=TOL, rtol=0)
    spread = np.sort(g["sim"], axis=1)
    assert np.mean(spread[:, -1] - spread[:, 0]) > 0.3                      # the point of planting structure
    ref_i, tgt, grp = g["ref_index"], g["tgt_index"], g["groups"]
    assert O.cirr_metrics(g["sim"], ref_i, tgt, grp) == pytest.approx(tuple(g["cirr"]), abs=1e-4)
    assert O.fiq_metrics(g["sim"], tgt) == pytest.approx(tuple(g["fiq"]), abs=1e-4)
    assert O.cirr_metrics(sim, ref_i, tgt, grp) == pytest.approx(tuple(g["cirr"]), abs=1e-4)   # on the oracle's own scores too
    dicts = json.loads(str(g["test_dicts"]))
    names = [f"img-{i:05d}" for i in range(n_img)]
    top, sub = O.cirr_test_dicts(g["sim"], ref_i, grp, [1000 + i for i in range(len(ref_i))], names)
    assert top == dicts["top"] and sub == dicts["sub"]


def _grad_functionals(name, g):
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    flat = g.detach().double().flatten().cpu()
    r = torch.randn((2, flat.numel()), generator=gen, dtype=torch.float32).double()
    probe = torch.randint(0, flat.numel(), (8,), generator=gen)
    return np.concatenate([[float(flat.norm())], (r @ flat).numpy(), [float(flat.sum())], flat[probe].numpy()])


def check_gradients_against_golden(g, grads, rel_tol):
    """grads: {state-dict name: tensor}; golden rows are [||g||, <g, r1>, <g, r2>, sum g, 8 probes] (oracle/gen_golden.py).  Every
    functional within rel_tol x ||g|| x (its natural scale: sqrt(n) for sum g, 1 for the rest)."""
    names = [str(n) for n in g["grad_names"]]
    assert set(names) == set(grads), (sorted(set(names) ^ set(grads))[:8])
    worst = 0.0
    for name, want in zip(names, g["grad_values"]):
        got = _grad_functionals(name, grads[name])
        if want[0] < 1e-8:              # key biases: the softmax is invariant to a per-query shift, their gradient is rounding noise
            assert got[0] < 1e-6, f"{name}: ||g|| = {got[0]:.2e}, the reference's is {want[0]:.2e} (mathematically zero)"
            continue
        norm = want[0]
        scale = np.array([1.0, 1.0, 1.0, np.sqrt(grads[name].numel())] + [1.0] * 8) * norm
        err = np.abs(got - want) / scale
        worst = max(worst, float(err.max()))
        assert err.max() < rel_tol, f"{name}: functional error {err.max():.2e} of ||g|| = {norm:.3e} ({got[:4]} vs {want[:4]})"
    return worst


def test_training_gradients_match_reference(golden_dir):
    """N4 backward: torch autograd over the oracle's `training_losses` == the reference's forward + backward (eval mode) on the
    train_eva triplets, for all 337 trainable tensors (incl. ln_vision, temp, prompt_tokens, both embedding tables)."""
    g = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    B = int(g["batch"])
    images = synth.make_images(2 * B, seed=int(g["seed"]))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    losses, grads = O.training_gradients(sd, cfg, images[:B], images[B:], ids, mask)
    for k in ("loss_itc", "loss_rtc", "loss_align"):
        assert abs(float(losses[k]) - float(g[k])) < 1e-5
    worst = check_gradients_against_golden(g, grads, 1e-4)
    assert all(not n.startswith("visual_encoder.") for n in grads) and "ln_vision.weight" in grads and "temp" in grads
    print(f"\n[oracle training gradients] worst functional error / ||g|| = {worst:.2e} over {len(grads)} tensors")


def test_training_gradients_with_dropout_match_reference(golden_dir):
    """The reference AS IT TRAINS (blip_fine_tune_2.py:290: `.train()`, Q-Former dropout p = 0.1 at Qformer.py:113,264,293,379; ViT in
    eval) with the masks injected by oracle/gen_golden.py: the oracle regenerates the same counter-based masks (drop_keep) and must
    reproduce the three losses and all 337 gradients; and the masks matter -- the eval-mode losses are different numbers."""
    g = np.load(golden_dir / "train_dropout_eva.npz", allow_pickle=False)
    g0 = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    assert float(g["dropout_p"]) == 0.1 and int(g["dropout_sites"]) == 61 + 37 + 49 + 37      # per pass: embeddings + 2 x 12 self-attention (+ 2 x 6 cross) + the FFN outputs
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    B = int(g["batch"])
    images = synth.make_images(2 * B, seed=int(g["seed"]))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    losses, grads = O.training_gradients(sd, cfg, images[:B], images[B:], ids, mask, drop=(int(g["drop_seed_effective"]), float(g["dropout_p"])))
    for k in ("loss_itc", "loss_rtc", "loss_align"):
        assert abs(float(losses[k]) - float(g[k])) < 1e-5
    assert abs(float(g["loss_itc"]) - float(g0["loss_itc"])) > 1e-3          # dropout changes the numbers
    worst = check_gradients_against_golden(g, grads, 1e-4)
    print(f"\n[oracle training gradients, dropout on] worst functional error / ||g|| = {worst:.2e} over {len(grads)} tensors")
    # the mask itself: keep rate, determinism, independence of sites
    k1 = O.drop_keep(123, O.drop_site(0, 3, O.DROP_FFN_Q), 200000, 0.1)
    k2 = O.drop_keep(123, O.drop_site(0, 3, O.DROP_FFN_T), 200000, 0.1)
    assert abs(k1.mean() - 0.9) < 3e-3 and abs((k1 & k2).mean() - 0.81) < 4e-3 and np.array_equal(k1, O.drop_keep(123, O.drop_site(0, 3, O.DROP_FFN_Q), 200000, 0.1))


def test_rerank_matches_reference(golden_dir):
    """N2: the oracle's inference_rerank against the reference's Blip2QformerCirRerank.inference_rerank (514-token
    cross-attention + itm_head + softmax), batched (3 queries x 4 candidates) and the single-query branch."""
    g = np.load(golden_dir / "rerank_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"]))
    np.testing.assert_array_equal(images[:, :, 0, :4].numpy(), g["image_probe"])
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    ref, cand = torch.from_numpy(g["ref_index"]), torch.from_numpy(g["cand_index"])
    with torch.no_grad():
        raw = O.encode_image_tokens(sd, cfg, images)
        prob = O.inference_rerank(sd, cfg, raw[ref], raw[cand.reshape(-1)], ids, mask)
        one = O.inference_rerank(sd, cfg, raw[ref[:1]], raw[cand[0]], ids[:1], mask[:1])
    np.testing.assert_allclose(prob.numpy(), g["prob"], atol=TOL, rtol=0)
    np.testing.assert_allclose(one.numpy(), g["prob_one"], atol=TOL, rtol=0)
    assert np.all((g["prob"] > 0) & (g["prob"] < 1)) and np.ptp(g["prob"]) > 0.02
    # the rerank class's own stage-1 score: text only (blip2_qformer_cir_rerank.py:373-397)
    with torch.no_grad():
        feats, _ = O.extract_target_features(sd, cfg, images)
        s1 = O.inference_rerank_stage1(sd, cfg, feats, ids, mask)
    np.testing.assert_allclose(feats.numpy(), g["feats"], atol=TOL, rtol=0)
    np.testing.assert_allclose(s1.numpy(), g["sim_stage1"], atol=TOL, rtol=0)


def test_training_losses_match_reference(golden_dir):
    """N4: the oracle's training forward (three losses) against the reference's Blip2QformerCirAlignPrompt.forward."""
    g = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    B = int(g["batch"])
    images = synth.make_images(2 * B, seed=int(g["seed"]))
    np.testing.assert_array_equal(images[:, :, 0, :4].numpy(), g["image_probe"])
    with torch.no_grad():
        out = O.training_losses(sd, cfg, images[:B], images[B:], torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    for k in ("loss_itc", "loss_rtc", "loss_align"):
        assert float(out[k]) == pytest.approx(float(g[k]), abs=2e-5), k


GPUREF_CASES = ["planted_full_eva", "planted_full_eva_s1", "planted_full_clip", "planted_full_clip_s1", "planted_big_eva", "planted_big_eva_s3"]


@pytest.mark.parametrize("case", GPUREF_CASES + ["planted_full_eva_h16", "planted_full_clip_h16", "planted_big_eva_h16"])
def test_gpuref_fixture_belongs_to_its_golden(golden_dir, case):
    """`<case>_gpuref.npz` = the unmodified reference in its GPU arithmetic (fp16-autocast ViT on fp16 trunk weights, fp32 Q-Former;
    oracle/gen_gpuref.py) on the inputs of `<case>.npz`: same sizes and seeds, the stored error summary is the one the arrays give,
    and the trunk really ran on fp16 matrices."""
    g = np.load(golden_dir / f"{case}.npz", allow_pickle=False)
    gr = np.load(golden_dir / f"{case}_gpuref.npz", allow_pickle=False)
    assert gr["sim_gpuref"].shape == g["sim"].shape and gr["sim_gpuref"].dtype == np.float32
    for k in ("seed", "n_img", "n_q", "vit_depth"):
        assert int(gr[k]) == int(g[k]), k
    d = gr["sim_gpuref"] - g["sim"]
    assert float(np.abs(d).max()) == pytest.approx(float(gr["max_err"]), abs=1e-12)
    assert float(np.sqrt((d.astype(np.float64) ** 2).mean())) == pytest.approx(float(gr["rms_err"]), rel=1e-9)
    assert int((np.abs(d) > 1e-3).sum()) == int(gr["n_over_1e3"])
    assert "torch.float16" in [str(x) for x in gr["trunk_weight_dtypes"]]
    assert 1e-5 < float(gr["rms_err"]) < 1e-3                     # a 16-bit path: neither the fp32 one again nor garbage


def test_the_references_own_gpu_arithmetic_does_not_hold_1e_3(golden_dir):
    """The evidence behind DESIGN.md section 4.3: measured against its own CPU fp32 path (the north star's yardstick), the reference's
    GPU arithmetic exceeds 1e-3 on three of the six full-depth planted cases -- by 2.3x on CLIP ViT-L, whose residual stream is fp16
    under autocast (clip_vit.py:173-182).  A 1e-3 bar on cosine scores is therefore a property of the CPU path, not of what the
    reference computes on a GPU; the engine is held to the reference's GPU-path error instead (tests/test_fp16_gpu.py)."""
    over = {}
    for case in GPUREF_CASES:
        gr = np.load(golden_dir / f"{case}_gpuref.npz", allow_pickle=False)
        over[case] = (float(gr["max_err"]), int(gr["n_over_1e3"]))
    assert sum(m > 1e-3 for m, _ in over.values()) >= 3, over
    assert over["planted_full_clip"][0] > 2e-3 and over["planted_big_eva"][1] >= 40


@pytest.mark.parametrize("case", ["planted_c2_subset_eva", "planted_c2_subset_eva_h16"])
def test_c2_size_subset_fixture_is_the_oracles_case(golden_dir, case):
    """The CIRR-val-sized reference fixtures (oracle/gen_c2_subset.py: 191 queries x 2297 images scored by the UNMODIFIED reference) are far
    too large to re-score on the CPU here (~40 min); what pins them to the oracle: the reference's features of the images the fixture
    probes (every 256th, first two query tokens) are what the oracle computes for those images from the same seeded weights / images --
    checked on the first probe -- and the fp32 HIP engine then reproduces all 438 727 scores within 1e-4 (tests/test_configs_gpu.py)."""
    from sprc_amd import planted as P
    g = np.load(golden_dir / f"{case}.npz", allow_pickle=False)
    h16 = bool(int(g["trunk_fp16"])) if "trunk_fp16" in g.files else False
    assert h16 == case.endswith("_h16") and g["sim"].shape == (191, 2297) and int(g["n_q"]) == 4181
    cfg = get_config(str(g["model_type"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]), planted=True, trunk_fp16=h16)
    _, img = next(iter(P.planted_images(int(g["n_img"]), int(g["seed"]))))
    with torch.no_grad():
        feats, _ = O.extract_target_features(sd, cfg, img[:1])
    np.testing.assert_allclose(feats[0, :2].numpy(), g["feats_probe"][0], atol=2e-5, rtol=0)
    ids, mask, ref = synth.make_queries(int(g["n_q"]), int(g["n_img"]), seed=int(g["seed"]) + 1)
    assert np.array_equal(ref.numpy()[g["query_index"]], g["ref_index"])
