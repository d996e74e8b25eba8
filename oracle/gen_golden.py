#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE (/root/reference) in the build container.

TEST INFRASTRUCTURE.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference's
unmodified modules executed here with seeded synthetic weights/inputs
(sprc_amd/synth.py).  Only data (inputs/expected outputs) is written; no reference
source travels.  Re-run with:  python oracle/gen_golden.py [--full]

Files written:
  tests/golden/tiny_eva.npz    ViT-g width, depth 2, 12-layer Q-Former : every stage boundary
  tests/golden/tiny_clip.npz   ViT-L width, depth 2                    : every stage boundary
  tests/golden/full_eva.npz    full depth (39 blocks), 2 images, 3 queries   (--full, ~3 min)
  tests/golden/full_clip.npz   full depth ViT-L (23 blocks), 2 images, 3 queries   (--full)
  tests/golden/planted_eva.npz planted-structure weights (scores spread > 1.0), depth-4 ViT-g, 160 gallery x 72 queries:
                               scores, planned targets, the reference's own metrics / submission dicts on them
  tests/golden/planted_full_eva.npz  the same at FULL depth (39 blocks), 96 gallery x 48 queries (--full): the case the
                               16-bit engines are held to max|dsim| < 1e-3 on
  tests/golden/train_eva.npz   training forward: the reference's three losses (loss_itc, loss_rtc, loss_align) on 5 triplets
  tests/golden/train_dropout_eva.npz  the same in TRAIN mode (Q-Former dropout p = 0.1 as blip_fine_tune_2.py:290 runs it), masks injected
  tests/golden/rerank_eva.npz  stage-2 rerank: the reference's Blip2QformerCirRerank.inference_rerank on 3 queries x 4 candidates
  tests/golden/metrics.json    reference compute_cirr_val_metrics / compute_fiq_val_metrics /
                               generate_cirr_test_dicts on synthetic sims (with engineered ties)
  tests/golden/captions.json   reference BlipCaptionProcessor + FashionIQ caption composition
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from oracle import ref_import  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

GOLD = ROOT / "tests" / "golden"
ROWS = [0, 1, 100, 256]          # token rows kept from [B,257,D] tensors (keeps fixtures small)


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def model_goldens(model_type: str, vit_depth, n_img: int, n_q: int, out: Path, seed: int = 0):
    cfg = get_config(model_type, vit_depth=vit_depth)
    sd = synth.make_state_dict(cfg, seed=seed)
    model = ref_import.build_reference_model(cfg, sd)
    images = synth.make_images(n_img, seed=seed)
    ids, mask, ref = synth.make_queries(n_q, n_img, seed=seed + 1)
    ids[0, :] = 0                     # one fully padded-but-CLS caption: shortest legal text
    ids[0, 0], ids[0, 1] = 101, 102
    mask[0, :] = 0
    mask[0, :2] = 1
    taps = {}
    vit = model.visual_encoder
    hooks = []
    if cfg.vit.kind == "eva_g":
        hooks.append(vit.blocks[0].register_forward_pre_hook(lambda m, a: taps.__setitem__("patch_embed", a[0].clone())))
        hooks.append(vit.blocks[0].register_forward_hook(lambda m, a, o: taps.__setitem__("block0", o.clone())))
    else:
        rb = vit.transformer.resblocks[0]   # LND layout inside the transformer (clip_vit.py:180-182)
        hooks.append(rb.register_forward_pre_hook(lambda m, a: taps.__setitem__("patch_embed", a[0].permute(1, 0, 2).clone())))
        hooks.append(rb.register_forward_hook(lambda m, a, o: taps.__setitem__("block0", o.permute(1, 0, 2).clone())))
    with torch.no_grad():
        vit_out = model.visual_encoder(images)
        feats, raw = model.extract_target_features(images, mode="mean")
        model.tokenizer.set_next(ids, mask)
        q_taps = {}
        bert = model.Qformer.bert
        calls = []
        h = bert.register_forward_hook(lambda m, a, k, o: calls.append(o.last_hidden_state.clone()), with_kwargs=True)
        sim = model.inference(raw[ref], feats, ["caption"] * n_q)
        h.remove()
        # image-only Q-Former output (call shape (i))
        img_q = bert(query_embeds=model.query_tokens.expand(n_img, -1, -1), encoder_hidden_states=raw,
                     encoder_attention_mask=torch.ones(raw.shape[:-1], dtype=torch.long),
                     return_dict=True).last_hidden_state
        fusion = torch.nn.functional.normalize(model.text_proj(calls[1][:, 32, :]), dim=-1)
    for hk in hooks:
        hk.remove()
    if sim.dim() == 1:
        sim = sim.unsqueeze(0)
    sample = {k: _np(v.flatten()[:8]) for k, v in list(sd.items())[:3]}
    np.savez_compressed(
        out,
        model_type=model_type, vit_depth=cfg.vit.depth, seed=seed, n_img=n_img, n_q=n_q, rows=np.array(ROWS),
        image_probe=_np(images[:, :, 0, :4]), weight_probe=np.concatenate(list(sample.values())),
        patch_embed=_np(taps["patch_embed"][:, ROWS]), block0=_np(taps["block0"][:, ROWS]),
        vit_out=_np(vit_out[:, ROWS]), raw=_np(raw[:, ROWS]), raw_full0=_np(raw[0]),
        img_q=_np(img_q), feats=_np(feats),
        input_ids=ids.numpy(), attention_mask=mask.numpy(), ref_index=ref.numpy(),
        pass1=_np(calls[0]), pass2=_np(calls[1]), fusion=_np(fusion), sim=_np(sim),
    )
    print(f"wrote {out}  feats{tuple(feats.shape)} sim{tuple(sim.shape)}  sim range [{sim.min():.4f},{sim.max():.4f}]")


# ------------------------------------------------------------------------------------------
def _synthetic_retrieval(nq: int, N: int, seed: int, ties: bool):
    rng = np.random.default_rng(seed)
    sim = rng.uniform(-0.2, 0.9, size=(nq, N)).astype(np.float32)
    if ties:   # coarse grid -> many exact ties in fl32(1 - sim), incl. around the target
        sim = (np.round(sim * 16) / 16).astype(np.float32)
    names = [f"img-{i:05d}" for i in range(N)]
    ref = rng.integers(0, N, size=nq)
    tgt = (ref + 1 + rng.integers(0, N - 1, size=nq)) % N
    groups = np.zeros((nq, 6), dtype=np.int64)
    for q in range(nq):
        others = [i for i in rng.permutation(N) if i != ref[q] and i != tgt[q]][:4]
        g = np.array([ref[q], tgt[q], *others])
        groups[q] = rng.permutation(g)
    return sim, names, ref, tgt, groups


def metrics_goldens(out: Path):
    vb, cts = ref_import.import_harness()
    from torch.utils.data import Dataset

    class FakeModel:
        device = torch.device("cpu")

        def __init__(self, sim):
            self.sim = torch.from_numpy(sim)

        def inference(self, reference_embeds, target_feats, captions):
            rows = [int(c[1:]) for c in captions]
            return self.sim[rows]

        def inference_rerank(self, reference_feats, target_feats, captions):
            # stage-2 double: P(match) of (query, gallery image) from a fixed table; the "raw embeddings" carry the image index
            rows = torch.tensor([int(c[1:]) for c in captions])
            cand = target_feats.view(len(rows), -1).long()
            return self.rerank_table[rows[:, None], cand].reshape(-1)

    class CirrVal(Dataset):
        def __init__(self, names, ref, tgt, groups):
            self.names, self.ref, self.tgt, self.groups = names, ref, tgt, groups

        def __len__(self):
            return len(self.ref)

        def __getitem__(self, i):
            return (self.names[self.ref[i]], self.names[self.tgt[i]], f"q{i}",
                    [self.names[g] for g in self.groups[i]])

    class CirrTest(CirrVal):
        def __getitem__(self, i):
            return (1000 + i, self.names[self.ref[i]], f"q{i}", [self.names[g] for g in self.groups[i]])

    class FiqVal(CirrVal):
        dress_types = ["dress"]

        def __getitem__(self, i):     # captions: two strings; the fake processor recovers the index
            return self.names[self.ref[i]], self.names[self.tgt[i]], [f"q{i}", "x"]

    txt = {"eval": lambda c: c}
    cases = {}
    for name, nq, N, seed, ties in [("plain", 70, 97, 0, False), ("ties", 70, 97, 1, True),
                                    ("single_batch", 5, 60, 2, True)]:
        sim, names, ref, tgt, groups = _synthetic_retrieval(nq, N, seed, ties)
        fm = FakeModel(sim)
        feats = (torch.zeros(N, 1), torch.zeros(N, 1))
        cirr = vb.compute_cirr_val_metrics(CirrVal(names, ref, tgt, groups), fm, feats, names, txt)
        # FashionIQ path builds "{Cap1} and {cap2}" -> "Q<i> and x"; map it back to q<i>
        fiq_txt = {"eval": lambda c: "q" + c.split(" ")[0][1:]}
        fiq = vb.compute_fiq_val_metrics(FiqVal(names, ref, tgt, groups), fm, feats, names, fiq_txt)
        top, sub = cts.generate_cirr_test_dicts(CirrTest(names, ref, tgt, groups), fm, feats, names, txt, False)
        # the reference's --rerank branch (cirr_test_submission.py:88-112) with the stage-2 scores of the table above
        table = np.random.default_rng(seed + 50).uniform(0.05, 0.95, size=(nq, N)).astype(np.float32)
        fm.rerank_table = torch.from_numpy(table)
        feats_rr = (torch.zeros(N, 1), torch.arange(N, dtype=torch.float32).unsqueeze(1))
        top_rr, sub_rr = cts.generate_cirr_test_dicts(CirrTest(names, ref, tgt, groups), fm, feats_rr, names, txt, True)
        cases[name] = dict(nq=nq, N=N, seed=seed, ties=ties, rerank_table=table.tolist(), rerank_top50=top_rr, rerank_subset3=sub_rr,
                           sim=sim.tolist(), ref=ref.tolist(), tgt=tgt.tolist(), groups=groups.tolist(),
                           cirr=list(cirr), fiq=list(fiq), test_top50=top, test_subset3=sub)
        print(name, "cirr", [round(x, 3) for x in cirr], "fiq", fiq)
    out.write_text(json.dumps(cases))
    print("wrote", out)


PLANT_RANKS = [0, 0, 1, 2, 3, 4, 5, 8, 9, 10, 15, 30, 48, 49, 50, 51, 75, 120]   # planned rank of the target (reference removed)


def planted_targets(sim: np.ndarray, ref: np.ndarray, seed: int = 0, margin: float = 5e-3):
    """Targets / subset groups for a planted-structure retrieval case: query q's target is the gallery image the
    REFERENCE ranks at position PLANT_RANKS[q % len] (after removing the reference image, validate_blip.py:258-261),
    moved to the nearest position whose score differs from both neighbours by more than `margin` (so that a K boundary
    never sits on a near-tie: Recall@K of a bf16 run is then decided by structure, not by rounding)."""
    rng = np.random.default_rng(seed)
    nq, N = sim.shape
    d = (np.float32(1.0) - sim.astype(np.float32)).astype(np.float32)
    order = np.argsort(d, axis=1, kind="stable")
    tgt = np.zeros(nq, dtype=np.int64)
    groups = np.zeros((nq, 6), dtype=np.int64)
    for q in range(nq):
        o = order[q][order[q] != ref[q]]
        dq = d[q][o]
        want = PLANT_RANKS[q % len(PLANT_RANKS)]
        ok = [p for p in range(1, N - 2) if dq[p] - dq[p - 1] > margin and dq[p + 1] - dq[p] > margin]
        if want == 0 and dq[1] - dq[0] > margin:
            pos = 0
        else:
            pos = min(ok, key=lambda p: (abs(p - want), p))
        tgt[q] = o[pos]
        others = [i for i in rng.permutation(N) if i != ref[q] and i != tgt[q]][:4]
        groups[q] = rng.permutation(np.array([ref[q], tgt[q], *others]))
    return tgt, groups


def planted_goldens(out: Path, n_img: int = 160, n_q: int = 72, vit_depth: int = 4, seed: int = 0, model_type: str = "pretrain",
                    trunk_fp16: bool = False):
    """Planted-structure ordering fixture: depth-4 ViT-g + the full Q-Former run by the REFERENCE on 160 gallery images
    and 72 composed queries; scores spread over > 1.0; targets at planned ranks; the reference's own
    compute_cirr_val_metrics / compute_fiq_val_metrics / generate_cirr_test_dicts evaluated on its own scores."""
    from torch.utils.data import Dataset
    cfg = get_config(model_type, vit_depth=vit_depth)
    # trunk_fp16: the checkpoint as a GPU-trained reference model saves it -- trunk Conv / Linear tensors hold fp16 values
    # (synth.round_trunk_to_fp16); the reference's CPU path below still computes in fp32 (models/__init__.py:246-247)
    sd = synth.make_state_dict(cfg, seed=seed, planted=True, trunk_fp16=trunk_fp16)
    model = ref_import.build_reference_model(cfg, sd)
    images = synth.make_images(n_img, seed=seed, planted=True)
    ids, mask, ref = synth.make_queries(n_q, n_img, seed=seed + 1)
    feats, raw = [], []
    with torch.no_grad():
        for s in range(0, n_img, 32):
            f, r = model.extract_target_features(images[s:s + 32], mode="mean")
            feats.append(f); raw.append(r)
        feats, raw = torch.cat(feats), torch.cat(raw)
        sims, fus = [], []
        calls = []
        h = model.Qformer.bert.register_forward_hook(lambda m, a, k, o: calls.append(o.last_hidden_state.clone()), with_kwargs=True)
        for s in range(0, n_q, 24):
            model.tokenizer.set_next(ids[s:s + 24], mask[s:s + 24])
            calls.clear()
            sims.append(model.inference(raw[ref[s:s + 24]], feats, ["caption"] * len(ids[s:s + 24])))
            fus.append(torch.nn.functional.normalize(model.text_proj(calls[1][:, 32, :]), dim=-1))
        h.remove()
    sim, fusion = _np(torch.cat(sims)), _np(torch.cat(fus))
    refn = ref.numpy()
    tgt, groups = planted_targets(sim, refn, seed)
    # the reference's own metric code on the reference's own scores
    vb, cts = ref_import.import_harness()
    names = [f"img-{i:05d}" for i in range(n_img)]

    class FakeModel:
        device = torch.device("cpu")

        def inference(self, reference_embeds, target_feats, captions):
            return torch.from_numpy(sim[[int(c[1:]) for c in captions]])

    class CirrVal(Dataset):
        def __len__(self):
            return n_q

        def __getitem__(self, i):
            return names[refn[i]], names[tgt[i]], f"q{i}", [names[g] for g in groups[i]]

    class CirrTest(CirrVal):
        def __getitem__(self, i):
            return 1000 + i, names[refn[i]], f"q{i}", [names[g] for g in groups[i]]

    class FiqVal(CirrVal):
        dress_types = ["dress"]

        def __getitem__(self, i):
            return names[refn[i]], names[tgt[i]], [f"q{i}", "x"]

    txt = {"eval": lambda c: c}
    fk = (torch.zeros(n_img, 1), torch.zeros(n_img, 1))
    cirr = vb.compute_cirr_val_metrics(CirrVal(), FakeModel(), fk, names, txt)
    fiq = vb.compute_fiq_val_metrics(FiqVal(), FakeModel(), fk, names, {"eval": lambda c: "q" + c.split(" ")[0][1:]})
    top, sub = cts.generate_cirr_test_dicts(CirrTest(), FakeModel(), fk, names, txt, False)
    s_sorted = np.sort(sim, axis=1)
    gaps = np.diff(s_sorted, axis=1)
    np.savez_compressed(
        out, model_type=model_type, vit_depth=cfg.vit.depth, seed=seed, n_img=n_img, n_q=n_q, trunk_fp16=int(trunk_fp16),
        image_probe=_np(images[:4, :, 0, :4]), input_ids=ids.numpy(), attention_mask=mask.numpy(), ref_index=refn,
        tgt_index=tgt, groups=groups, sim=sim, fusion=fusion, feats_head=_np(feats[:4]), raw_head=_np(raw[:2][:, ROWS]),
        cirr=np.array(cirr, dtype=np.float64), fiq=np.array(fiq, dtype=np.float64),
        test_dicts=np.array(json.dumps({"top": top, "sub": sub})))
    print(f"wrote {out}: sim range [{sim.min():.3f},{sim.max():.3f}] mean row spread {np.mean(s_sorted[:, -1] - s_sorted[:, 0]):.3f} "
          f"median gap {np.median(gaps):.1e} gaps<1e-5: {int((gaps < 1e-5).sum())}/{gaps.size}  cirr {[round(x, 2) for x in cirr]} fiq {fiq}")


def rerank_goldens(out: Path, seed: int = 2):
    """Stage-2 rerank fixture (N2): the REFERENCE's Blip2QformerCirRerank.inference_rerank (blip2_qformer_cir_rerank.py:
    399-445) on 3 queries x 4 candidates (depth-2 ViT-g, full Q-Former), plus the reference's generate_cirr_test_dicts with
    rerank=True on a fake model that returns those probabilities."""
    cfg = get_config("pretrain", vit_depth=2)
    sd = synth.make_state_dict(cfg, seed=seed)
    model = ref_import.build_reference_model(cfg, sd, variant="rerank")
    n_img, n_q, T = 6, 3, 4
    images = synth.make_images(n_img, seed=seed)
    ids, mask, ref = synth.make_queries(n_q, n_img, seed=seed + 1)
    cand = torch.tensor([[(int(ref[q]) + 1 + t) % n_img for t in range(T)] for q in range(n_q)])
    with torch.no_grad():
        _, raw = model.extract_target_features(images, mode="mean")
        model.tokenizer.set_next(ids, mask)
        prob = model.inference_rerank(raw[ref], raw[cand.reshape(-1)], ["caption"] * n_q)
        model.tokenizer.set_next(ids[:1], mask[:1])
        prob_one = model.inference_rerank(raw[ref[:1]], raw[cand[0]], ["caption"])          # the B == 1 branch (:404-406)
        # the rerank class's own STAGE-1 score (:373-397): text only, against the gallery features of the same model
        feats, _ = model.extract_target_features(images, mode="mean")
        model.tokenizer.set_next(ids, mask)
        sim_stage1 = model.inference(raw[ref], feats, ["caption"] * n_q)
    np.savez_compressed(out, model_type="pretrain", vit_depth=2, seed=seed, n_img=n_img, n_q=n_q, T=T,
                        image_probe=_np(images[:, :, 0, :4]), input_ids=ids.numpy(), attention_mask=mask.numpy(),
                        ref_index=ref.numpy(), cand_index=cand.numpy(), prob=_np(prob), prob_one=_np(prob_one),
                        sim_stage1=_np(sim_stage1), feats=_np(feats))
    print(f"wrote {out}: prob {prob.numpy().round(4).tolist()}")


GRAD_WEIGHTS = {"loss_itc": 1.0, "loss_rtc": 0.4, "loss_align": 0.4}      # blip_fine_tune_2.py:293-299 with its defaults (:379-380)


def grad_functionals(name: str, g: torch.Tensor) -> np.ndarray:
    """A gradient tensor as 12 numbers (the Q-Former's gradients are 700 MB): [||g||, <g, r1>, <g, r2>, sum(g), 8 probe entries];
    r1, r2 ~ N(0, 1) and the probe positions come from a generator seeded by crc32(name).  Any error pattern moves <g, r>."""
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    flat = g.detach().double().flatten()
    r = torch.randn((2, flat.numel()), generator=gen, dtype=torch.float32).double()
    probe = torch.randint(0, flat.numel(), (8,), generator=gen)
    return np.concatenate([[float(flat.norm())], (r @ flat).numpy(), [float(flat.sum())], flat[probe].numpy()])


class _InjectedDropout(torch.nn.Module):
    """Stands in for ONE nn.Dropout instance of the reference's Q-Former in train mode: the same arithmetic (x * keep / (1 - p)) with
    the keep mask of oracle.sprc_oracle.drop_keep -- a counter-based hash both the HIP kernels and the oracle regenerate -- instead of
    a draw from torch's global generator.  `state` carries the effective seed and the running pass number (one bert.forward = one pass)."""

    def __init__(self, state: dict, layer: int, kind: int):
        super().__init__()
        self.state, self.layer, self.kind = state, layer, kind

    def forward(self, x):
        from oracle import sprc_oracle as O
        if not self.training:
            return x
        site = O.drop_site(self.state["pass"], self.layer, self.kind)
        self.state["calls"].append(site)
        keep = torch.from_numpy(O.drop_keep(self.state["seed"], site, x.numel(), self.state["p"])).view(x.shape)
        return x * keep.to(x.dtype) * (1.0 / (1.0 - self.state["p"]))


def inject_dropout_masks(model, seed: int, p: float) -> dict:
    """Replace every nn.Dropout of model.Qformer.bert (Qformer.py:75,158,288,374: embeddings, attention probabilities, BertSelfOutput,
    BertOutput) by an `_InjectedDropout` numbered as oracle.drop_site numbers them; a forward pre-hook on bert counts the passes."""
    from oracle import sprc_oracle as O
    bert = model.Qformer.bert
    state = {"seed": seed, "p": p, "pass": -1, "calls": []}
    assert isinstance(bert.embeddings.dropout, torch.nn.Dropout) and abs(bert.embeddings.dropout.p - p) < 1e-12
    bert.embeddings.dropout = _InjectedDropout(state, 0, O.DROP_EMB)
    for l, layer in enumerate(bert.encoder.layer):
        layer.attention.self.dropout = _InjectedDropout(state, l, O.DROP_SELF_P)
        layer.attention.output.dropout = _InjectedDropout(state, l, O.DROP_SELF_OUT)
        if getattr(layer, "has_cross_attention", False):
            layer.crossattention.self.dropout = _InjectedDropout(state, l, O.DROP_CROSS_P)
            layer.crossattention.output.dropout = _InjectedDropout(state, l, O.DROP_CROSS_OUT)
        layer.output.dropout = _InjectedDropout(state, l, O.DROP_FFN_T)
        layer.output_query.dropout = _InjectedDropout(state, l, O.DROP_FFN_Q)
    left = [n for n, m in bert.named_modules() if isinstance(m, torch.nn.Dropout)]
    assert not left, f"un-injected dropout modules: {left}"
    bert.register_forward_pre_hook(lambda m, a: state.__setitem__("pass", state["pass"] + 1))
    return state


DROPOUT_SEED, DROPOUT_P = 7, 0.1                    # model.dropout_seed of the dropout golden; BERT's hidden / attention dropout (blip2.py:48)


def effective_drop_seed(dropout_seed: int, step: int = 1) -> int:
    """the 64-bit seed sprc_amd.model gives the masks of training step `step` (model.py: _TrainFn.forward)"""
    return (dropout_seed * 0x9E3779B1 + step) & 0xFFFFFFFFFFFFFFFF


def train_goldens(out: Path, seed: int = 4, dropout: bool = False, autocast: bool = False):
    """Training forward + backward (N4): the REFERENCE's Blip2QformerCirAlignPrompt.forward (align_prompt.py:95-200) in eval mode on 5
    (reference, target, caption) triplets, depth-2 ViT-g + the full Q-Former: the three losses, and the gradient of
    loss_itc + 0.4 loss_rtc + 0.4 loss_align (blip_fine_tune_2.py:293-304) with respect to every trainable tensor, as
    `grad_functionals`; tensors the reference leaves without a gradient (itm_head, the LM head) are listed."""
    cfg = get_config("pretrain", vit_depth=2)
    sd = synth.make_state_dict(cfg, seed=seed)
    # autocast=True: forward + backward under torch.autocast("cpu", float16) with the loss scaled by GradScaler's initial 2^16 and the
    # gradients unscaled afterwards -- the reference's training ARITHMETIC (blip_fine_tune_2.py:290-303: `with torch.cuda.amp.autocast():`
    # ... `scaler.scale(loss).backward()`), CPU autocast standing in for CUDA autocast (op lists and accumulation orders differ: a proxy).
    # dropout=True: the reference AS IT TRAINS (blip_fine_tune_2.py:290 `blip_model.train()`: Q-Former dropout p = 0.1 active, the ViT
    # pinned to eval by align_prompt.py:67-68) with the masks injected (`inject_dropout_masks`); False: eval mode
    model = ref_import.build_reference_model(cfg, sd, eval_mode=not dropout)
    B = 5
    images = synth.make_images(2 * B, seed=seed)
    ids, mask, _ = synth.make_queries(B, B, seed=seed + 1)
    state = None
    if dropout:
        assert model.training and model.Qformer.training and not model.visual_encoder.training
        state = inject_dropout_masks(model, effective_drop_seed(DROPOUT_SEED), DROPOUT_P)
    import contextlib
    amp = (lambda: torch.autocast("cpu", dtype=torch.float16)) if autocast else contextlib.nullcontext
    model.tokenizer.set_next(ids, mask)
    with torch.no_grad(), amp():
        out_d = model({"image": images[:B], "target": images[B:], "text_input": ["caption"] * B})
    if state is not None:                                   # the second evaluation below must draw the same masks
        n_sites = len(state["calls"])
        assert state["pass"] == 3 and n_sites == len(set(state["calls"]))
        state["pass"], state["calls"] = -1, []
    model.tokenizer.set_next(ids, mask)
    model.zero_grad()
    with amp():
        losses = model({"image": images[:B], "target": images[B:], "text_input": ["caption"] * B})
        total = sum(GRAD_WEIGHTS[k] * v for k, v in losses.items())
    if state is not None:
        assert all(abs(float(losses[k]) - float(out_d[k])) < 1e-6 for k in losses), "the injected masks are not reproducible"
    loss_scale = 65536.0 if autocast else 1.0
    while True:                                             # GradScaler's rule (torch/amp/grad_scaler.py): a step whose gradients hold inf / nan is
        model.zero_grad()                                   # skipped and the scale halved (backoff_factor 0.5) -- repeated until a step goes through
        (total * loss_scale).backward(retain_graph=autocast)
        finite = all(bool(torch.isfinite(p_.grad).all()) for p_ in model.parameters() if p_.grad is not None)
        if finite or not autocast:
            break
        print(f"  loss scale {loss_scale:g}: non-finite gradients, halving (GradScaler backoff)")
        loss_scale *= 0.5
    if autocast:
        for p_ in model.parameters():
            if p_.grad is not None:
                p_.grad.div_(loss_scale)
    grads, no_grad, frozen = {}, [], []
    for name, p_ in model.named_parameters():
        if not p_.requires_grad:
            frozen.append(name)
        elif p_.grad is None:
            no_grad.append(name)
        elif name in sd or name == "temp":
            grads[name] = grad_functionals(name, p_.grad)
        else:
            no_grad.append(name + " (not in the synthetic state dict)")
    assert all(n.startswith("visual_encoder.") for n in frozen)
    np.savez_compressed(out, model_type="pretrain", vit_depth=2, seed=seed, batch=B, image_probe=_np(images[:, :, 0, :4]),
                        input_ids=ids.numpy(), attention_mask=mask.numpy(),
                        grad_names=np.array(list(grads)), grad_values=np.stack(list(grads.values())),
                        no_grad_names=np.array(no_grad), grad_weights=json.dumps(GRAD_WEIGHTS),
                        dropout_p=np.float64(DROPOUT_P if dropout else 0.0), dropout_seed=np.int64(DROPOUT_SEED if dropout else 0),
                        drop_seed_effective=np.uint64(effective_drop_seed(DROPOUT_SEED) if dropout else 0),
                        dropout_sites=np.int64(n_sites if dropout else 0), autocast_fp16=np.int64(int(autocast)), loss_scale=np.float64(loss_scale),
                        **{k: np.float64(v.item()) for k, v in out_d.items()})
    print(f"wrote {out}:", {k: round(v.item(), 6) for k, v in out_d.items()}, f"{len(grads)} gradient tensors, no grad: {no_grad[:6]} ...")


def caption_goldens(out: Path):
    import importlib
    import types
    ref_import.install_shims()
    # blip_processors imports torchvision/omegaconf/randaugment at module level; only the pure-string
    # BlipCaptionProcessor is exercised.
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tf = types.ModuleType("torchvision.transforms.functional")
    tf.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    tr.functional = tf
    tr.Normalize = lambda *a, **k: None
    tv.transforms = tr
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": tf})
    sys.modules["lavis.processors.randaugment"] = types.SimpleNamespace(RandomAugment=object)
    base = importlib.import_module("lavis.processors.base_processor")
    sys.modules["lavis.processors"].BaseProcessor = base.BaseProcessor        # registry.py:124
    bp = importlib.import_module("lavis.processors.blip_processors")
    proc = bp.BlipCaptionProcessor()
    raw = [
        "Make the dog bigger.", "  Remove the \"second\" person (left)!  ", "A*B#C:D;E~F", "trailing newline\n",
        "multiple     spaces   here", "UPPER Case And Punctuation!!!", "", "x", " ".join(f"w{i}" for i in range(70)),
        "is shorter.?, ", "has a v-neck; and it's (more) colourful", "Shows two dogs instead of one: both brown.",
    ]
    table = [[c, proc(c)] for c in raw]
    pairs = [["is shorter.", "has longer sleeves"], ["Is Red?", "  and blue,"], ["more colorful.", "less formal. "]]
    fiq = []
    for c1, c2 in pairs:   # validate_blip.py:180-184 composition, executed here as the reference writes it
        flattened = [c1, c2]
        composed = f"{flattened[0].strip('.?, ').capitalize()} and {flattened[1].strip('.?, ')}"
        fiq.append([c1, c2, composed, proc(composed)])
    out.write_text(json.dumps({"pre_caption": table, "fiq": fiq}, indent=1))
    print("wrote", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also generate the full-depth ViT-g golden (slow)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    only = set(a.only.split(",")) if a.only else None

    def want(k):
        return only is None or k in only

    if want("captions"):
        caption_goldens(GOLD / "captions.json")
    if want("metrics"):
        metrics_goldens(GOLD / "metrics.json")
    if want("tiny_eva"):
        model_goldens("pretrain", 2, n_img=4, n_q=6, out=GOLD / "tiny_eva.npz")
    if want("tiny_clip"):
        model_goldens("pretrain_vitL", 2, n_img=3, n_q=4, out=GOLD / "tiny_clip.npz")
    if want("train"):
        train_goldens(GOLD / "train_eva.npz")
    if want("train_dropout"):                   # the reference as it trains: Q-Former dropout p = 0.1 on, reproducible injected masks
        train_goldens(GOLD / "train_dropout_eva.npz", dropout=True)
    if only is not None and "train_autocast" in only:   # (explicit only: ~10 min) the reference's training ARITHMETIC: fp16 autocast + loss scaling
        train_goldens(GOLD / "train_autocast_eva.npz", autocast=True)
    if want("rerank"):
        rerank_goldens(GOLD / "rerank_eva.npz")
    if want("planted"):
        planted_goldens(GOLD / "planted_eva.npz")
    if a.full and want("planted_full"):      # full depth (39 blocks), 96 gallery images x 48 queries: the dtype-parity case
        planted_goldens(GOLD / "planted_full_eva.npz", n_img=96, n_q=48, vit_depth=None)
    if a.full and want("planted_clip"):      # the same on config C5's backbone (CLIP ViT-L, 24 blocks): the fp8 path's structured case
        planted_goldens(GOLD / "planted_full_clip.npz", n_img=96, n_q=48, vit_depth=None, model_type="pretrain_vitL")
    # second draws of weights, images and queries (seed 1) of both full-depth cases: the 16-bit engines' max|dsim| is an extreme-value
    # statistic over 4608 scores -- one draw says little about how much room there is under 1e-3
    if a.full and want("planted_full_s1"):
        planted_goldens(GOLD / "planted_full_eva_s1.npz", n_img=96, n_q=48, vit_depth=None, seed=1)
    # a larger draw (256 gallery images x 128 queries = 32768 scores, seed 2): the tail of the 16-bit engines' error distribution
    if a.full and want("planted_big"):
        planted_goldens(GOLD / "planted_big_eva.npz", n_img=256, n_q=128, vit_depth=None, seed=2)
    # a second 256 x 128 ViT-g draw (round 4, seed 3): the tail of the error distributions behind the "engine vs the reference's GPU
    # arithmetic" statement (tests/test_fp16_gpu.py) rests on more than one large draw
    if a.full and want("planted_big_s3"):
        planted_goldens(GOLD / "planted_big_eva_s3.npz", n_img=256, n_q=128, vit_depth=None, seed=3)
    if a.full and want("planted_clip_s1"):
        planted_goldens(GOLD / "planted_full_clip_s1.npz", n_img=96, n_q=48, vit_depth=None, model_type="pretrain_vitL", seed=1)
    # the same cases on a checkpoint whose trunk Conv / Linear tensors hold fp16 VALUES -- what a GPU-trained reference checkpoint
    # (the released SPRC weights included) contains; the reference's CPU path computes on them in fp32
    if a.full and want("planted_big_h16"):
        planted_goldens(GOLD / "planted_big_eva_h16.npz", n_img=256, n_q=128, vit_depth=None, seed=2, trunk_fp16=True)
    if a.full and want("planted_full_h16"):
        planted_goldens(GOLD / "planted_full_eva_h16.npz", n_img=96, n_q=48, vit_depth=None, trunk_fp16=True)
    if a.full and want("planted_clip_h16"):
        planted_goldens(GOLD / "planted_full_clip_h16.npz", n_img=96, n_q=48, vit_depth=None, model_type="pretrain_vitL", trunk_fp16=True)
    if a.full and want("full_eva"):
        model_goldens("pretrain", None, n_img=2, n_q=3, out=GOLD / "full_eva.npz")
    if a.full and want("full_clip"):
        model_goldens("pretrain_vitL", None, n_img=2, n_q=3, out=GOLD / "full_clip.npz")


if __name__ == "__main__":
    main()
