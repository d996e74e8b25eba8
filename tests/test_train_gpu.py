"""Training forward: the three losses of Blip2QformerCirAlignPrompt.forward on the HIP engine
against the numbers the unmodified REFERENCE produced (tests/golden/train_eva.npz, eval mode) and against the oracle on a
ViT-L case; the building-block loss kernels against plain torch expressions."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sprc_oracle as O  # noqa: E402
from sprc_amd import _lib as L  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402
from sprc_amd.model import Blip2QformerCirAlignPrompt  # noqa: E402
from sprc_amd.tokenizer import TokenBatch  # noqa: E402

DEV = "cuda:0"


class _Tok:
    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        return TokenBatch(self.ids, self.mask)


@pytest.mark.parametrize("dtype,tol", [("fp32", 5e-5), ("bf16", 2e-2)])
def test_forward_losses_match_reference(golden_dir, dtype, tol):
    g = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    B = int(g["batch"])
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype=dtype, max_batch=8)
    assert not model.load_state_dict(sd, strict=False).missing_keys
    model = model.to(DEV)
    model.tokenizer = _Tok(torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    images = synth.make_images(2 * B, seed=int(g["seed"]))
    with torch.no_grad():
        out = model({"image": images[:B].to(DEV), "target": images[B:].to(DEV), "text_input": ["caption"] * B})
    torch.cuda.synchronize()
    got = {k: float(v) for k, v in out.items()}
    print(f"\n[train forward {dtype}]", {k: round(v, 6) for k, v in got.items()}, "reference:", {k: round(float(g[k]), 6) for k in got})
    assert set(got) == {"loss_itc", "loss_rtc", "loss_align"}
    for k in got:
        assert got[k] == pytest.approx(float(g[k]), abs=tol), k
    assert not any(v.requires_grad for v in out.values())                     # under no_grad: values only


def test_forward_losses_vitl_against_the_oracle():
    cfg = get_config("pretrain_vitL", vit_depth=1)
    sd = synth.make_state_dict(cfg, seed=8)
    B = 7
    images = synth.make_images(2 * B, seed=9)
    ids, mask, _ = synth.make_queries(B, B, seed=10)
    with torch.no_grad():
        want = O.training_losses(sd, cfg, images[:B], images[B:], ids, mask)
    from sprc_amd import engine as E
    eng = E.Engine(cfg, sd, DEV, dtype="fp32", max_batch=8)
    got = eng.training_losses(images[:B].to(DEV), images[B:].to(DEV), ids, mask, temp=float(sd["temp"]))
    for k in want:
        assert float(got[k]) == pytest.approx(float(want[k]), abs=5e-5), k


def test_loss_kernels():
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    for B in (1, 5, 64, 200):
        sim = torch.rand((B, B + 3), generator=g) * 2 - 1
        want = torch.nn.functional.cross_entropy(sim[:, :B].double() / 0.07, torch.arange(B))
        d, out = sim.to(DEV), torch.zeros(1, device=DEV)
        L.check(lib.sprc_contrastive_ce(d.data_ptr(), B + 3, B, 0.07, out.data_ptr(), st))
        assert float(out) == pytest.approx(float(want), rel=2e-6, abs=1e-6)
    h, prompt = torch.randn((6, 64, 768), generator=g), torch.randn((32, 768), generator=g)
    want = torch.nn.functional.mse_loss(h[:, :32].double().mean(1), prompt.double().mean(0).expand(6, -1))
    hd, pd, out = h.to(DEV), prompt.to(DEV), torch.zeros(1, device=DEV)
    L.check(lib.sprc_align_mse(hd.data_ptr(), 64 * 768, 32, 768, pd.data_ptr(), 6, out.data_ptr(), st))
    assert float(out) == pytest.approx(float(want), rel=1e-5)


# ---- backward (N4) -------------------------------------------------------------------------------------------------------
def _train_case(golden_dir, name="train_eva.npz"):
    """model in EVAL mode for the eval-mode golden; the dropout golden's model is in train mode with the golden's dropout seed"""
    g = np.load(golden_dir / name, allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g["seed"]))
    B = int(g["batch"])
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=8)
    assert not model.load_state_dict(sd, strict=False).missing_keys
    model = model.to(DEV)
    assert model.training                        # nn.Module's default, as for the reference class
    if "dropout_p" in g.files and float(g["dropout_p"]) > 0:
        model.train()
        model.dropout_seed = int(g["dropout_seed"])
        assert model.dropout_p == float(g["dropout_p"])
    else:
        model.eval()
    model.tokenizer = _Tok(torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    images = synth.make_images(2 * B, seed=int(g["seed"]))
    return g, cfg, sd, model, {"image": images[:B].to(DEV), "target": images[B:].to(DEV), "text_input": ["caption"] * B}


def test_backward_matches_the_reference_gradients(golden_dir):
    """`loss = loss_itc + 0.4 loss_rtc + 0.4 loss_align; loss.backward()` as blip_fine_tune_2.py:293-301 writes it, through
    model.forward's autograd.Function -> the HIP backward kernels: the gradient of EVERY trainable tensor (337: Q-Former incl. both
    embedding tables, ln_vision, the heads, query / prompt tokens, temp) against the reference's forward + backward (goldens: 12
    functionals per tensor, oracle/gen_golden.py), relative to the tensor's gradient norm."""
    import json
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_oracle_golden import check_gradients_against_golden
    g, cfg, sd, model, batch = _train_case(golden_dir)
    w = json.loads(str(g["grad_weights"]))
    losses = model(batch)
    assert all(v.requires_grad for v in losses.values())
    for k in losses:
        assert float(losses[k]) == pytest.approx(float(g[k]), abs=5e-5), k
    total = sum(w[k] * v for k, v in losses.items())
    total.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert not any(n.startswith("visual_encoder.") or n.startswith("itm_head.") for n in grads)      # frozen trunk / unused head
    worst = check_gradients_against_golden(g, grads, 3e-4)
    print(f"\n[train backward fp32] worst functional error / ||g|| = {worst:.2e} over {len(grads)} tensors "
          f"(two torch-CPU evaluations of the same graph differ by 9.5e-5 on vision_proj.bias, a cancellation)")
    assert worst < 3e-4


def test_backward_with_dropout_matches_the_reference_in_train_mode(golden_dir):
    """the reference TRAINS with `blip_model.train()` (blip_fine_tune_2.py:290): Q-Former dropout p = 0.1 on the
    embeddings, the attention probabilities and the two output.dense branches (Qformer.py:113,264,293,379).  `model.train()` now does
    that here: counter-based masks (sprc_dropout_f32 / sprc_attention.drop_*) regenerated in backward.  Golden: the unmodified reference
    in train mode with the SAME masks injected (oracle/gen_golden.py: inject_dropout_masks) -- losses and all 337 gradients."""
    import json
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_oracle_golden import check_gradients_against_golden
    g, cfg, sd, model, batch = _train_case(golden_dir, "train_dropout_eva.npz")
    w = json.loads(str(g["grad_weights"]))
    losses = model(batch)
    for k in losses:
        assert float(losses[k].detach()) == pytest.approx(float(g[k]), abs=5e-5), k
    sum(w[k] * v for k, v in losses.items()).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    worst = check_gradients_against_golden(g, grads, 3e-4)
    print(f"\n[train backward fp32, dropout p = 0.1] losses {[round(float(v), 5) for v in losses.values()]}; worst functional error / ||g|| = {worst:.2e} over {len(grads)} tensors")
    # a second step draws NEW masks (the step counter enters the seed); eval mode switches them off and gives the eval-mode golden's numbers
    l2 = model(batch)
    assert abs(float(l2["loss_itc"].detach()) - float(g["loss_itc"])) > 1e-4
    g0 = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    model.eval()
    l3 = model(batch)
    assert float(l3["loss_itc"].detach()) == pytest.approx(float(g0["loss_itc"]), abs=5e-5)
    # under torch.no_grad() (validation inside the training script, blip_fine_tune_2.py:322-330) there is no dropout either way
    model.train()
    with torch.no_grad():
        l4 = model(batch)
    assert float(l4["loss_itc"]) == pytest.approx(float(g0["loss_itc"]), abs=2e-4)


def test_dropout_kernel_matches_the_cpu_mask():
    """sprc_dropout_f32 == the oracle's drop_keep (the integer hash both sides compute), with and without the fused residual add"""
    lib = L.load()
    n, seed, site, p = 100003, 0x1234567890ABCDEF, 777, 0.1
    x, r = torch.randn(n), torch.randn(n)
    xd, rd, y = x.to(DEV), r.to(DEV), torch.empty(n, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    keep = torch.from_numpy(O.drop_keep(seed, site, n, p))
    L.check(lib.sprc_dropout_f32(xd.data_ptr(), None, y.data_ptr(), n, seed, site, p, st))
    assert torch.equal(y.cpu(), torch.where(keep, x * (1.0 / (1.0 - p)), torch.zeros(())).float()) or \
        torch.allclose(y.cpu(), x * keep.float() * (1.0 / (1.0 - p)), rtol=1e-6, atol=0)
    assert torch.equal(y.cpu() != 0, keep & (x != 0))
    L.check(lib.sprc_dropout_f32(xd.data_ptr(), rd.data_ptr(), y.data_ptr(), n, seed, site, p, st))
    torch.testing.assert_close(y.cpu(), x * keep.float() * (1.0 / (1.0 - p)) + r, rtol=1e-6, atol=1e-6)
    L.check(lib.sprc_dropout_f32(xd.data_ptr(), None, y.data_ptr(), n, seed, site, 0.0, st))
    assert torch.equal(y.cpu(), x)


def test_fp16_frozen_trunk_in_the_training_step(golden_dir):
    """train_vit_dtype="fp16": the frozen ViT of a training step on the fp16 engine, as the reference's loop runs it under autocast
    (blip_fine_tune_2.py:293); ln_vision, the Q-Former and the heads stay on the fp32 path.  Losses within 1e-3 of the fp32-trunk step
    (= of the reference's), every gradient within 2 % of its norm, same set of trained tensors."""
    g, cfg, sd, model, batch = _train_case(golden_dir)
    import json
    w = json.loads(str(g["grad_weights"]))

    def grads_of(m):
        losses = m(batch)
        sum(w[k] * v for k, v in losses.items()).backward()
        torch.cuda.synchronize()
        return {k: float(v) for k, v in losses.items()}, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    l32, g32 = grads_of(model)
    m16 = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=8, train_vit_dtype="fp16")
    assert not m16.load_state_dict(sd, strict=False).missing_keys
    m16 = m16.to(DEV).eval()
    m16.tokenizer = model.tokenizer
    l16, g16 = grads_of(m16)
    assert m16._train_engine().dt == L.SPRC_F16 and model._train_engine().dt == L.SPRC_F32
    assert set(g16) == set(g32) and len(g16) > 300
    worst = max(float((g16[n] - g32[n]).norm() / g32[n].norm().clamp_min(1e-12)) for n in g32 if float(g32[n].norm()) > 1e-8)
    print(f"\n[train step, fp16 frozen trunk] losses {l16} vs fp32 trunk {l32}; worst relative gradient difference {worst:.2e} over {len(g16)} tensors")
    assert all(abs(l16[k] - l32[k]) < 1e-3 for k in l32) and worst < 2e-2
    with pytest.raises(ValueError):
        Blip2QformerCirAlignPrompt(cfg=cfg, train_vit_dtype="bf16")


def test_grad_scaler_and_adamw_step_as_in_the_reference_loop(golden_dir):
    """The reference's update (blip_fine_tune_2.py:257-262, :293-304): AdamW + GradScaler; a few steps on one batch must lower the loss,
    the inference engine must see the moved weights, and the ViT trunk must stay untouched."""
    g, cfg, sd, model, batch = _train_case(golden_dir)
    opt = torch.optim.AdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 2e-5, "betas": (0.9, 0.98), "eps": 1e-7,
                              "weight_decay": 0.05}])
    scaler = torch.cuda.amp.GradScaler()
    trunk0 = model.state_dict()["visual_encoder.blocks.0.attn.qkv.weight"].clone()
    hist = []
    for _ in range(4):
        opt.zero_grad()
        with torch.cuda.amp.autocast():
            d = model(batch)
            loss = d["loss_itc"] + 0.4 * d["loss_rtc"] + 0.4 * d["loss_align"]
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        hist.append(float(loss))
    print(f"\n[train loop] loss over 4 AdamW steps: {[round(x, 4) for x in hist]}")
    assert hist[-1] < hist[0] - 1e-3
    assert torch.equal(model.state_dict()["visual_encoder.blocks.0.attn.qkv.weight"], trunk0)
    with torch.no_grad():
        after = model(batch)
    assert float(after["loss_itc"]) < float(g["loss_itc"])                      # the inference engine was rebuilt from the new weights


def test_backward_kernels_against_torch_autograd():
    """Each backward kernel alone against torch.autograd on the CPU (float64)."""
    import ctypes as C
    from sprc_amd import train as T
    k = T._K(torch.device(DEV))
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator().manual_seed(11)
    r = lambda *s_: torch.randn(s_, generator=gen)                              # noqa: E731
    # transpose / colsum
    x = r(70, 133)
    assert torch.equal(k.transpose(x.to(DEV), pad=32)[:, :70].cpu(), x.t())
    out = torch.ones(133, device=DEV)
    k.colsum(x.to(DEV), out)
    torch.testing.assert_close(out.cpu().double(), 1 + x.double().sum(0), atol=1e-5, rtol=0)
    # gelu
    z, dy = r(1000) * 3, r(1000)
    zz = z.double().requires_grad_(True)
    torch.nn.functional.gelu(zz).backward(dy.double())
    torch.testing.assert_close(k.gelu(z.to(DEV)).cpu().double(), torch.nn.functional.gelu(z.double()), atol=1e-6, rtol=0)
    torch.testing.assert_close(k.gelu_bwd(z.to(DEV), dy.to(DEV)).cpu().double(), zz.grad, atol=1e-6, rtol=0)
    # layernorm
    for M, D in ((37, 768), (9, 1408)):
        x, gam, bet, dy = r(M, D) * 2 + 0.3, r(D) * 0.1 + 1, r(D) * 0.1, r(M, D)
        xx, gg, bb = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
        torch.nn.functional.layer_norm(xx, (D,), gg, bb, 1e-12).backward(dy.double())
        dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        dx = k.ln_bwd(x.to(DEV), gam.to(DEV), dy.to(DEV), 1e-12, dg, db)
        torch.testing.assert_close(dx.cpu().double(), xx.grad, atol=2e-5, rtol=0)
        torch.testing.assert_close(dg.cpu().double(), gg.grad, atol=2e-5, rtol=0)
        torch.testing.assert_close(db.cpu().double(), bb.grad, atol=2e-5, rtol=0)
    # attention (self with a padding mask; cross over 257 keys)
    for B, Tq, Tk, masked in ((3, 64, 64, True), (2, 32, 257, False)):
        H = 12
        q, kk, v, do = r(B * Tq, H * 64) * 0.5, r(B * Tk, H * 64) * 0.5, r(B * Tk, H * 64), r(B * Tq, H * 64)
        mask = None
        if masked:
            mask = (1.0 - (torch.arange(Tk)[None, :] < torch.tensor([Tk, Tk - 9, 40])[:B, None]).float()) * -10000.0
        qq, k2, vv = (t.double().requires_grad_(True) for t in (q, kk, v))
        s_ = torch.einsum("bqhd,bkhd->bhqk", qq.view(B, Tq, H, 64), k2.view(B, Tk, H, 64)) / 8.0
        if mask is not None:
            s_ = s_ + mask.double()[:, None, None, :]
        o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s_, -1), vv.view(B, Tk, H, 64)).reshape(B * Tq, H * 64)
        o.backward(do.double())
        md = None if mask is None else mask.to(DEV)
        got_o = k.attention(q.to(DEV), kk.to(DEV), v.to(DEV), B, H, Tq, Tk, md, 0.125)
        torch.testing.assert_close(got_o.cpu().double(), o.detach(), atol=2e-5, rtol=0)
        dq, dk, dv = k.attention_bwd(q.to(DEV), kk.to(DEV), v.to(DEV), do.to(DEV), B, H, Tq, Tk, md, 0.125)
        for got, want in ((dq, qq.grad), (dk, k2.grad), (dv, vv.grad)):
            torch.testing.assert_close(got.cpu().double(), want, atol=3e-5, rtol=0)
    # similarity (max over 32 tokens) + cross entropy + temp, l2norm
    B, J, E_ = 6, 32, 256
    fu, fe = torch.nn.functional.normalize(r(B, E_), dim=-1), torch.nn.functional.normalize(r(B, J, E_), dim=-1)
    f1, f2, tt = fu.double().requires_grad_(True), fe.double().requires_grad_(True), torch.tensor(0.07, dtype=torch.float64, requires_grad=True)
    sim = torch.einsum("be,nje->bnj", f1, f2).max(-1).values
    (2.5 * torch.nn.functional.cross_entropy(sim / tt, torch.arange(B))).backward()
    simd = torch.einsum("be,nje->bnj", fu, fe).max(-1).values.to(DEV).contiguous()
    dsim, dtemp = torch.empty((B, B), device=DEV), torch.zeros(1, device=DEV)
    L.check(lib.sprc_contrastive_ce_bwd(simd.data_ptr(), B, B, 0.07, 2.5, dsim.data_ptr(), dtemp.data_ptr(), st))
    dfu, dfe, js = torch.zeros((B, E_), device=DEV), torch.zeros((B, J, E_), device=DEV), torch.empty((B, B), dtype=torch.int32, device=DEV)
    fud, fed = fu.to(DEV), fe.to(DEV)
    L.check(lib.sprc_sim_max_bwd(fud.data_ptr(), fed.data_ptr(), dsim.data_ptr(), B, B, J, E_, dfu.data_ptr(), dfe.data_ptr(), js.data_ptr(), st))
    torch.testing.assert_close(dfu.cpu().double(), f1.grad, atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(dfe.cpu().double(), f2.grad, atol=1e-5, rtol=1e-4)
    assert float(dtemp) == pytest.approx(float(tt.grad), rel=1e-4)
    x, dy = r(9, 256), r(9, 256)
    xx = x.double().requires_grad_(True)
    torch.nn.functional.normalize(xx, dim=-1).backward(dy.double())
    dx = torch.empty((9, 256), device=DEV)
    xd, dyd = x.to(DEV), dy.to(DEV)
    L.check(lib.sprc_l2norm_bwd(xd.data_ptr(), 256, dyd.data_ptr(), 256, dx.data_ptr(), 256, 9, 256, st))
    torch.testing.assert_close(dx.cpu().double(), xx.grad, atol=1e-6, rtol=1e-5)
    # align mse
    h, prompt = r(4, 64, 768), r(32, 768)
    hh = h.double().requires_grad_(True)
    (0.4 * torch.nn.functional.mse_loss(hh[:, :32].mean(1), prompt.double().mean(0).expand(4, -1))).backward()
    dh = torch.zeros((4, 64, 768), device=DEV)
    hd, pd = h.to(DEV), prompt.to(DEV)
    L.check(lib.sprc_align_mse_bwd(hd.data_ptr(), 64 * 768, 32, 768, pd.data_ptr(), 4, 0.4, dh.data_ptr(), 64 * 768, st))
    torch.testing.assert_close(dh.cpu().double(), hh.grad, atol=1e-8, rtol=1e-4)


def test_fp16_products_training_step_against_the_references_autocast_gradients(golden_dir):
    """the training step in the REFERENCE'S ARITHMETIC.  blip_fine_tune_2.py:290-303 runs forward + backward under fp16
    autocast with a GradScaler; `train_products="fp16"` runs every product of the trainable part on fp16 operand copies (fp32 accumulation,
    fp32 outputs, fp32 master weights) and the frozen trunk on the fp16 engine.  Golden: the unmodified reference under
    torch.autocast("cpu", float16) with GradScaler's backoff rule (tests/golden/train_autocast_eva.npz: its scale settled at 2^13 -- 2^16 .. 2^14
    overflow the reference's fp16 gradients; CPU autocast stands in for CUDA autocast).  Two yardsticks, both as functional errors relative
    to each tensor's gradient norm: the reference's autocast gradients are themselves up to 7.1e-2 (median 1.1e-2) from its fp32 gradients;
    the engine, which rounds operands only, must sit CLOSER to the fp32 gradients than that, and within the autocast path's own distance of
    the autocast gradients."""
    import json
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_oracle_golden import _grad_functionals
    g32 = np.load(golden_dir / "train_eva.npz", allow_pickle=False)
    g16 = np.load(golden_dir / "train_autocast_eva.npz", allow_pickle=False)
    assert int(g16["autocast_fp16"]) == 1 and [str(n) for n in g16["grad_names"]] == [str(n) for n in g32["grad_names"]]
    cfg = get_config(str(g32["model_type"]), vit_depth=int(g32["vit_depth"]))
    sd = synth.make_state_dict(cfg, seed=int(g32["seed"]))
    B = int(g32["batch"])
    model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=8, train_vit_dtype="fp16", train_products="fp16")
    assert not model.load_state_dict(sd, strict=False).missing_keys
    model = model.to(DEV).eval()
    model.tokenizer = _Tok(torch.from_numpy(g32["input_ids"]), torch.from_numpy(g32["attention_mask"]))
    images = synth.make_images(2 * B, seed=int(g32["seed"]))
    batch = {"image": images[:B].to(DEV), "target": images[B:].to(DEV), "text_input": ["caption"] * B}
    w = json.loads(str(g32["grad_weights"]))
    scale = float(g16["loss_scale"])
    losses = model(batch)
    (sum(w[k] * v for k, v in losses.items()) * scale).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad / scale for n, p in model.named_parameters() if p.grad is not None}
    assert all(bool(torch.isfinite(v).all()) for v in grads.values()) and len(grads) == len(g32["grad_names"])
    for k in losses:
        assert float(losses[k]) == pytest.approx(float(g32[k]), abs=2e-3), k

    def worst_against(g):               # every error relative to the FP32 gradient norm; the key biases (mathematically zero gradients) are skipped
        errs = []
        for name, want, base in zip([str(n) for n in g["grad_names"]], g["grad_values"], g32["grad_values"]):
            if base[0] < 1e-8:
                continue
            got = _grad_functionals(name, grads[name])
            errs.append(float(np.abs(got[:3] - want[:3]).max() / base[0]))
        return max(errs), float(np.median(errs))
    w32, m32 = worst_against(g32)
    w16, m16 = worst_against(g16)
    ref = [float(np.abs(a[:3] - b[:3]).max() / a[0]) for a, b in zip(g32["grad_values"], g16["grad_values"]) if a[0] >= 1e-8]
    print(f"\n[train step, fp16 products] functional error / ||g||: vs the reference's fp32 gradients worst {w32:.2e} median {m32:.2e}; vs its autocast "
          f"gradients worst {w16:.2e} median {m16:.2e}; the reference's autocast vs its own fp32: worst {max(ref):.2e} median {float(np.median(ref)):.2e}")
    assert w32 < max(ref) and m32 < float(np.median(ref))
    assert w16 < 1.5 * max(ref)
    # ... and ABSOLUTE bounds on the distance from the fp32 gradients (the relative bar alone would let a stale operand copy or a
    # missing accumulation that costs a few percent of one tensor's norm through): measured on MI355X worst 2.0e-2, median 3.7e-3
    assert w32 < 3e-2 and m32 < 6e-3
