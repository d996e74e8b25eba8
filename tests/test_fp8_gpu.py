"""fp8 (OCP e4m3fn) path -- BASELINE.json config C5 "ViT-L backbone, fp8 MFMA": GEMM with fp8 operands on the MX-scaled
v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales; per-tensor activation scale and per-output-channel weight scales in the
epilogue, fp32 accumulation), LayerNorm with an fp8 operand copy, the calibration pass, and
the fp8 ViT inside the full pipeline against the REFERENCE goldens (bound measured and stated here)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sprc_amd import _lib as L  # noqa: E402
from sprc_amd import engine as E  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

DEV = "cuda:0"

import sys as _sys, os as _os  # noqa: E402
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _cases as CASES  # noqa: E402  (session-lived full-depth state dicts: tests/_cases.py)
F8 = torch.float8_e4m3fn


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _quant(x, scale):
    return (x / scale).clamp(-448, 448).to(F8)


@pytest.mark.parametrize("M,N,K", [(2300, 4100, 1408), (300, 384, 128), (32896 // 8 + 100, 1024, 4096), (514, 1408, 6144), (129, 130, 256)])
def test_gemm_fp8_operands(M, N, K):
    """exact up to fp32 accumulation: the reference multiplies the SAME quantised operands in fp64 (every tile path:
    128x128, 256x256 anti-phase, peeled remainder, split-K remainder)"""
    A, W, b, r = _rand((M, K), 1), _rand((N, K), 2, 0.05), _rand((N,), 3), _rand((M, N), 4)
    a_scale = float(A.abs().max()) / 448.0
    Wq, ws = E.quantize_fp8_rows(W)
    Aq = _quant(A, a_scale)
    ref = (Aq.double() * a_scale) @ (Wq.double() * ws.double()[:, None]).t() + b.double()
    scratch = torch.empty(8 * 128 * N, dtype=torch.float32, device=DEV)
    kw = dict(bias=b.to(DEV), w_scale=ws.to(DEV), a_scale=a_scale, scratch=scratch)
    out = E.gemm(Aq.to(DEV), Wq.to(DEV), out_dtype=L.SPRC_F32, resid=r.to(DEV), **kw).cpu()
    torch.testing.assert_close(out.double(), ref + r.double(), atol=2e-3 * math.sqrt(K / 64), rtol=1e-4)
    for _ in range(3):       # repeated launches are bit-identical (a race in the staged pipeline shows up as run-to-run differences)
        again = E.gemm(Aq.to(DEV), Wq.to(DEV), out_dtype=L.SPRC_F32, resid=r.to(DEV), **kw).cpu()
        assert torch.equal(again, out)
    out16 = E.gemm(Aq.to(DEV), Wq.to(DEV), out_dtype=L.SPRC_BF16, **kw).cpu()
    torch.testing.assert_close(out16.float(), ref.float().to(torch.bfloat16).float(), atol=3e-2, rtol=1e-2)
    # fp8 output with GELU: equals the fp8 rounding of the fp32 result (one e4m3 ulp where the fp32 sums differ in the last bits)
    g = torch.nn.functional.gelu(ref)
    o_scale = 448.0 / float(g.abs().max())
    out8 = E.gemm(Aq.to(DEV), Wq.to(DEV), out_dtype=L.SPRC_FP8, act=L.ACT_GELU, out_scale=o_scale, **kw).cpu()
    assert out8.dtype == F8
    want8 = (g * o_scale).float().clamp(-448, 448).to(F8)
    diff = (out8.float() - want8.float()).abs()
    # one e4m3 step; near zero two subnormal steps (2^-9 each): the bf16/fp8 GELU is the fast form (|err| <= 2.6e-5, times o_scale)
    ulp = (want8.float().abs() * 2.0 ** -3).clamp_min(2.0 ** -8)
    assert float((diff > ulp).float().mean()) == 0.0 and float((diff > 0).float().mean()) < 0.02
    # quantisation error of the whole product against the UNQUANTISED operands: the e4m3 noise level, ~2^-4 / sqrt(K) per output
    full = A.double() @ W.double().t() + b.double()
    rel = float((out.double() - r.double() - full).norm() / full.norm())
    print(f"\n[fp8 gemm {M}x{N}x{K}] relative error vs unquantised operands: {rel:.3e}")
    assert rel < 0.06


def test_gemm_fp8_rejects_bad_arguments():
    A, W = torch.zeros((64, 128), dtype=F8, device=DEV), torch.zeros((64, 128), dtype=F8, device=DEV)
    ws = torch.ones(64, device=DEV)
    with pytest.raises(L.SprcError):
        E.gemm(A, W, out_dtype=L.SPRC_F32)                                   # no scales
    with pytest.raises(L.SprcError):
        E.gemm(A, W, out_dtype=L.SPRC_FP8, w_scale=ws, a_scale=1.0)          # fp8 output without out_scale
    with pytest.raises(L.SprcError):
        E.gemm(A[:, :64].contiguous(), W[:, :64].contiguous(), out_dtype=L.SPRC_F32, w_scale=ws, a_scale=1.0)   # K % 128
    with pytest.raises(L.SprcError):
        E.gemm(A.view(torch.uint8).to(torch.bfloat16), W.view(torch.uint8).to(torch.bfloat16), out_dtype=L.SPRC_FP8, out_scale=1.0)


def test_layernorm_fp8_copy_and_absmax():
    import ctypes as C
    lib = L.load()
    M, D, eps = 77, 1024, 1e-5
    x, g, b = _rand((M, D), 30, 3.0) + 0.5, _rand((D,), 31) * 0.1 + 1, _rand((D,), 32) * 0.1
    ref = torch.nn.functional.layer_norm(x, (D,), g, b, eps)
    s = float(ref.abs().max()) / 448.0
    y8 = torch.empty((M, D), dtype=F8, device=DEV)
    a = L.LayerNormArgs()
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    a.M, a.D, a.out_dtype = M, D, L.SPRC_FP8
    a.x, a.ldx, a.gamma, a.beta, a.eps = xd.data_ptr(), D, gd.data_ptr(), bd.data_ptr(), eps
    a.y16, a.ld16, a.y16_scale = y8.data_ptr(), D, 1.0 / s
    L.check(lib.sprc_layernorm(C.byref(a), torch.cuda.current_stream().cuda_stream))
    want = (ref / s).clamp(-448, 448).to(F8)
    d = (y8.cpu().float() - want.float()).abs()
    assert float((d > (want.float().abs() * 2.0 ** -3).clamp_min(2.0 ** -9)).float().mean()) == 0.0 and float((d > 0).float().mean()) < 0.01
    # absmax over a bf16 tensor accumulates into the caller's slot
    t = (_rand((1000, 333), 40, 5.0)).to(torch.bfloat16).to(DEV)
    am = torch.tensor([1.5], device=DEV)
    L.check(lib.sprc_absmax_bf16(t.data_ptr(), t.numel(), am.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert float(am) == max(1.5, float(t.float().abs().max()))


@pytest.mark.parametrize("name", ["tiny_clip.npz", "full_clip.npz", "tiny_eva.npz", "full_eva.npz"])
def test_fp8_vit_in_the_pipeline(golden_dir, name):
    """fp8 ViT (qkv / fc1 / fc2 on e4m3 operands; attention, proj, Q-Former bf16; residual / LN / softmax fp32) against the
    reference goldens with RANDOM weights (scores all within 0.14 +- 0.01: the easy case -- the structured one is
    test_fp8_on_the_planted_full_depth_cases below).  Scales are calibrated on the golden's own images (static per-tensor activation scales).  Measured
    on MI355X (round 3): max |dsim| 5.6e-4 .. 1.9e-3 over the four goldens (bf16: 8e-4 .. 1e-3) -- e4m3 has 3 mantissa bits; the bound
    asserted is the largest measured value + 50 %."""
    g = np.load(golden_dir / name, allow_pickle=False)
    cfg = get_config(str(g["model_type"]), vit_depth=int(g["vit_depth"]))
    sd = CASES.state_dict(cfg, int(g["seed"]))
    images = synth.make_images(int(g["n_img"]), seed=int(g["seed"])).to(DEV)
    ref_eng = E.Engine(cfg, sd, DEV, dtype="bf16", max_batch=8)
    amax = ref_eng.calibrate_fp8(images)
    assert amax.shape == (cfg.vit.depth, 3) and bool((amax > 0).all())
    raw16 = ref_eng.vit_forward(images)
    del ref_eng
    eng = E.Engine(cfg, sd, DEV, dtype="fp8", max_batch=8, fp8_amax=amax, fp8_base="bf16")
    raw = eng.vit_forward(images)
    feats, _ = eng.qformer_image(raw)
    ref = torch.from_numpy(g["ref_index"]).to(DEV)
    fusion, _ = eng.qformer_fuse(raw[ref], torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    sim = E.sim_max(fusion, feats).cpu().numpy()
    dsim = float(np.abs(sim - g["sim"]).max())
    cos = float((feats.cpu().numpy() * g["feats"]).sum(-1).min())
    rows = g["rows"].tolist()
    print(f"\n[{name} fp8] max|dsim|={dsim:.2e} min cos(feats)={cos:.6f} max|draw| vs reference={np.abs(raw.cpu().numpy()[:, rows] - g['raw']).max():.2e} "
          f"(bf16 engine: {np.abs(raw16.cpu().numpy()[:, rows] - g['raw']).max():.2e})")
    assert torch.isfinite(raw).all() and cos > 0.99 and dsim < 3e-3
    with pytest.raises(ValueError):
        E.Engine(cfg, sd, DEV, dtype="fp8")                                  # no calibration data


# ---- the structured case: what e4m3 operands do to scores that are spread over 1.0 ------------------------------------------------------
def _planted(golden_dir, name):
    return CASES.planted_case(golden_dir, name)


def _scores(eng, g, images, bs=32):
    raw = torch.cat([eng.vit_forward(images[s:s + bs].to(DEV)) for s in range(0, images.shape[0], bs)])
    feats, _ = eng.qformer_image(raw)
    ref = torch.from_numpy(g["ref_index"]).to(DEV)
    fusion, _ = eng.qformer_fuse(raw[ref], torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"]))
    return E.sim_max(fusion, feats)


@pytest.mark.parametrize("name,base,layers", [("planted_full_clip.npz", "bf16", "all"), ("planted_full_clip.npz", "fp16", "all"),
                                              ("planted_full_clip.npz", "fp16", "mlp"), ("planted_full_eva.npz", "bf16", "all")])
def test_fp8_on_the_planted_full_depth_cases(golden_dir, name, base, layers):
    """Config C5's backbone (CLIP ViT-L, 23 blocks; and ViT-g, 39 blocks) at FULL depth on the planted-structure case generated from the
    reference (oracle/gen_golden.py: planted_goldens; 96 gallery images x 48 queries, scores spread over 1.0).  e4m3 operands carry 3
    mantissa bits: every fp8 product adds ~5 % relative noise to its output, whatever the scales, and on scores that are NOT all alike
    this shows: the north star's 1e-3 is out of reach for this dtype (the 16-bit engines hold it on the same files, checked below), so what
    is asserted is what was measured on MI355X + 50 %, and what the noise does to the ORDER -- Recall@K against the reference's own
    metric code, and how far the targets move in the ranking -- is reported and bounded too.  C5 is a throughput configuration."""
    from sprc_amd import harness as H
    g, cfg, sd, images = _planted(golden_dir, name)
    ref_eng = E.Engine(cfg, sd, DEV, dtype=base, max_batch=32)
    amax = torch.stack([ref_eng.calibrate_fp8(images[s:s + 32].to(DEV)) for s in range(0, images.shape[0], 32)]).amax(0)
    s16 = _scores(ref_eng, g, images).cpu().numpy()
    del ref_eng
    eng = E.Engine(cfg, sd, DEV, dtype="fp8", max_batch=32, fp8_amax=amax, fp8_base=base, fp8_layers=layers)
    assert eng.vit.fp8 == (L.FP8_ALL if layers == "all" else L.FP8_MLP) and bool(eng.x3) == (base == "fp16")
    sim = _scores(eng, g, images)
    s = sim.cpu().numpy()
    d = s - g["sim"]
    err, rms, err16 = float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())), float(np.abs(s16 - g["sim"]).max())
    cirr = np.array(H.cirr_metrics_from_sim(sim, g["ref_index"], g["tgt_index"], g["groups"]))
    fiq = np.array(H.fiq_metrics_from_sim(sim, g["tgt_index"]))
    # rank of every query's target (reference image removed), ours vs the reference's
    def target_ranks(scores):
        sc = scores.copy()
        sc[np.arange(sc.shape[0]), g["ref_index"]] = -np.inf
        return (sc > sc[np.arange(sc.shape[0]), g["tgt_index"]][:, None]).sum(1)
    shift = np.abs(target_ranks(s) - target_ranks(g["sim"].copy()))
    top1 = float((s.argmax(1) == g["sim"].argmax(1)).mean())
    print(f"\n[{name} fp8 over {base}, {layers}] max|dsim|={err:.2e} rms={rms:.2e} ({base} engine alone: {err16:.2e}); top-1 equal to the reference's for "
          f"{100 * top1:.1f} % of the queries; target rank shift mean {shift.mean():.2f} max {int(shift.max())}; CIRR metrics {np.round(cirr, 2).tolist()} vs reference "
          f"{np.round(g['cirr'], 2).tolist()}; FashionIQ {np.round(fiq, 2).tolist()} vs {np.round(g['fiq'], 2).tolist()}")
    assert err16 < (1e-3 if base == "fp16" else 2e-2)             # the 16-bit base on the same file
    assert np.isfinite(s).all() and err < 0.11 and rms < 0.035      # measured 6e-2 .. 7.3e-2 / 1.3e-2 .. 2.2e-2
    assert shift.mean() < 4.0 and np.abs(cirr - g["cirr"]).max() <= 15.0 and np.abs(fiq - g["fiq"]).max() <= 15.0
