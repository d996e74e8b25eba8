"""GPU parity of every building-block operator of libsprc_hip.so (called through the C ABI).

References are plain fp32/fp64 torch expressions evaluated on the CPU on the same seeded inputs;
inputs are asymmetric random (never symmetric/identity) so operand or output transposes show.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sprc_amd import _lib as L  # noqa: E402
from sprc_amd import engine as E  # noqa: E402

DEV = "cuda:0"


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(257 * 2, 1408, 1408), (100, 96, 64), (128, 128, 128), (300, 4224, 1408),
                                   (514, 1408, 6144), (64, 256, 768), (1, 128, 64), (129, 130, 192)])
def test_gemm_bf16(M, N, K):
    A, W, b = _bf(_rand((M, K), 1)), _bf(_rand((N, K), 2, 0.05)), _rand((N,), 3)
    ref = A.float().double() @ W.float().double().t() + b.double()
    out = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_F32).cpu()
    torch.testing.assert_close(out.double(), ref, atol=2e-3 * math.sqrt(K / 64), rtol=1e-4)
    out16 = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_BF16).cpu()
    torch.testing.assert_close(out16.float(), ref.float().to(torch.bfloat16).float(), atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("M,N,K", [(2300, 4100, 64), (2300, 4100, 128), (2300, 4100, 192), (2300, 4100, 704),
                                   (4224, 4096, 256), (4224 + 77, 4096, 320), (2300, 4100, 384), (2300, 4100, 448)])
def test_gemm_bf16_256_tile_and_peel(M, N, K):
    """Shapes the dispatcher sends to the 256x256 anti-phase kernel (ragged M and N; K = 1 .. 7 and 11 K-tiles: prologue, early
    pieces, every steady-loop / run-time-tail split of the lean K loop, counted-vmcnt tails) and to the peeled split (first M & ~255 rows on 256x256, remainder on 128x128)."""
    A, W, b, r = _bf(_rand((M, K), 21)), _bf(_rand((N, K), 22, 0.05)), _rand((N,), 23), _rand((M, N), 24)
    ref = A.double() @ W.double().t() + b.double()
    out = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_F32).cpu()
    torch.testing.assert_close(out.double(), ref, atol=2e-3 * math.sqrt(K / 64), rtol=1e-4)
    rd = r.to(DEV)
    E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), resid=rd, out_dtype=L.SPRC_F32, out=rd)      # in-place residual stream
    torch.testing.assert_close(rd.cpu().double(), ref + r.double(), atol=2e-3 * math.sqrt(K / 64), rtol=1e-4)
    out16 = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_BF16, act=L.ACT_GELU).cpu()
    want = torch.nn.functional.gelu(ref).float().to(torch.bfloat16).float()
    torch.testing.assert_close(out16.float(), want, atol=2e-2, rtol=1e-2)
    # repeated launches are bit-identical (no race in the staged pipeline shows up as run-to-run differences)
    for _ in range(3):
        again = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_F32).cpu()
        assert torch.equal(again, out)


def test_gemm_bf16_random_shapes_through_the_dispatcher():
    """60 seeded random problems (ragged M and N, N not a multiple of 4 -> scalar epilogue, K = 1..16 K-tiles, every
    epilogue combination) so that each dispatcher branch -- 128x128, 256x256, peeled splits with 0..12 whole panels,
    short-K fp32+residual -- is hit with sizes nobody tuned for; float64 reference on the CPU."""
    rng = np.random.default_rng(2024)
    for case in range(60):
        big = case % 3 == 0
        M = int(rng.integers(2000, 9000)) if big else int(rng.integers(1, 1200))
        N = int(rng.choice([768, 1024, 1408, 2304, 3072])) if big else int(rng.integers(1, 700))
        if not big and case % 4 != 1:
            N = (N + 3) // 4 * 4
        K = 64 * int(rng.integers(1, 17))
        out32 = bool(rng.integers(0, 2))
        act = int(rng.integers(0, 3))
        use_res, use_bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        A, W = _bf(_rand((M, K), 100 + case)), _bf(_rand((N, K), 200 + case, 0.05))
        b = _rand((N,), 300 + case) if use_bias else None
        r = _rand((M, N), 400 + case) if use_res else None
        z = A.double() @ W.double().t()
        if b is not None:
            z = z + b.double()
        if act == L.ACT_GELU:
            z = torch.nn.functional.gelu(z)
        elif act == L.ACT_QUICKGELU:
            z = z * torch.sigmoid(1.702 * z)
        if r is not None:
            z = z + r.double()
        out = E.gemm(A.to(DEV), W.to(DEV), bias=None if b is None else b.to(DEV), resid=None if r is None else r.to(DEV),
                     out_dtype=L.SPRC_F32 if out32 else L.SPRC_BF16, act=act).cpu()
        what = f"case {case}: M={M} N={N} K={K} out32={out32} act={act} res={use_res} bias={use_bias}"
        if out32:
            torch.testing.assert_close(out.double(), z, atol=3e-3 * math.sqrt(K / 64), rtol=1e-4, msg=lambda m: f"{what}\n{m}")
        else:
            torch.testing.assert_close(out.double(), z, atol=4e-2, rtol=1.6e-2, msg=lambda m: f"{what}\n{m}")


@pytest.mark.parametrize("odt,act", [(L.SPRC_F32, L.ACT_NONE), (L.SPRC_BF16, L.ACT_GELU)])
def test_gemm_bf16_splitk_remainder(odt, act):
    """K >= 4096 with caller scratch: the remainder rows of the peeled split are reduced by 8 workgroups per tile (fixed
    summation order); same result contract as the plain launch, and bit-stable across launches."""
    M, N, K = 4096 + 100, 4096, 4096
    A, W, b, r = _bf(_rand((M, K), 31)), _bf(_rand((N, K), 32, 0.03)), _rand((N,), 33), _rand((M, N), 34)
    z = A.double() @ W.double().t() + b.double()
    if act == L.ACT_GELU:
        z = torch.nn.functional.gelu(z)
    ref = z + r.double()
    scratch = torch.empty(8 * 128 * N, dtype=torch.float32, device=DEV)
    out = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), resid=r.to(DEV), out_dtype=odt, act=act, scratch=scratch).cpu()
    plain = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), resid=r.to(DEV), out_dtype=odt, act=act).cpu()
    if odt == L.SPRC_F32:
        torch.testing.assert_close(out.double(), ref, atol=2e-3 * math.sqrt(K / 64), rtol=1e-4)
        torch.testing.assert_close(out, plain, atol=1e-4, rtol=1e-5)          # only the summation order differs
        assert torch.equal(out[:4096], plain[:4096])                            # main rows: same kernel, same bits
    else:
        torch.testing.assert_close(out.float(), ref.float().to(torch.bfloat16).float(), atol=3e-2, rtol=2e-2)
    again = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), resid=r.to(DEV), out_dtype=odt, act=act, scratch=scratch).cpu()
    assert torch.equal(again, out)


@pytest.mark.parametrize("M,N,K", [(257, 1408, 1408), (70, 96, 32), (130, 256, 768), (33, 128, 608)])
def test_gemm_f32(M, N, K):
    A, W, b = _rand((M, K), 4), _rand((N, K), 5, 0.05), _rand((N,), 6)
    ref = A.double() @ W.double().t() + b.double()
    out = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_F32).cpu()
    torch.testing.assert_close(out.double(), ref, atol=5e-5 * math.sqrt(K / 32), rtol=1e-5)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_GELU, L.ACT_QUICKGELU])
def test_gemm_epilogues(dtype, act):
    M, N, K = 200, 384, 256
    A, W, b, r = _rand((M, K), 7), _rand((N, K), 8, 0.1), _rand((N,), 9), _rand((M, N), 10)
    if dtype == "bf16":
        A, W = _bf(A), _bf(W)
    z = A.float().double() @ W.float().double().t() + b.double()
    if act == L.ACT_GELU:
        z = torch.nn.functional.gelu(z)
    elif act == L.ACT_QUICKGELU:
        z = z * torch.sigmoid(1.702 * z)
    ref = z + r.double()
    rd = r.to(DEV)
    out = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), resid=rd, out_dtype=L.SPRC_F32, act=act).cpu()
    torch.testing.assert_close(out.double(), ref, atol=3e-3 if dtype == "bf16" else 1e-4, rtol=1e-4)
    # in-place residual (C aliases resid), as the ViT/Q-Former residual stream uses it
    E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), resid=rd, out_dtype=L.SPRC_F32, act=act, out=rd)
    torch.testing.assert_close(rd.cpu().double(), ref, atol=3e-3 if dtype == "bf16" else 1e-4, rtol=1e-4)


def test_gemm_rowmaps():
    # A rows = "rows [32:64] of every 64-row group"; C rows = "rows [:32] of every 64-row group"
    Bn, K, N = 5, 128, 256
    A, W = _bf(_rand((Bn * 64, K), 11)), _bf(_rand((N, K), 12, 0.1))
    sel = torch.cat([torch.arange(g * 64 + 32, g * 64 + 64) for g in range(Bn)])
    dst = torch.cat([torch.arange(g * 64, g * 64 + 32) for g in range(Bn)])
    ref = A[sel].float() @ W.float().t()
    out = torch.full((Bn * 64, N), 7.0, dtype=torch.float32, device=DEV)
    E.gemm(A.to(DEV), W.to(DEV), out_dtype=L.SPRC_F32, out=out, M=Bn * 32, amap=E.rowmap(32, 64, 32), cmap=E.rowmap(32, 64, 0))
    out = out.cpu()
    torch.testing.assert_close(out[dst], ref, atol=2e-3, rtol=1e-4)
    rest = torch.ones(Bn * 64, dtype=torch.bool)
    rest[dst] = False
    assert torch.all(out[rest] == 7.0)
    # single-row groups: "row 32 of every sample" (the text [CLS] row of align_prompt.py:349)
    out1 = E.gemm(A.to(DEV), W.to(DEV), out_dtype=L.SPRC_F32, M=Bn, amap=E.rowmap(1, 64, 32),
                  out=torch.empty((Bn, N), dtype=torch.float32, device=DEV)).cpu()
    torch.testing.assert_close(out1, A[32::64].float() @ W.float().t(), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("Bn,N,K,act,out32", [(233, 3072, 768, L.ACT_GELU, False), (233, 768, 3072, L.ACT_NONE, True),
                                              (5, 256, 128, L.ACT_NONE, True), (40, 768, 768, L.ACT_QUICKGELU, False)])
def test_gemm_pair_equals_two_launches(Bn, N, K, act, out32):
    """sprc_gemm_pair (the query / text FFN pair of a Q-Former layer): rows [:32] of every 64-row sample through W0, rows
    [32:] through W1, in one launch -- identical to the two separate row-mapped launches, and close to float64."""
    A = _bf(_rand((Bn * 64, K), 51)).to(DEV)
    W0, W1 = _bf(_rand((N, K), 52, 0.05)).to(DEV), _bf(_rand((N, K), 53, 0.05)).to(DEV)
    b0, b1 = _rand((N,), 54).to(DEV), _rand((N,), 55).to(DEV)
    r = _rand((Bn * 64, N), 56).to(DEV) if out32 else None
    odt = L.SPRC_F32 if out32 else L.SPRC_BF16
    qmap, tmap = E.rowmap(32, 64, 0), E.rowmap(32, 64, 32)
    mk = lambda: torch.full((Bn * 64, N), 7.0, dtype=torch.float32 if out32 else torch.bfloat16, device=DEV)
    one = E.gemm_pair(A, W0, W1, b0, b1, qmap, tmap, qmap, tmap, Bn * 32, mk(), resid=r, out_dtype=odt, act=act)
    two = mk()
    E.gemm(A, W0, bias=b0, resid=r, out_dtype=odt, act=act, out=two, M=Bn * 32, amap=qmap, cmap=qmap)
    E.gemm(A, W1, bias=b1, resid=r, out_dtype=odt, act=act, out=two, M=Bn * 32, amap=tmap, cmap=tmap)
    torch.testing.assert_close(one.float(), two.float(), atol=1e-2 if not out32 else 1e-4, rtol=1e-2 if not out32 else 1e-5)
    rows = torch.arange(Bn * 64)
    is_q = (rows % 64) < 32
    z = torch.where(is_q[:, None], A.cpu().double() @ W0.cpu().double().t() + b0.cpu().double(),
                    A.cpu().double() @ W1.cpu().double().t() + b1.cpu().double())
    if act == L.ACT_GELU:
        z = torch.nn.functional.gelu(z)
    elif act == L.ACT_QUICKGELU:
        z = z * torch.sigmoid(1.702 * z)
    if r is not None:
        z = z + r.cpu().double()
    tol = 3e-3 * math.sqrt(K / 64) if out32 else 4e-2
    torch.testing.assert_close(one.cpu().double(), z, atol=tol, rtol=1.6e-2 if not out32 else 1e-4)
    with pytest.raises(L.SprcError, match="differ"):
        E.gemm_pair(A, W0, W1[:, :K // 2].contiguous() if False else W1, b0, b1, qmap, E.rowmap(16, 64, 32), qmap, tmap, Bn * 32, mk(),
                    resid=r, out_dtype=odt, act=act)


def test_gemm_rejects_bad_arguments():
    A, W = _bf(_rand((8, 48), 1)).to(DEV), _bf(_rand((8, 48), 2)).to(DEV)
    with pytest.raises(L.SprcError, match="multiple"):
        E.gemm(A, W)                     # K = 48 not a multiple of 64


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("nq,N", [(5, 7), (64, 256), (33, 130)])
def test_sim_max(dtype, nq, N):
    f = torch.nn.functional.normalize(_rand((nq, 256), 20), dim=-1)
    G = torch.nn.functional.normalize(_rand((N, 32, 256), 21), dim=-1)
    if dtype == "bf16":
        f, G = _bf(f), _bf(G)
    ref = (f.float().double() @ G.float().double().reshape(N * 32, 256).t()).view(nq, N, 32).max(-1).values
    sim = E.sim_max(f.to(DEV).contiguous(), G.to(DEV).contiguous()).cpu()
    torch.testing.assert_close(sim.double(), ref, atol=2e-6 if dtype == "f32" else 1e-5, rtol=0)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,eps", [(1408, 1e-6), (1024, 1e-5), (768, 1e-12), (256, 1e-5)])
def test_layernorm(D, eps):
    M = 77
    x, g, b = _rand((M, D), 30, 3.0) + 0.5, _rand((D,), 31) * 0.1 + 1, _rand((D,), 32) * 0.1
    ref = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), eps)
    y32, y16 = E.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), eps, L.SPRC_BF16)
    torch.testing.assert_close(y32.cpu().double(), ref, atol=2e-6 * 3, rtol=1e-5)
    assert torch.equal(y16.cpu(), y32.cpu().to(torch.bfloat16))
    # row maps: normalise rows [32:64] of each 64-row group in place
    xm = _rand((3 * 64, D), 33).to(DEV)
    before = xm.clone()
    E.layernorm(xm, g.to(DEV), b.to(DEV), eps, L.SPRC_F32, want16=False, xmap=E.rowmap(32, 64, 32), ymap=E.rowmap(32, 64, 32),
                M=3 * 32, y32=xm)
    sel = torch.cat([torch.arange(gp * 64 + 32, gp * 64 + 64) for gp in range(3)])
    ref2 = before.cpu().clone()
    ref2[sel] = torch.nn.functional.layer_norm(before.cpu()[sel], (D,), g, b, eps)
    torch.testing.assert_close(xm.cpu(), ref2, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(257 * 2, 1408, 1408), (2300, 1408, 704), (4096 + 100, 1408, 6144), (233 * 32, 768, 3072), (70, 96, 64)])
def test_gemm_f16_delta_output(M, N, K):
    """SPRC_F16 output (a residual-branch delta): the fp32 result rounded once to fp16 -- every tile path (128x128, 256x256
    anti-phase, peeled remainder, split-K remainder with caller scratch)."""
    A, W, b = _bf(_rand((M, K), 61)), _bf(_rand((N, K), 62, 0.03)), _rand((N,), 63)
    ref = A.double() @ W.double().t() + b.double()
    scratch = torch.empty(8 * 128 * N, dtype=torch.float32, device=DEV)
    out = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_F16, scratch=scratch)
    assert out.dtype == torch.float16
    f32 = E.gemm(A.to(DEV), W.to(DEV), bias=b.to(DEV), out_dtype=L.SPRC_F32, scratch=scratch)
    assert torch.equal(out.cpu(), f32.cpu().to(torch.float16))                # same accumulation, one fp16 rounding
    torch.testing.assert_close(out.cpu().double(), ref, atol=2e-3 * math.sqrt(K / 64) + 2e-3, rtol=1e-3)
    with pytest.raises(L.SprcError):                                           # fp16 output carries no residual / activation
        E.gemm(A.to(DEV), W.to(DEV), resid=f32, out_dtype=L.SPRC_F16)
    with pytest.raises(L.SprcError):
        E.gemm(A.to(DEV), W.to(DEV), act=L.ACT_GELU, out_dtype=L.SPRC_F16)


@pytest.mark.parametrize("D,eps", [(1408, 1e-6), (768, 1e-12)])
def test_layernorm_fused_residual_add(D, eps):
    """LN(x + add16) with the sum written back over x (pre-LN residual update) and the bf16 result written over add16
    (the aliasing models.hip uses); row-mapped variant for the Q-Former's query/text row groups."""
    M = 131
    x, d = _rand((M, D), 70, 2.0), (_rand((M, D), 71) * 0.5).to(torch.float16)
    g, b = _rand((D,), 72) * 0.1 + 1, _rand((D,), 73) * 0.1
    s_ref = x + d.float()                                                       # one fp32 add per element: exact reference
    y_ref = torch.nn.functional.layer_norm(s_ref.double(), (D,), g.double(), b.double(), eps)
    xd, dd = x.to(DEV), d.to(DEV)
    buf16 = dd.clone().view(torch.bfloat16)                                     # y16 aliases add16 (2-byte elements, same rows)
    y32, y16 = E.layernorm(xd, g.to(DEV), b.to(DEV), eps, L.SPRC_BF16, y16=buf16, add16=buf16.view(torch.float16), sum32=xd)
    assert torch.equal(xd.cpu(), s_ref)                                         # x <- x + add16, bit-exact
    torch.testing.assert_close(y32.cpu().double(), y_ref, atol=1e-5, rtol=1e-5)
    assert torch.equal(buf16.cpu(), y32.cpu().to(torch.bfloat16))
    # without sum32: x untouched (post-LN BERT form: a = LN(dense + x))
    xd2 = x.to(DEV)
    y32b, _ = E.layernorm(xd2, g.to(DEV), b.to(DEV), eps, L.SPRC_BF16, add16=dd)
    assert torch.equal(xd2.cpu(), x) and torch.equal(y32b.cpu(), y32.cpu())
    # row maps: rows [32:64] of every 64-row group
    xm, dm = _rand((3 * 64, D), 74).to(DEV), (_rand((3 * 64, D), 75) * 0.3).to(torch.float16).to(DEV)
    mp = E.rowmap(32, 64, 32)
    out32 = torch.zeros((3 * 64, D), device=DEV)
    E.layernorm(xm, g.to(DEV), b.to(DEV), eps, L.SPRC_F32, want16=False, xmap=mp, ymap=mp, M=3 * 32, y32=out32, add16=dm)
    sel = torch.cat([torch.arange(gp * 64 + 32, gp * 64 + 64) for gp in range(3)])
    want = torch.nn.functional.layer_norm((xm.cpu() + dm.cpu().float())[sel], (D,), g, b, eps)
    torch.testing.assert_close(out32.cpu()[sel], want, atol=1e-5, rtol=1e-5)
    rest = torch.ones(3 * 64, dtype=torch.bool); rest[sel] = False
    assert torch.all(out32.cpu()[rest] == 0)


def _attn_ref(q, k, v, scale, mask=None):
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask[:, None, None, :].double()
    return (torch.softmax(s, dim=-1) @ v.double())


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("B,H,Tq,Tk,dh,masked", [(2, 16, 257, 257, 88, False), (3, 12, 64, 64, 64, True),
                                                 (2, 12, 32, 257, 64, False), (2, 16, 257, 257, 64, False),
                                                 (1, 12, 32, 32, 64, False), (2, 3, 40, 70, 88, True)])
def test_attention(dtype, B, H, Tq, Tk, dh, masked):
    D = H * dh
    q, k, v = _rand((B, Tq, D), 40), _rand((B, Tk, D), 41), _rand((B, Tk, D), 42)
    if dtype == "bf16":
        q, k, v = _bf(q), _bf(k), _bf(v)
    mask = None
    if masked:
        keep = torch.ones(B, Tk)
        for b in range(B):
            keep[b, Tk - 3 - 5 * b:] = 0
        mask = (1.0 - keep) * -10000.0
    scale = dh ** -0.5
    ref = _attn_ref(q.float().view(B, Tq, H, dh).transpose(1, 2), k.float().view(B, Tk, H, dh).transpose(1, 2),
                    v.float().view(B, Tk, H, dh).transpose(1, 2), scale, mask).transpose(1, 2).reshape(B * Tq, D)
    out = E.attention(q.to(DEV).view(B * Tq, D), k.to(DEV).view(B * Tk, D), v.to(DEV).view(B * Tk, D), B, H, Tq, Tk, dh,
                      D, D, D, scale, key_mask=None if mask is None else mask.to(DEV)).cpu()
    tol = 2e-2 if dtype == "bf16" else 2e-5
    torch.testing.assert_close(out.float().double(), ref, atol=tol, rtol=tol)


def test_attention_random_shapes():
    """30 seeded random attention problems per dtype: ragged Tq / Tk (tails of the 32-row tiles, a single query or key),
    every head size the kernels take, random padding masks (each batch keeps at least one key)."""
    rng = np.random.default_rng(77)
    for case in range(30):
        B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        Tq, Tk = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        dh = 8 * int(rng.integers(1, 13))
        masked = bool(rng.integers(0, 2))
        D = H * dh
        for dtype in ("bf16", "f32"):
            if dtype == "f32" and dh % 4:
                continue
            q, k, v = _rand((B, Tq, D), 500 + case), _rand((B, Tk, D), 600 + case), _rand((B, Tk, D), 700 + case)
            if dtype == "bf16":
                q, k, v = _bf(q), _bf(k), _bf(v)
            mask = None
            if masked:
                keep = (torch.from_numpy(rng.random((B, Tk))) > 0.3).float()
                keep[:, int(rng.integers(0, Tk))] = 1.0
                mask = (1.0 - keep) * -10000.0
            scale = dh ** -0.5
            ref = _attn_ref(q.float().view(B, Tq, H, dh).transpose(1, 2), k.float().view(B, Tk, H, dh).transpose(1, 2),
                            v.float().view(B, Tk, H, dh).transpose(1, 2), scale, mask).transpose(1, 2).reshape(B * Tq, D)
            out = E.attention(q.to(DEV).view(B * Tq, D), k.to(DEV).view(B * Tk, D), v.to(DEV).view(B * Tk, D), B, H, Tq, Tk,
                              dh, D, D, D, scale, key_mask=None if mask is None else mask.to(DEV)).cpu()
            tol = 2e-2 if dtype == "bf16" else 2e-5
            what = f"case {case} {dtype}: B={B} H={H} Tq={Tq} Tk={Tk} dh={dh} masked={masked}"
            torch.testing.assert_close(out.float().double(), ref, atol=tol, rtol=tol, msg=lambda m: f"{what}\n{m}")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("dh", [64, 72, 88, 96])
def test_attention_streaming_key_tails(dtype, dh):
    """The streaming (DMA ring + transposing LDS reads) kernel of the ViT blocks on every kind of last key tile: none (Tk % 32 == 0),
    the short path of <= 4 real keys (257 tokens: one), the masked generic one; head dims with and without the ones column (88) and
    without padding (64, 96); query counts that leave waves of a workgroup without rows."""
    B, H = 2, 3
    D = H * dh
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    for Tq, Tk in ((257, 257), (130, 256), (257, 258), (200, 260), (161, 261), (257, 272), (193, 287), (129, 33), (257, 36)):
        q, k, v = (_rand((B, t, D), 900 + i + Tk).to(tdt) for i, t in enumerate((Tq, Tk, Tk)))
        scale = dh ** -0.5
        ref = _attn_ref(q.float().view(B, Tq, H, dh).transpose(1, 2), k.float().view(B, Tk, H, dh).transpose(1, 2),
                        v.float().view(B, Tk, H, dh).transpose(1, 2), scale).transpose(1, 2).reshape(B * Tq, D)
        out = E.attention(q.to(DEV).view(B * Tq, D), k.to(DEV).view(B * Tk, D), v.to(DEV).view(B * Tk, D), B, H, Tq, Tk, dh, D, D, D, scale).cpu()
        assert out.dtype == tdt
        tol = 2e-2 if dtype == "bf16" else 3e-3
        torch.testing.assert_close(out.float().double(), ref, atol=tol, rtol=tol, msg=lambda m: f"Tq={Tq} Tk={Tk} dh={dh} {dtype}\n{m}")


def test_attention_packed_qkv_layout():
    # q/k/v interleaved as [token][3][H][dh] exactly like the ViT qkv GEMM output (eva_vit.py:125)
    B, H, T, dh = 2, 16, 257, 88
    D = H * dh
    qkv = _bf(_rand((B * T, 3 * D), 43))
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = _attn_ref(q.float().view(B, T, H, dh).transpose(1, 2), k.float().view(B, T, H, dh).transpose(1, 2),
                    v.float().view(B, T, H, dh).transpose(1, 2), dh ** -0.5).transpose(1, 2).reshape(B * T, D)
    d = qkv.to(DEV)
    out = E.attention(d[:, :D], d[:, D:2 * D], d[:, 2 * D:], B, H, T, T, dh, 3 * D, 3 * D, 3 * D, dh ** -0.5).cpu()
    torch.testing.assert_close(out.float().double(), ref, atol=2e-2, rtol=2e-2)


# ------------------------------------------------------------------------------------------------
def test_im2row_assemble_embed_l2norm_mask():
    import ctypes as C
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    B, S, P, kpad = 3, 224, 14, 640
    img = _rand((B, 3, S, S), 50)
    rows = torch.empty((B * 256, kpad), dtype=torch.float32, device=DEV)
    img_d = img.to(DEV)                     # keep device inputs alive: the C ABI only sees raw pointers
    L.check(lib.sprc_im2row(img_d.data_ptr(), rows.data_ptr(), B, S, P, kpad, L.SPRC_F32, st))
    ref = torch.nn.functional.unfold(img, kernel_size=P, stride=P).transpose(1, 2).reshape(B * 256, 588)
    assert torch.equal(rows.cpu()[:, :588], ref) and torch.all(rows.cpu()[:, 588:] == 0)
    rows16 = torch.empty((B * 256, kpad), dtype=torch.bfloat16, device=DEV)
    L.check(lib.sprc_im2row(img_d.data_ptr(), rows16.data_ptr(), B, S, P, kpad, L.SPRC_BF16, st))
    assert torch.equal(rows16.cpu()[:, :588], ref.to(torch.bfloat16))

    D, T = 1408, 257
    po, cls, pos = _rand((B * 256, D), 51), _rand((D,), 52), _rand((T, D), 53)
    x = torch.empty((B, T, D), dtype=torch.float32, device=DEV)
    po_d, cls_d, pos_d = po.to(DEV), cls.to(DEV), pos.to(DEV)
    L.check(lib.sprc_vit_assemble(po_d.data_ptr(), cls_d.data_ptr(), pos_d.data_ptr(), x.data_ptr(), B, T, D, st))
    refx = torch.cat([cls.expand(B, 1, D), po.view(B, 256, D)], dim=1) + pos
    assert torch.equal(x.cpu(), refx)

    Hd, Lq, Lt, V = 768, 32, 32, 500
    qe, we, pe = _rand((B, Lq, Hd), 54), _rand((V, Hd), 55), _rand((512, Hd), 56)
    g, bt = _rand((Hd,), 57) * 0.1 + 1, _rand((Hd,), 58) * 0.1
    ids = torch.randint(0, V, (B, Lt), generator=torch.Generator().manual_seed(59))
    y32 = torch.empty((B, Lq + Lt, Hd), dtype=torch.float32, device=DEV)
    y16 = torch.empty((B, Lq + Lt, Hd), dtype=torch.bfloat16, device=DEV)
    a = L.QformerEmbedArgs()
    dv = [t.to(DEV) for t in (qe, ids, we, pe, g, bt)]
    a.B, a.Lq, a.Lt, a.hidden, a.out_dtype, a.vocab = B, Lq, Lt, Hd, L.SPRC_BF16, V
    a.query_embeds, a.q_bstride, a.input_ids, a.word_emb, a.pos_emb = dv[0].data_ptr(), Lq * Hd, dv[1].data_ptr(), dv[2].data_ptr(), dv[3].data_ptr()
    a.gamma, a.beta, a.eps, a.y32, a.y16 = dv[4].data_ptr(), dv[5].data_ptr(), 1e-12, y32.data_ptr(), y16.data_ptr()
    L.check(lib.sprc_qformer_embed(C.byref(a), st))
    emb = torch.cat([qe, we[ids] + pe[:Lt]], dim=1)
    refe = torch.nn.functional.layer_norm(emb, (Hd,), g, bt, 1e-12)
    torch.testing.assert_close(y32.cpu(), refe, atol=1e-5, rtol=1e-5)
    assert torch.equal(y16.cpu(), y32.cpu().to(torch.bfloat16))
    # broadcast query tokens, no text (image-only call shape)
    a.Lt, a.q_bstride, a.input_ids = 0, 0, None
    L.check(lib.sprc_qformer_embed(C.byref(a), st))
    refq = torch.nn.functional.layer_norm(qe[0], (Hd,), g, bt, 1e-12)
    torch.testing.assert_close(y32.cpu().view(-1, Hd)[:B * Lq].view(B, Lq, Hd), refq.expand(B, -1, -1), atol=1e-5, rtol=1e-5)

    xr = _rand((70, 256), 60, 3.0)
    xr[5] = 0                                   # zero row: F.normalize's eps clamp
    o32 = torch.empty_like(xr, device=DEV)
    o16 = torch.empty((70, 256), dtype=torch.bfloat16, device=DEV)
    xr_d = xr.to(DEV)
    L.check(lib.sprc_l2norm_rows(xr_d.data_ptr(), 256, o32.data_ptr(), o16.data_ptr(), 256, 70, 256, L.SPRC_BF16, st))
    torch.testing.assert_close(o32.cpu(), torch.nn.functional.normalize(xr, dim=-1), atol=1e-6, rtol=1e-6)
    assert torch.equal(o16.cpu(), o32.cpu().to(torch.bfloat16))

    m = (torch.rand((B, Lt), generator=torch.Generator().manual_seed(61)) > 0.4).long()
    om = torch.empty((B, Lq + Lt), dtype=torch.float32, device=DEV)
    m_d = m.to(DEV)
    L.check(lib.sprc_qformer_mask(m_d.data_ptr(), om.data_ptr(), B, Lq, Lt, st))
    refm = (1.0 - torch.cat([torch.ones(B, Lq), m.float()], dim=1)) * -10000.0
    assert torch.equal(om.cpu(), refm)


def test_cast_matches_torch_rne():
    lib = L.load()
    x = torch.cat([_rand((100003,), 70, 5.0), torch.tensor([0.0, -0.0, 1.0, 65504.0, 1e-40, float("inf"), -float("inf")])])
    d = x.to(DEV)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=DEV)
    L.check(lib.sprc_cast_f32_to_bf16(d.data_ptr(), out.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out.cpu().view(torch.int16), x.to(torch.bfloat16).view(torch.int16))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nq,N,k,ties", [(7, 50, 10, False), (33, 2297, 51, False), (9, 64, 64, True), (5, 40, 64, True),
                                         (16, 1000, 50, True), (3, 1, 5, False), (4, 20000, 64, True)])
def test_topk_and_rank_of_are_bit_exact(nq, N, k, ties):
    from oracle import sprc_oracle as O
    rng = np.random.default_rng(nq * 1000 + N)
    sim = rng.uniform(-0.3, 1.0, (nq, N)).astype(np.float32)
    if ties:
        sim = (np.round(sim * 32) / 32).astype(np.float32)
    want_sim, want_idx = O.topk_stable(sim, min(k, N))
    d = torch.from_numpy(sim).to(DEV)
    vals, idx = E.topk(d, k)
    vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
    kk = min(k, N)
    np.testing.assert_array_equal(idx[:, :kk], want_idx.astype(np.int32))
    np.testing.assert_array_equal(vals[:, :kk], want_sim)
    assert np.all(idx[:, kk:] == -1)
    listed = rng.integers(-1, N, (nq, 8)).astype(np.int32)
    r = E.rank_of(d, torch.from_numpy(listed)).cpu().numpy()
    np.testing.assert_array_equal(r, O.rank_of(sim, listed))
    # global-index variant (what the sharded merge uses): keys tie-break on gidx, not on position
    perm = np.stack([rng.permutation(N) for _ in range(nq)]).astype(np.int32)
    _, idx2 = E.topk(torch.from_numpy(np.take_along_axis(sim, perm, axis=1)).to(DEV), k, gidx=torch.from_numpy(perm).to(DEV))
    np.testing.assert_array_equal(idx2.cpu().numpy()[:, :kk], want_idx.astype(np.int32))


def test_topk_rank_random_shapes():
    """40 seeded random ranking problems: any k <= 64, N from 1 to 30 000, tie densities from none to heavy, strided
    score rows (a column slice of a wider matrix), with and without explicit global indices -- integer-exact vs the oracle."""
    from oracle import sprc_oracle as O
    rng = np.random.default_rng(4242)
    for case in range(40):
        nq, N, k = int(rng.integers(1, 70)), int(rng.choice([1, 2, 31, 64, 65, 500, 2297, 6346, 30000])), int(rng.integers(1, 65))
        levels = int(rng.choice([0, 4, 64, 4096]))
        sim = rng.uniform(-1.0, 1.0, (nq, N)).astype(np.float32)
        if levels:
            sim = (np.round(sim * levels) / levels).astype(np.float32)
        pad = int(rng.integers(0, 9))
        wide = torch.zeros((nq, N + pad), dtype=torch.float32, device=DEV)
        wide[:, :N] = torch.from_numpy(sim).to(DEV)
        d = wide[:, :N]                                      # row stride N + pad
        kk = min(k, N)
        want_v, want_i = O.topk_stable(sim, kk)
        v, i = E.topk(d, k)
        np.testing.assert_array_equal(i.cpu().numpy()[:, :kk], want_i.astype(np.int32), err_msg=f"case {case} nq={nq} N={N} k={k}")
        np.testing.assert_array_equal(v.cpu().numpy()[:, :kk], want_v)
        listed = rng.integers(-1, N, (nq, 5)).astype(np.int32)
        np.testing.assert_array_equal(E.rank_of(d, torch.from_numpy(listed)).cpu().numpy(), O.rank_of(sim, listed))
        base = int(rng.integers(0, 1 << 20))
        _, ib = E.topk(d, k, idx_base=base)
        np.testing.assert_array_equal(ib.cpu().numpy()[:, :kk], want_i.astype(np.int32) + base)


def test_profiler_start_pause_resume_and_busy_time():
    """sprc_prof_enable(1 | 0 | 2) = start afresh / pause / resume; collect sums per class: launches, algorithmic flops,
    the sum of the launch durations and the union of their intervals (equal on one stream, up to event resolution)."""
    lib = L.load()
    A, W = _bf(_rand((512, 256), 1)).to(DEV), _bf(_rand((384, 256), 2)).to(DEV)
    x, g, b = _rand((300, 768), 3).to(DEV), torch.ones(768, device=DEV), torch.zeros(768, device=DEV)

    def collect():
        torch.cuda.synchronize()
        prof = (L.ProfEntry * len(L.K_CLASSES))()
        L.check(lib.sprc_prof_collect(prof), "sprc_prof_collect")
        return {n: prof[i] for i, n in enumerate(L.K_CLASSES)}

    lib.sprc_prof_enable(1)
    E.gemm(A, W)
    E.gemm(A, W)
    lib.sprc_prof_enable(0)
    E.gemm(A, W)                                         # not recorded
    E.layernorm(x, g, b, 1e-5, L.SPRC_BF16)              # not recorded
    lib.sprc_prof_enable(2)
    E.gemm(A, W)
    E.layernorm(x, g, b, 1e-5, L.SPRC_BF16)
    lib.sprc_prof_enable(0)
    p = collect()
    assert p["gemm_bf16"].launches == 3 and p["rowops"].launches == 1 and p["attention"].launches == 0
    assert p["gemm_bf16"].flops == 3 * 2.0 * 512 * 384 * 256
    assert 0.0 < p["gemm_bf16"].busy_ms <= p["gemm_bf16"].ms * 1.001 + 1e-3
    assert p["gemm_bf16"].ms < 50.0                      # three tiny launches: the paused section is not in the sum
    lib.sprc_prof_enable(1)                              # start afresh drops the old records
    lib.sprc_prof_enable(0)
    assert collect()["gemm_bf16"].launches == 0


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_attention_two_key_segments(dtype):
    """Second key/value segment + per-sample row indices (the stage-2 rerank's cross-attention over cat(reference,
    candidate) tokens): equals attention over the explicitly concatenated keys."""
    B, H, Tq, T1, T2, dh, NA, NB = 5, 12, 32, 257, 257, 64, 3, 4
    D = H * dh
    q = _rand((B, Tq, D), 90)
    ka, va = _rand((NA, T1, D), 91), _rand((NA, T1, D), 92)
    kb, vb = _rand((NB, T2, D), 93), _rand((NB, T2, D), 94)
    ia, ib = torch.tensor([2, 0, 1, 2, 0], dtype=torch.int32), torch.tensor([3, 3, 0, 1, 2], dtype=torch.int32)
    if dtype == "bf16":
        q, ka, va, kb, vb = (_bf(t) for t in (q, ka, va, kb, vb))
    k = torch.cat([ka[ia.long()], kb[ib.long()]], dim=1).float()
    v = torch.cat([va[ia.long()], vb[ib.long()]], dim=1).float()
    ref = _attn_ref(q.float().view(B, Tq, H, dh).transpose(1, 2), k.view(B, T1 + T2, H, dh).transpose(1, 2),
                    v.view(B, T1 + T2, H, dh).transpose(1, 2), dh ** -0.5).transpose(1, 2).reshape(B * Tq, D)
    out = E.attention(q.to(DEV).view(B * Tq, D), ka.to(DEV).view(NA * T1, D), va.to(DEV).view(NA * T1, D), B, H, Tq, T1, dh, D, D, D,
                      dh ** -0.5, k2=kb.to(DEV).view(NB * T2, D), v2=vb.to(DEV).view(NB * T2, D), Tk2=T2, ld2=D,
                      kv_index=ia.to(DEV), kv2_index=ib.to(DEV)).cpu()
    tol = 2e-2 if dtype == "bf16" else 2e-5
    torch.testing.assert_close(out.float().double(), ref, atol=tol, rtol=tol)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_attention_one_query_tile_over_a_long_key_axis(dtype):
    """The Q-Former's cross-attention shape (Qformer.py:175-281: 32 query tokens x 257 encoder tokens, 12 heads x 64, no key mask) runs on
    ONE-WAVE workgroups of the streaming DMA kernel (round 6): every kind of last key tile, fewer than 32 queries, head dims below 64, K|V
    read out of wider token rows (the model's [B * 257, 9216] layout) -- and, fp16, the split-precision output rows (SPRC_F16X3) written
    straight from the registers: their hi segment equals the plain output's bits, lo / e4m3 segments decode to the same values."""
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    for B, H, Tq, Tk, dh in ((2, 12, 32, 257, 64), (3, 5, 32, 128, 64), (2, 3, 17, 129, 40), (1, 4, 1, 300, 64), (2, 2, 32, 260, 8), (2, 12, 32, 288, 64)):
        D = H * dh
        ld = 3 * D + 16                                      # K at column 0, V at column D + 8 of a wider row
        q = _rand((B * Tq, D), 1200 + Tk).to(tdt).to(DEV)
        kv = _rand((B * Tk, ld), 1300 + Tk).to(tdt).to(DEV)
        k, v = kv[:, :D], kv[:, D + 8:2 * D + 8]
        scale = dh ** -0.5
        ref = _attn_ref(q.cpu().float().view(B, Tq, H, dh).transpose(1, 2), k.cpu().float().reshape(B, Tk, H, dh).transpose(1, 2),
                        v.cpu().float().reshape(B, Tk, H, dh).transpose(1, 2), scale).transpose(1, 2).reshape(B * Tq, D)
        out = E.attention(q, k, v, B, H, Tq, Tk, dh, D, ld, ld, scale)
        tol = 2e-2 if dtype == "bf16" else 3e-3
        torch.testing.assert_close(out.cpu().float().double(), ref, atol=tol, rtol=tol, msg=lambda m: f"B={B} H={H} Tq={Tq} Tk={Tk} dh={dh} {dtype}\n{m}")
        if dtype == "fp16" and D % 4 == 0:
            rows = E.attention(q, k, v, B, H, Tq, Tk, dh, D, ld, ld, scale, out_x3=True)
            hi, lo8, hi8 = E.split_decode(rows, D)
            torch.testing.assert_close(hi.float(), out.float(), atol=0, rtol=2.0 ** -10)     # (one ulp: the two epilogues may contract o * inv differently)
            assert float((hi != out).float().mean()) < 1e-2, (Tq, Tk, dh)
            # lo: the fp16 rounding residual of the fp32 value (<= 2^-11 relative, stored to 3 mantissa bits); hi8: the value itself in e4m3
            assert float(lo8.abs().max()) <= float(out.float().abs().max()) * 2.0 ** -10
            torch.testing.assert_close(hi8, out.float(), atol=2.0 ** -9, rtol=2.0 ** -3)


def test_cu_partition_streams():
    """sprc_stream_create_partition (ABI 6): CU-masked streams over disjoint shares of every XCD.  A product launched on a partition sizes
    its persistent grid by the partition's CU count and gives the bits of the same product on the whole chip; two partitions run side by
    side (fork / join on a NON-blocking stream: the masked streams are blocking ones, and any null-stream operation serialises them)."""
    import ctypes as C
    lib = L.load()
    main = torch.cuda.Stream(device=DEV)
    full = lib.sprc_stream_cus(None)
    assert full >= 64 and full % 16 == 0
    handles = []
    for i in range(2):
        h = C.c_void_p()
        L.check(lib.sprc_stream_create_partition(i, 2, C.byref(h)), "sprc_stream_create_partition")
        assert lib.sprc_stream_cus(h) == full // 2
        handles.append(h)
    assert lib.sprc_stream_create_partition(2, 2, C.byref(C.c_void_p())) != 0          # part out of range
    A, W = _rand((4096 + 40, 1408), 31).to(torch.float16).to(DEV), _rand((1408 + 128, 1408), 32).to(torch.float16).to(DEV)
    with torch.cuda.stream(main):
        want = E.gemm(A, W)
        outs = []
        for h in handles:
            st = torch.cuda.ExternalStream(h.value, device=DEV)
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(E.gemm(A, W))
            main.wait_stream(st)
    torch.cuda.synchronize()
    for o in outs:                                          # (the tile choice follows the CU count: same reduction order per element, not asserted bit for bit)
        torch.testing.assert_close(o.float(), want.float(), atol=2e-2, rtol=2e-3)
    for h in handles:
        L.check(lib.sprc_stream_destroy(h), "sprc_stream_destroy")
        assert lib.sprc_stream_cus(h) == full                                            # no longer registered
