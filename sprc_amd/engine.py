"""Host-side driver of libsprc_hip.so: packs a reference-layout state dict into the C model
descriptors of include/sprc.h and issues one C call per batch on torch's current stream.

torch is used for device memory, streams and the one-time weight repack only; every
computation of the retrieval path runs in the HIP kernels.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .config import SprcConfig

_TORCH_DT = {L.SPRC_F32: torch.float32, L.SPRC_BF16: torch.bfloat16, L.SPRC_F16: torch.float16, L.SPRC_FP8: torch.float8_e4m3fn}
_SPRC_DT = {v: k for k, v in _TORCH_DT.items()}
# layer kinds of the split-precision Q-Former (include/sprc.h: SPRC_X3_*)
X3_QKV, X3_ATTN_OUT, X3_CROSS_Q, X3_CROSS_OUT, X3_FFN_IN, X3_FFN_OUT, X3_CKV, X3_HEADS, X3_ALL = 1, 2, 4, 8, 16, 32, 64, 128, 255
# Default split-precision masks of the fp16 engine: (image pass, query-side passes).  Round 4 (split product = fp16 main term + two
# e4m3 correction segments, 2 units of matrix time instead of round 3's 3): profiles/r04_gpuref_report_e8.txt, five full-depth planted
# goldens + three on fp16-valued checkpoints, rms of the score error / what the kinds cost per bench step:
#   254 : 174  (round 3's choice)        1.4e-4 .. 3.2e-4
#   254 : 190  (+ the query side's FFN input products, +0.6 ms)   1.2e-4 .. 2.8e-4: on every ViT-g case at or below the error of the
#              reference's own GPU arithmetic (tests/test_fp16_gpu.py) -- the default
#   254 : 254  (+ Q|K|V and the query side's K|V projection, +1.7 ms)   1.1e-4 .. 2.9e-4: nothing more to gain, what is left is the fp16 ViT
# The gallery features carry 4x the error variance of the fused queries at a fifth of the Q-Former's work, so the image pass splits
# everything but the self-attention Q|K|V product (the most expensive and least sensitive kind).
X3_DEFAULT = (X3_ALL & ~X3_QKV, X3_ATTN_OUT | X3_CROSS_Q | X3_CROSS_OUT | X3_FFN_IN | X3_FFN_OUT | X3_HEADS)
FP8_MAX = 448.0                                   # largest finite e4m3fn


def quantize_fp8_rows(w: torch.Tensor):
    """Per-output-channel e4m3fn quantisation of a weight [N, K]: W[n, :] = scale[n] * Wq[n, :]."""
    w = w.float()
    scale = (w.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX).contiguous()
    return (w / scale[:, None]).to(torch.float8_e4m3fn).contiguous(), scale


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def rowmap(rows_per_group: int = 0, group_stride: int = 0, group_offset: int = 0) -> L.RowMap:
    return L.RowMap(rows_per_group, group_stride, group_offset)


# --------------------------------------------------------------------------------------------
# thin operator wrappers (used by the unit parity tests and by Engine)
# ---- the split-precision row layout (sprc.h: SPRC_F16X3), restated with torch ops: what the producers' kernels write ---------------
SPLIT_LO_SHIFT, SPLIT_W_SHIFT, SPLIT_WLO_SHIFT = 12, 6, 18        # [x_lo 2^12 | x] . [W 2^6 | W_lo 2^18], product scaled by 2^-18
SPLIT_W_ABSMAX, SPLIT_X_ABSMAX = 3.5, 224.0                       # beyond these the e4m3 correction segments saturate (see Engine._split_rows)


def _e4m3_bytes(t: torch.Tensor) -> torch.Tensor:
    return t.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)         # RNE, saturating


def split_rows(x32: torch.Tensor, weight: bool = False) -> torch.Tensor:
    """[M, K] fp32 -> split rows as an fp16 VIEW [M, 2K] (4K bytes per row): [K fp16: hi | K e4m3: (x - hi) 2^12 | K e4m3: x]
    (activations) or [K fp16: W_hi | K e4m3: W 2^6 | K e4m3: (W - W_hi) 2^18] (weight=True)."""
    hi = x32.to(torch.float16)
    lo = x32 - hi.float()
    a, b = ((x32 * 2.0 ** SPLIT_W_SHIFT, lo * 2.0 ** SPLIT_WLO_SHIFT) if weight else (lo * 2.0 ** SPLIT_LO_SHIFT, x32))
    rows = torch.cat([hi.contiguous().view(torch.uint8).reshape(x32.shape[0], -1), _e4m3_bytes(a), _e4m3_bytes(b)], dim=1).contiguous()
    return rows.view(torch.float16)


def split_decode(rows: torch.Tensor, K: int):
    """split rows (fp16 view [M, >= 2K]) -> (hi [M, K] fp16, lo8 [M, K] float = the stored (x - hi), hi8 [M, K] float = the stored x)"""
    b = rows.contiguous().view(torch.uint8)
    hi = b[:, :2 * K].contiguous().view(torch.float16)
    lo8 = b[:, 2 * K:3 * K].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** -SPLIT_LO_SHIFT
    hi8 = b[:, 3 * K:4 * K].contiguous().view(torch.float8_e4m3fn).float()
    return hi, lo8, hi8


# --------------------------------------------------------------------------------------------
def gemm(A: torch.Tensor, W: torch.Tensor, bias=None, resid=None, out_dtype=None, act=L.ACT_NONE, out=None,
         M=None, amap=None, cmap=None, ldc=None, scratch=None, w_scale=None, a_scale=0.0, out_scale=0.0, K=None, k8=0) -> torch.Tensor:
    """fp8 operands (torch.float8_e4m3fn A and W): w_scale [N] fp32, a_scale; out_dtype SPRC_FP8 also needs out_scale.
    Split-precision product: A and W are fp16 views of split rows (`split_rows`), K = the logical reduction length, k8 = 2 K."""
    lib = L.load()
    dt = _SPRC_DT[A.dtype]
    assert W.dtype == A.dtype and A.is_cuda and A.stride(-1) == 1 and W.stride(-1) == 1
    N = W.shape[0]
    K = W.shape[1] if K is None else K
    M = A.shape[0] if M is None else M
    odt = dt if out_dtype is None else out_dtype
    if out is None:
        out = torch.empty((M, 2 * N if odt == L.SPRC_F16X3 else N), dtype=torch.float16 if odt == L.SPRC_F16X3 else _TORCH_DT[odt], device=A.device)
    g = L.GemmArgs()
    g.k8 = k8
    g.M, g.N, g.K, g.dtype, g.out_dtype, g.act, g.max32 = M, N, K, dt, odt, act, 0
    g.A, g.lda, g.amap = A.data_ptr(), A.stride(0), amap or rowmap()
    g.W, g.ldw = W.data_ptr(), W.stride(0)
    g.bias = _ptr(bias)
    g.resid, g.ldr = _ptr(resid), (resid.stride(0) if resid is not None else 0)
    g.C, g.ldc, g.cmap = out.data_ptr(), (out.stride(0) if ldc is None else ldc), cmap or rowmap()
    if scratch is not None:                      # optional split-K scratch (include/sprc.h)
        g.scratch, g.scratch_bytes = scratch.data_ptr(), scratch.numel() * scratch.element_size()
    g.w_scale, g.a_scale, g.out_scale = _ptr(w_scale), a_scale, out_scale
    L.check(lib.sprc_gemm(C.byref(g), _stream()), "sprc_gemm")
    return out


def gemm_pair(A: torch.Tensor, W0: torch.Tensor, W1: torch.Tensor, bias0, bias1, amap0, amap1, cmap0, cmap1, M: int,
              out: torch.Tensor, resid=None, out_dtype=None, act=L.ACT_NONE, K=None, k8=0) -> torch.Tensor:
    """Two products of identical shape in one launch (sprc_gemm_pair): rows amap0 of A through W0 into rows cmap0 of `out`,
    rows amap1 through W1 into rows cmap1.  K / k8: a split-precision pair (operands = fp16 views of split rows, see `gemm`)."""
    lib = L.load()
    dt = _SPRC_DT[A.dtype]
    N = W0.shape[0]
    K = W0.shape[1] if K is None else K
    odt = dt if out_dtype is None else out_dtype
    gs = []
    for W, bias, amap, cmap in ((W0, bias0, amap0, cmap0), (W1, bias1, amap1, cmap1)):
        g = L.GemmArgs()
        g.k8 = k8
        g.M, g.N, g.K, g.dtype, g.out_dtype, g.act, g.max32 = M, N, K, dt, odt, act, 0
        g.A, g.lda, g.amap = A.data_ptr(), A.stride(0), amap
        g.W, g.ldw = W.data_ptr(), W.stride(0)
        g.bias = _ptr(bias)
        g.resid, g.ldr = _ptr(resid), (resid.stride(0) if resid is not None else 0)
        g.C, g.ldc, g.cmap = out.data_ptr(), out.stride(0), cmap
        gs.append(g)
    L.check(lib.sprc_gemm_pair(C.byref(gs[0]), C.byref(gs[1]), _stream()), "sprc_gemm_pair")
    return out


def layernorm(x: torch.Tensor, gamma, beta, eps: float, out_dtype: int, want32=True, want16=True,
              xmap=None, ymap=None, M=None, y32=None, y16=None, add16=None, sum32=None):
    lib = L.load()
    Mx, D = x.shape
    M = Mx if M is None else M
    if want32 and y32 is None:
        y32 = torch.empty((Mx, D), dtype=torch.float32, device=x.device)
    if want16 and y16 is None:
        y16 = torch.empty((Mx, D), dtype=_TORCH_DT[out_dtype], device=x.device)
    a = L.LayerNormArgs()
    a.M, a.D, a.out_dtype = M, D, out_dtype
    a.x, a.ldx, a.xmap = x.data_ptr(), x.stride(0), xmap or rowmap()
    a.gamma, a.beta, a.eps = gamma.data_ptr(), beta.data_ptr(), eps
    a.y32, a.ld32, a.ymap = _ptr(y32), D, ymap or rowmap()
    a.y16, a.ld16 = _ptr(y16), D
    if add16 is not None:                        # fused residual add: LN(x + add16); sum32 (optional) <- x + add16
        assert add16.dtype == torch.float16 and add16.stride(-1) == 1
        a.add16, a.ld_add = add16.data_ptr(), add16.stride(0)
        if sum32 is not None:
            a.sum32, a.ld_sum = sum32.data_ptr(), sum32.stride(0)
    L.check(lib.sprc_layernorm(C.byref(a), _stream()), "sprc_layernorm")
    return y32, y16


def attention(q, k, v, B, H, Tq, Tk, head_dim, ldq, ldk, ldv, scale, key_mask=None, out=None,
              k2=None, v2=None, Tk2=0, ld2=0, kv_index=None, kv2_index=None, drop=None, out_x3=False):
    """k2 / v2 (optional): a second key segment of Tk2 tokens appended to the key axis; kv_index / kv2_index: int32 [B] batch
    rows of the two segments (see sprc_attention_args).  drop = (p, seed, site): training-mode dropout on the probabilities (fp32).
    out_x3 (fp16, Tq <= 128): the output as split-precision rows (SPRC_F16X3: an fp16 view [B * Tq, 2 H head_dim], `split_decode`)."""
    lib = L.load()
    dt = _SPRC_DT[q.dtype]
    if out is None:
        out = torch.empty((B * Tq, (2 if out_x3 else 1) * H * head_dim), dtype=q.dtype, device=q.device)
    a = L.AttentionArgs()
    a.B, a.H, a.Tq, a.Tk, a.head_dim, a.dtype = B, H, Tq, Tk, head_dim, dt
    a.q, a.ldq, a.k, a.ldk, a.v, a.ldv = q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    a.key_mask, a.scale = _ptr(key_mask), scale
    if k2 is not None:
        a.k2, a.ldk2, a.v2, a.ldv2, a.Tk2 = k2.data_ptr(), ld2, v2.data_ptr(), ld2, Tk2
    a.kv_index, a.kv2_index = _ptr(kv_index), _ptr(kv2_index)
    if drop is not None:
        a.drop_p, a.drop_seed, a.drop_site = drop
    a.out_x3 = int(bool(out_x3))
    L.check(lib.sprc_attention(C.byref(a), _stream()), "sprc_attention")
    return out


def sim_max(fusion: torch.Tensor, feats: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sim[nq,N] = max_j <fusion[q], feats[n,j]>; fusion [nq,E], feats [N,32,E] (fp32 or bf16)."""
    lib = L.load()
    nq, E = fusion.shape
    N, J, E2 = feats.shape
    assert J == 32 and E2 == E and fusion.dtype == feats.dtype and fusion.is_contiguous() and feats.is_contiguous()
    dt = _SPRC_DT[fusion.dtype]
    if out is None:
        out = torch.empty((nq, N), dtype=torch.float32, device=fusion.device)
    L.check(lib.sprc_sim_max(fusion.data_ptr(), feats.data_ptr(), out.data_ptr(), out.stride(0), nq, N, E, dt, _stream()),
            "sprc_sim_max")
    return out


def topk(sim: torch.Tensor, k: int, gidx: Optional[torch.Tensor] = None, idx_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """First k entries of the stable argsort of fl32(1 - sim) per row -> (sim values, int32 indices)."""
    lib = L.load()
    assert sim.dtype == torch.float32 and sim.stride(-1) == 1
    nq, N = sim.shape
    vals = torch.empty((nq, k), dtype=torch.float32, device=sim.device)
    idx = torch.empty((nq, k), dtype=torch.int32, device=sim.device)
    if gidx is not None:
        assert gidx.dtype == torch.int32 and gidx.is_contiguous() and gidx.shape == sim.shape
    L.check(lib.sprc_topk(sim.data_ptr(), sim.stride(0), _ptr(gidx), idx_base, nq, N, k, vals.data_ptr(), idx.data_ptr(),
                          _stream()), "sprc_topk")
    return vals, idx


def rank_of(sim: torch.Tensor, listed: torch.Tensor) -> torch.Tensor:
    """Position of listed[q,l] in the stable order of row q (int32; -1 where listed < 0)."""
    lib = L.load()
    assert sim.dtype == torch.float32 and sim.stride(-1) == 1
    listed = listed.to(device=sim.device, dtype=torch.int32).contiguous()
    nq, N = sim.shape
    out = torch.empty_like(listed)
    L.check(lib.sprc_rank_of(sim.data_ptr(), sim.stride(0), listed.data_ptr(), nq, N, listed.shape[1], out.data_ptr(),
                             _stream()), "sprc_rank_of")
    return out


# --------------------------------------------------------------------------------------------
def _on_device(fn):
    """Run an Engine method with the engine's GPU current: the library's launches, stream lookups and per-device kernel
    attributes all refer to the current device (a process may hold engines on several GPUs)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapper


class Engine:
    """Packed weights + workspaces for one model on one GPU."""

    def __init__(self, cfg: SprcConfig, state_dict: Dict[str, torch.Tensor], device, dtype: str = "bf16",
                 max_batch: int = 128, fp8_amax: Optional[torch.Tensor] = None, fp8_margin: float = 1.0,
                 qformer_x3=None, fp8_base: str = "bf16", fp8_layers: str = "all"):
        """dtype "fp8": a 16-bit engine (`fp8_base`: "bf16", or "fp16" with the split-precision Q-Former) whose ViT qkv /
        fc1 / fc2 GEMMs (`fp8_layers` "all") or fc1 / fc2 GEMMs only ("mlp") run on e4m3fn operands (BASELINE.json config C5:
        a throughput configuration -- the e4m3 noise, 2e-2 rms on structured scores, dwarfs what the base dtype or the layer
        choice change: tests/test_fp8_gpu.py, tools/fp8_sweep.py).
        fp8_amax [depth, 3]: max |x| of those GEMMs' inputs from `calibrate_fp8` on representative images (static
        per-tensor activation scales = amax * fp8_margin / 448); weights get per-output-channel scales.
        dtype "fp16": fp16 MFMA operands -- the reference's GPU numerics are a fp16-autocast ViT and an fp32 Q-Former
        (align_prompt.py:366-368: the Q-Former runs OUTSIDE `maybe_autocast`); qformer_x3 (default: on for fp16) keeps the
        Q-Former at ~fp32 product precision on the fp16 MFMA by splitting weights and activations into hi + lo halves
        (SPRC_F16X3, include/sprc.h): an fp16 product + two e4m3 correction products at twice the rate, instead of the 16x slower
        exact-fp32 MFMA."""
        self.lib = L.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SprcError("sprc_amd.Engine needs a GPU device: the HIP kernels are the only compute path")
        self.fp8 = dtype == "fp8"
        if self.fp8:
            if fp8_amax is None or tuple(fp8_amax.shape) != (cfg.vit.depth, 3):
                raise ValueError(f"the fp8 engine needs fp8_amax [{cfg.vit.depth}, 3] from Engine.calibrate_fp8")
            self._act_scale = (fp8_amax.detach().float().cpu().clamp_min(1e-6) * fp8_margin / FP8_MAX).tolist()
            if fp8_base not in ("fp16", "bf16") or fp8_layers not in ("all", "mlp"):
                raise ValueError(f"fp8_base {fp8_base!r} / fp8_layers {fp8_layers!r}")
        self.fp8_mode = 0 if not self.fp8 else L.FP8_ALL if fp8_layers == "all" else L.FP8_MLP
        self.dt = L.DTYPES[fp8_base if self.fp8 else dtype]
        self.is16 = L.is16(self.dt)               # bf16 or fp16 MFMA operands (fp32 accumulate / residual stream / LN / softmax)
        self.tdt = _TORCH_DT[self.dt]
        # bit masks over the Q-Former's layer kinds (X3_*) for the image pass and for the query-side passes: None = the default
        # of the dtype, True / False = all / none, an int = that mask for both, a pair = (image mask, query mask)
        if qformer_x3 is None:
            qformer_x3 = X3_DEFAULT if self.dt == L.SPRC_F16 else 0
        if isinstance(qformer_x3, (tuple, list)):
            self.x3_image, self.x3_fuse = (int(v) for v in qformer_x3)
        else:
            self.x3_image = self.x3_fuse = X3_ALL if qformer_x3 is True else int(qformer_x3)
        self.x3 = self.x3_image | self.x3_fuse      # what the weights are packed for
        # fp16 ViT: split-precision patch embedding (SPRC_PATCH_X3=0: A/B switch)
        self.patch_x3 = self.dt == L.SPRC_F16 and os.environ.get("SPRC_PATCH_X3", "1") != "0"
        if self.x3 and self.dt != L.SPRC_F16:
            raise ValueError("qformer_x3 (split-precision Q-Former) is a mode of the fp16 engine")
        self.max_batch = max_batch
        self._keep: List[torch.Tensor] = []
        self._ws: Dict[str, torch.Tensor] = {}
        self._pack_vit(state_dict)
        self._pack_qformer(state_dict)

    # ---- packing -------------------------------------------------------------------------
    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _w(self, t: torch.Tensor, k_pad: Optional[int] = None) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=torch.float32)
        if k_pad is not None and k_pad != t.shape[1]:
            t = torch.nn.functional.pad(t, (0, k_pad - t.shape[1]))
        t = t.to(self.tdt).contiguous()
        self._keep.append(t)
        return t

    def _lin(self, w: torch.Tensor, b: Optional[torch.Tensor], k_pad: Optional[int] = None) -> L.Linear:
        return L.Linear(self._w(w, k_pad).data_ptr(), None if b is None else self._f32(b).data_ptr())

    def _split_rows(self, w32: torch.Tensor) -> torch.Tensor:
        """[out, in] fp32 -> the weight side of a split-precision product (sprc.h: SPRC_F16X3), uint8 [out, 4 in]:
        [ in x fp16: W_hi | in x e4m3: W * 2^6 | in x e4m3: (W - W_hi) * 2^18 ]."""
        # the e4m3 correction segments SATURATE at +-448: W 2^6 clips for |W| > 7, (W - W_hi) 2^18 for |W| > ~3.5 (half an fp16 ulp of
        # 4 .. 8 is 2^-9).  Past that the correction terms are wrong and the product silently falls back to plain-fp16 accuracy, so the
        # range is checked when the weights are packed.  Activation side (the producers' kernels, common.hpp: store_split4):
        # the residual (x - x_hi) 2^12 clips for |x| > ~224, x itself for |x| > 448 -- LayerNorm outputs and GELU hiddens of a trained
        # Q-Former are O(1 .. 30); SPRC_X3_CHECK=1 makes split_rows_checked() test an activation tensor in debug runs.
        amax = float(w32.abs().max()) if w32.numel() else 0.0
        if amax > SPLIT_W_ABSMAX:
            import warnings
            warnings.warn(f"split-precision weight rows: |W| max {amax:.3g} exceeds {SPLIT_W_ABSMAX} -- the e4m3 correction segments saturate "
                          "and this layer's products fall back to plain fp16 accuracy (pack the layer without its X3_* bit, or rescale it)", RuntimeWarning)
        rows = split_rows(w32, weight=True)
        assert rows.shape == (w32.shape[0], 2 * w32.shape[1])
        self._keep.append(rows)
        return rows

    def _lin_q(self, w: torch.Tensor, b: Optional[torch.Tensor], kind: int) -> L.Linear:
        """A Q-Former linear of layer kind `kind` (X3_*): plain compute-dtype weights, or -- when the kind's bit is set in the
        split-precision mask -- split rows (`_split_rows`)."""
        if not (self.x3 & kind):
            return self._lin(w, b)
        rows = self._split_rows(w.detach().to(device=self.device, dtype=torch.float32))
        return L.Linear(rows.data_ptr(), None if b is None else self._f32(b).data_ptr())

    def _lin_patch(self, w: torch.Tensor, b: Optional[torch.Tensor]) -> L.Linear:
        """The patch embedding's weights [width, 3 P P] padded to patch_k_pad columns; fp16 engine: split precision
        (sprc_vit_model.patch_x3; rows [W_hi | W 2^6 | W_lo 2^18] as every split weight: the embedding's output is the first value of the residual stream, its rounding error is never averaged away)."""
        if not self.patch_x3:
            return self._lin(w, b, self.patch_k_pad)
        w32 = torch.nn.functional.pad(w.detach().to(device=self.device, dtype=torch.float32), (0, self.patch_k_pad - w.shape[1]))
        rows = self._split_rows(w32)
        return L.Linear(rows.data_ptr(), None if b is None else self._f32(b).data_ptr())

    def _lin8(self, w: torch.Tensor, b: Optional[torch.Tensor]):
        """-> (Linear with e4m3fn weights, device pointer of the per-output-channel scales)"""
        wq, scale = quantize_fp8_rows(w.detach().to(self.device))
        self._keep += [wq, scale]
        return L.Linear(wq.data_ptr(), None if b is None else self._f32(b).data_ptr()), scale.data_ptr()

    def _pack_vit(self, sd):
        v = self.cfg.vit
        p = "visual_encoder."
        D = v.width
        kq = 64 if self.is16 else 32
        self.patch_k_pad = (v.patch_k + kq - 1) // kq * kq
        layers = (L.VitLayer * v.depth)()
        for i in range(v.depth):
            ly = layers[i]
            if v.kind == "eva_g":
                b = f"{p}blocks.{i}."
                qkv_b = torch.cat([sd[b + "attn.q_bias"].float(), torch.zeros_like(sd[b + "attn.v_bias"]).float(),
                                   sd[b + "attn.v_bias"].float()])               # eva_vit.py:120-122
                names = ("norm1", "norm2", "attn.qkv.weight", "attn.proj", "mlp.fc1", "mlp.fc2")
            else:
                b = f"{p}transformer.resblocks.{i}."
                names = ("ln_1", "ln_2", "attn.in_proj_weight", "attn.out_proj", "mlp.c_fc", "mlp.c_proj")
                qkv_b = sd[b + "attn.in_proj_bias"]
            if self.fp8:
                if self.fp8_mode == L.FP8_ALL:
                    ly.qkv, ly.qkv_ws = self._lin8(sd[b + names[2]], qkv_b)
                else:
                    ly.qkv = self._lin(sd[b + names[2]], qkv_b)
                ly.fc1, ly.fc1_ws = self._lin8(sd[b + names[4] + ".weight"], sd[b + names[4] + ".bias"])
                ly.fc2, ly.fc2_ws = self._lin8(sd[b + names[5] + ".weight"], sd[b + names[5] + ".bias"])
                ly.s_ln1, ly.s_ln2, ly.s_mlp = self._act_scale[i]
            else:
                ly.qkv = self._lin(sd[b + names[2]], qkv_b)
                ly.fc1 = self._lin(sd[b + names[4] + ".weight"], sd[b + names[4] + ".bias"])
                ly.fc2 = self._lin(sd[b + names[5] + ".weight"], sd[b + names[5] + ".bias"])
            ly.ln1_w, ly.ln1_b = self._f32(sd[b + names[0] + ".weight"]).data_ptr(), self._f32(sd[b + names[0] + ".bias"]).data_ptr()
            ly.ln2_w, ly.ln2_b = self._f32(sd[b + names[1] + ".weight"]).data_ptr(), self._f32(sd[b + names[1] + ".bias"]).data_ptr()
            ly.proj = self._lin(sd[b + names[3] + ".weight"], sd[b + names[3] + ".bias"])
        m = L.VitModel()
        m.dtype, m.width, m.depth, m.heads, m.head_dim, m.mlp = self.dt, D, v.depth, v.heads, v.head_dim, v.mlp
        m.act = L.ACT_GELU if v.act == "gelu" else L.ACT_QUICKGELU
        m.tokens, m.patch_size, m.image, m.patch_k_pad = v.tokens, v.patch, v.image, self.patch_k_pad
        m.has_ln_pre, m.ln_eps, m.ln_vision_eps = int(v.ln_pre), v.ln_eps, self.cfg.ln_vision_eps
        if v.kind == "eva_g":
            m.patch = self._lin_patch(sd[p + "patch_embed.proj.weight"].reshape(D, -1), sd[p + "patch_embed.proj.bias"])
            m.cls = self._f32(sd[p + "cls_token"].reshape(D)).data_ptr()
            m.pos = self._f32(sd[p + "pos_embed"].reshape(v.tokens, D)).data_ptr()
        else:
            m.patch = self._lin_patch(sd[p + "conv1.weight"].reshape(D, -1), None)
            m.cls = self._f32(sd[p + "class_embedding"].reshape(D)).data_ptr()
            m.pos = self._f32(sd[p + "positional_embedding"].reshape(v.tokens, D)).data_ptr()
            m.ln_pre_w = self._f32(sd[p + "ln_pre.weight"]).data_ptr()
            m.ln_pre_b = self._f32(sd[p + "ln_pre.bias"]).data_ptr()
        m.ln_vision_w = self._f32(sd["ln_vision.weight"]).data_ptr()
        m.ln_vision_b = self._f32(sd["ln_vision.bias"]).data_ptr()
        m.layers = C.cast(layers, C.POINTER(L.VitLayer))
        m.fp8 = int(self.fp8_mode)
        m.patch_x3 = int(self.patch_x3)
        self._vit_layers, self.vit = layers, m

    def _pack_qformer(self, sd):
        q = self.cfg.qformer
        p = "Qformer.bert."
        layers = (L.QfLayer * q.layers)()
        ckv_w, ckv_b, n_cross = [], [], 0

        def ln(prefix):
            return self._f32(sd[prefix + ".weight"]).data_ptr(), self._f32(sd[prefix + ".bias"]).data_ptr()

        for l in range(q.layers):
            b = f"{p}encoder.layer.{l}."
            ly = layers[l]
            a = b + "attention."
            ly.qkv = self._lin_q(torch.cat([sd[a + "self.query.weight"], sd[a + "self.key.weight"], sd[a + "self.value.weight"]]).float(),
                               torch.cat([sd[a + "self.query.bias"], sd[a + "self.key.bias"], sd[a + "self.value.bias"]]).float(), X3_QKV)
            ly.attn_out = self._lin_q(sd[a + "output.dense.weight"], sd[a + "output.dense.bias"], X3_ATTN_OUT)
            ly.attn_ln_w, ly.attn_ln_b = ln(a + "output.LayerNorm")
            if l % q.cross_freq == 0:
                c = b + "crossattention."
                ly.has_cross, ly.cross_index = 1, n_cross
                ly.cq = self._lin_q(sd[c + "self.query.weight"], sd[c + "self.query.bias"], X3_CROSS_Q)
                ckv_w += [sd[c + "self.key.weight"].float(), sd[c + "self.value.weight"].float()]
                ckv_b += [sd[c + "self.key.bias"].float(), sd[c + "self.value.bias"].float()]
                ly.cross_out = self._lin_q(sd[c + "output.dense.weight"], sd[c + "output.dense.bias"], X3_CROSS_OUT)
                ly.cross_ln_w, ly.cross_ln_b = ln(c + "output.LayerNorm")
                n_cross += 1
            ly.ffn_t_in = self._lin_q(sd[b + "intermediate.dense.weight"], sd[b + "intermediate.dense.bias"], X3_FFN_IN)
            ly.ffn_t_out = self._lin_q(sd[b + "output.dense.weight"], sd[b + "output.dense.bias"], X3_FFN_OUT)
            ly.ffn_t_ln_w, ly.ffn_t_ln_b = ln(b + "output.LayerNorm")
            ly.ffn_q_in = self._lin_q(sd[b + "intermediate_query.dense.weight"], sd[b + "intermediate_query.dense.bias"], X3_FFN_IN)
            ly.ffn_q_out = self._lin_q(sd[b + "output_query.dense.weight"], sd[b + "output_query.dense.bias"], X3_FFN_OUT)
            ly.ffn_q_ln_w, ly.ffn_q_ln_b = ln(b + "output_query.LayerNorm")
        m = L.QformerModel()
        m.dtype, m.hidden, m.n_layers, m.heads, m.head_dim, m.ffn = self.dt, q.hidden, q.layers, q.heads, q.head_dim, q.ffn
        m.num_query, m.enc_width, m.embed_dim, m.max_txt, m.n_cross = q.num_query, self.cfg.vit.width, self.cfg.embed_dim, self.cfg.max_txt_len, n_cross
        m.ln_eps, m.vocab = q.ln_eps, q.vocab
        m.word_emb = self._f32(sd[p + "embeddings.word_embeddings.weight"]).data_ptr()
        m.pos_emb = self._f32(sd[p + "embeddings.position_embeddings.weight"]).data_ptr()
        m.emb_ln_w, m.emb_ln_b = ln(p + "embeddings.LayerNorm")
        m.query_tokens = self._f32(sd["query_tokens"].reshape(q.num_query, q.hidden)).data_ptr()
        m.ckv_all = self._lin_q(torch.cat([w.to(self.device) for w in ckv_w]), torch.cat([b_.to(self.device) for b_ in ckv_b]), X3_CKV)
        m.vision_proj = self._lin_q(sd["vision_proj.weight"], sd["vision_proj.bias"], X3_HEADS)
        m.text_proj = self._lin_q(sd["text_proj.weight"], sd["text_proj.bias"], X3_HEADS)
        # image-text-matching head of the stage-2 rerank (blip2_qformer_cir_rerank.py:88): fp32, two rows
        self.itm_w = self._f32(sd["itm_head.weight"]) if "itm_head.weight" in sd else None
        self.itm_b = self._f32(sd["itm_head.bias"]) if "itm_head.bias" in sd else None
        m.layers = C.cast(layers, C.POINTER(L.QfLayer))
        m.x3, m.x3_image, m.x3_fuse = int(self.x3), int(self.x3_image), int(self.x3_fuse)
        self._qf_layers, self.qf = layers, m
        # learned prompt tokens of the training losses (align_prompt.py:76-79, :170-193)
        self.prompt_tokens = self._f32(sd["prompt_tokens"].reshape(q.num_query, q.hidden)) if "prompt_tokens" in sd else None

    # ---- fp8 calibration ---------------------------------------------------------------------
    @_on_device
    def calibrate_fp8(self, images: torch.Tensor) -> torch.Tensor:
        """amax[depth, 3] = max |x| of the inputs of the qkv / fc1 / fc2 GEMMs of every block over `images`, collected by
        sprc_vit_forward on THIS (bf16 or fp16) engine; feed it to Engine(..., dtype="fp8", fp8_amax=...)."""
        if not self.is16 or self.fp8:
            raise L.SprcError("calibrate_fp8 runs on a bf16 or fp16 engine")
        amax = torch.zeros((self.cfg.vit.depth, 3), dtype=torch.float32, device=self.device)
        self.vit.calib_amax = amax.data_ptr()
        try:
            self.vit_forward(images)
            torch.cuda.synchronize(self.device)
        finally:
            self.vit.calib_amax = None
        return amax.cpu()

    # ---- workspaces ------------------------------------------------------------------------
    def _workspace(self, kind: str, B: int) -> torch.Tensor:
        is_vit = kind.startswith("vit")                                                                         # "vit", "vit#1", "qf", "qf_image"
        fn = self.lib.sprc_vit_workspace_bytes if is_vit else self.lib.sprc_qformer_workspace_bytes
        need = int(fn(C.byref(self.vit if is_vit else self.qf), B))
        ws = self._ws.get(kind)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[kind] = ws
        return ws

    # ---- forward passes ----------------------------------------------------------------------
    @_on_device
    def vit_forward(self, images: torch.Tensor, out: Optional[torch.Tensor] = None, pre_ln_out: Optional[torch.Tensor] = None, slot: int = 0) -> torch.Tensor:
        """raw[B,257,D] fp32 = ln_vision(ViT(images)).  pre_ln_out (optional, fp32 [B*257, D] or [B,257,D]): receives ln_vision's INPUT
        (the training step differentiates ln_vision, sprc_amd/train.py).  slot: which activation workspace to use -- two forwards that are
        in flight on two streams at once (bench.py --vit-overlap 2) need one each."""
        v = self.cfg.vit
        images = images.to(device=self.device, dtype=torch.float32).contiguous()
        B = images.shape[0]
        if tuple(images.shape[1:]) != (3, v.image, v.image):
            raise ValueError(f"Input image size ({images.shape[2]}*{images.shape[3]}) doesn't match model ({v.image}*{v.image}).")
        raw = out if out is not None else torch.empty((B, v.tokens, v.width), dtype=torch.float32, device=self.device)
        if pre_ln_out is not None:
            assert pre_ln_out.dtype == torch.float32 and pre_ln_out.is_contiguous() and pre_ln_out.numel() == B * v.tokens * v.width
        try:
            for s in range(0, B, self.max_batch):
                n = min(self.max_batch, B - s)
                ws = self._workspace("vit" if slot == 0 else f"vit#{slot}", n)
                if pre_ln_out is not None:
                    self.vit.pre_ln_out = pre_ln_out.data_ptr() + s * v.tokens * v.width * 4
                L.check(self.lib.sprc_vit_forward(C.byref(self.vit), images[s:s + n].data_ptr(), n, raw[s:s + n].data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _stream(self.device)), "sprc_vit_forward")
        finally:
            self.vit.pre_ln_out = None
        return raw

    @_on_device
    def qformer_image(self, raw: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """feats[B,32,E] fp32 (unit rows) + compute-dtype copy (bf16 mode)."""
        raw = raw.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(raw.shape[1:]) != (self.cfg.vit.tokens, self.cfg.vit.width):        # the C call plans for exactly this shape
            raise ValueError(f"raw embeddings must be [B, {self.cfg.vit.tokens}, {self.cfg.vit.width}], got {tuple(raw.shape)}")
        B, E, Lq = raw.shape[0], self.cfg.embed_dim, self.cfg.qformer.num_query
        feats = torch.empty((B, Lq, E), dtype=torch.float32, device=self.device)
        f16 = torch.empty((B, Lq, E), dtype=self.tdt, device=self.device) if self.is16 else None
        for s in range(0, B, self.max_batch):
            n = min(self.max_batch, B - s)
            ws = self._workspace("qf_image", n)          # its own workspace: the image pass may run on a side stream beside a fusion pass
            L.check(self.lib.sprc_qformer_image(C.byref(self.qf), raw[s:s + n].data_ptr(), n, feats[s:s + n].data_ptr(),
                                                None if f16 is None else f16[s:s + n].data_ptr(), ws.data_ptr(), ws.numel(),
                                                _stream(self.device)), "sprc_qformer_image")
        return feats, f16

    @_on_device
    def qformer_fuse(self, ref_embeds: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        """fusion[B,E] fp32 (unit rows) + compute-dtype copy (bf16 mode)."""
        ref = ref_embeds.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(ref.shape[1:]) != (self.cfg.vit.tokens, self.cfg.vit.width):
            raise ValueError(f"reference embeddings must be [B, {self.cfg.vit.tokens}, {self.cfg.vit.width}], got {tuple(ref.shape)}")
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        B, E = ref.shape[0], self.cfg.embed_dim
        if ids.shape != (B, self.cfg.max_txt_len) or mask.shape != ids.shape:
            raise ValueError(f"input_ids/attention_mask must be [{B},{self.cfg.max_txt_len}]")
        if not input_ids.is_cuda and (int(input_ids.min()) < 0 or int(input_ids.max()) >= self.cfg.qformer.vocab):
            raise IndexError("token id out of range")       # device-resident ids are clamped by the kernel (no host sync)
        fusion = torch.empty((B, E), dtype=torch.float32, device=self.device)
        f16 = torch.empty((B, E), dtype=self.tdt, device=self.device) if self.is16 else None
        for s in range(0, B, self.max_batch):
            n = min(self.max_batch, B - s)
            ws = self._workspace("qf", n)
            L.check(self.lib.sprc_qformer_fuse(C.byref(self.qf), ref[s:s + n].data_ptr(), ref.shape[1], ids[s:s + n].data_ptr(),
                                               mask[s:s + n].data_ptr(), n, fusion[s:s + n].data_ptr(),
                                               None if f16 is None else f16[s:s + n].data_ptr(), ws.data_ptr(), ws.numel(),
                                               _stream(self.device)), "sprc_qformer_fuse")
        return fusion, f16

    @_on_device
    def qformer_fuse_kv(self, kv: torch.Tensor, kv_index: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        """`qformer_fuse` with the reference images' cross-attention K|V projections GIVEN: kv = `encode_kv` rows [n, 257, kv_width]
        (compute dtype), query b reads row kv_index[b].  An image that is the reference of several queries is projected once (6.7 of a
        query's 29.3 GFLOP); same bits as `qformer_fuse` on the same images.  Optional: bench.py does not use it."""
        if not self.is16 or kv.dtype != self.tdt or kv.dim() != 3 or kv.shape[1:] != (self.cfg.vit.tokens, self.kv_width) or not kv.is_contiguous():
            raise ValueError("kv must be a contiguous Engine.encode_kv output [n, tokens, kv_width] of a 16-bit engine")
        idx = kv_index.to(device=self.device, dtype=torch.int32).contiguous()
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        B, E = idx.shape[0], self.cfg.embed_dim
        if ids.shape != (B, self.cfg.max_txt_len) or mask.shape != ids.shape:
            raise ValueError(f"input_ids/attention_mask must be [{B},{self.cfg.max_txt_len}]")
        if B and (int(idx.min()) < 0 or int(idx.max()) >= kv.shape[0]):
            raise IndexError("kv_index out of range")
        fusion = torch.empty((B, E), dtype=torch.float32, device=self.device)
        f16 = torch.empty((B, E), dtype=self.tdt, device=self.device)
        for s in range(0, B, self.max_batch):
            n = min(self.max_batch, B - s)
            ws = self._workspace("qf", n)
            L.check(self.lib.sprc_qformer_fuse_kv(C.byref(self.qf), kv.data_ptr(), kv.shape[1], idx[s:s + n].data_ptr(), ids[s:s + n].data_ptr(),
                                                  mask[s:s + n].data_ptr(), n, fusion[s:s + n].data_ptr(), f16[s:s + n].data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _stream(self.device)), "sprc_qformer_fuse_kv")
        return fusion, f16

    @_on_device
    def qformer_text(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """feat[B,E] fp32 (unit rows) = normalize(text_proj(Qformer(text)[:, 0])): the Q-Former as a plain text encoder -- the
        stage-1 query feature of the rerank model class (blip2_qformer_cir_rerank.py:373-390)."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        B = ids.shape[0]
        if ids.shape != (B, self.cfg.max_txt_len) or mask.shape != ids.shape:
            raise ValueError(f"input_ids/attention_mask must be [{B},{self.cfg.max_txt_len}]")
        if not input_ids.is_cuda and (int(input_ids.min()) < 0 or int(input_ids.max()) >= self.cfg.qformer.vocab):
            raise IndexError("token id out of range")
        feat = torch.empty((B, self.cfg.embed_dim), dtype=torch.float32, device=self.device)
        for s in range(0, B, self.max_batch):
            n = min(self.max_batch, B - s)
            ws = self._workspace("qf", n)
            L.check(self.lib.sprc_qformer_text(C.byref(self.qf), ids[s:s + n].data_ptr(), mask[s:s + n].data_ptr(), n,
                                               feat[s:s + n].data_ptr(), None, ws.data_ptr(), ws.numel(), _stream(self.device)),
                    "sprc_qformer_text")
        return feat

    # ---- stage-2 rerank ------------------------------------------------------------------
    @property
    def kv_width(self) -> int:
        return self.qf.n_cross * 2 * self.cfg.qformer.hidden

    @_on_device
    def encode_kv(self, raw: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """kv[B, tokens, n_cross*2*hidden] (compute dtype): cross-attention K|V projections of image tokens, once per image."""
        raw = raw.to(device=self.device, dtype=torch.float32).contiguous()
        B, T, D = raw.shape
        if D != self.cfg.vit.width:
            raise ValueError(f"raw embeddings must be [B, tokens, {self.cfg.vit.width}]")
        kv = out if out is not None else torch.empty((B, T, self.kv_width), dtype=self.tdt, device=self.device)
        step = max(1, min(self.max_batch, 256))
        for s in range(0, B, step):
            n = min(step, B - s)
            need = int(self.lib.sprc_qformer_kv_workspace_bytes(C.byref(self.qf), n, T))
            ws = self._ws.get("kv")
            if ws is None or ws.numel() < need:
                ws = self._ws["kv"] = torch.empty(need, dtype=torch.uint8, device=self.device)
            L.check(self.lib.sprc_qformer_encode_kv(C.byref(self.qf), raw[s:s + n].data_ptr(), n, T, kv[s:s + n].data_ptr(),
                                                    ws.data_ptr(), ws.numel(), _stream(self.device)), "sprc_qformer_encode_kv")
        return kv

    @_on_device
    def itm(self, kv_a: torch.Tensor, index_a: torch.Tensor, kv_b: torch.Tensor, index_b: torch.Tensor,
            input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """prob[P] = P(match) of P (query, candidate) pairs: pair p attends over cat(kv_a[index_a[p]], kv_b[index_b[p]])."""
        if self.itm_w is None:
            raise L.SprcError("the state dict has no itm_head.*: stage-2 rerank needs the rerank checkpoint's ITM head")
        ia = index_a.to(device=self.device, dtype=torch.int32).contiguous()
        ib = index_b.to(device=self.device, dtype=torch.int32).contiguous()
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        P = ia.shape[0]
        if ib.shape[0] != P or ids.shape != (P, self.cfg.max_txt_len) or mask.shape != ids.shape:
            raise ValueError("index_a / index_b / input_ids / attention_mask must describe the same P pairs")
        for kv in (kv_a, kv_b):
            if kv.dtype != self.tdt or kv.dim() != 3 or kv.shape[2] != self.kv_width or not kv.is_contiguous():
                raise ValueError("kv_a / kv_b must come from Engine.encode_kv")
        if P and (int(ia.max()) >= kv_a.shape[0] or int(ib.max()) >= kv_b.shape[0] or int(ia.min()) < 0 or int(ib.min()) < 0):
            raise IndexError("pair index out of range")
        prob = torch.empty((P,), dtype=torch.float32, device=self.device)
        for s in range(0, P, self.max_batch):
            n = min(self.max_batch, P - s)
            need = int(self.lib.sprc_qformer_itm_workspace_bytes(C.byref(self.qf), n))
            ws = self._ws.get("itm")
            if ws is None or ws.numel() < need:
                ws = self._ws["itm"] = torch.empty(need, dtype=torch.uint8, device=self.device)
            L.check(self.lib.sprc_qformer_itm(C.byref(self.qf), self.itm_w.data_ptr(), self.itm_b.data_ptr(),
                                              kv_a.data_ptr(), kv_a.shape[1], ia[s:s + n].data_ptr(),
                                              kv_b.data_ptr(), kv_b.shape[1], ib[s:s + n].data_ptr(),
                                              ids[s:s + n].data_ptr(), mask[s:s + n].data_ptr(), n, prob[s:s + n].data_ptr(),
                                              ws.data_ptr(), ws.numel(), _stream(self.device)), "sprc_qformer_itm")
        return prob

    # ---- training forward (forward only) ------------------------------------------------------
    @_on_device
    def training_losses(self, image: torch.Tensor, target: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                        temp: float = 0.07) -> Dict[str, torch.Tensor]:
        """The three losses of `Blip2QformerCirAlignPrompt.forward` (align_prompt.py:95-200), eval semantics:
        loss_itc = CE(max-cosine(fusion, target feats) / temp), loss_rtc = CE(max-cosine(text-only prompt feat, target feats) / temp),
        loss_align = MSE(mean fused query token of pass 1, mean prompt token).  No autograd history (no backward kernels)."""
        if self.prompt_tokens is None:
            raise L.SprcError("the state dict has no prompt_tokens: the training losses need them (align_prompt.py:76-79)")
        B = image.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        raw_ref, raw_tgt = self.vit_forward(image), self.vit_forward(target)
        target_feats, _ = self.qformer_image(raw_tgt)
        E_ = self.cfg.embed_dim
        fusion = torch.empty((B, E_), dtype=torch.float32, device=self.device)
        tfeat = torch.empty((B, E_), dtype=torch.float32, device=self.device)
        losses = torch.zeros(3, dtype=torch.float32, device=self.device)
        ws = self._workspace("qf", B)
        st = _stream(self.device)
        L.check(self.lib.sprc_qformer_fuse_train(C.byref(self.qf), raw_ref.data_ptr(), raw_ref.shape[1], ids.data_ptr(), mask.data_ptr(), B,
                                                 fusion.data_ptr(), None, self.prompt_tokens.data_ptr(), losses[2:].data_ptr(),
                                                 ws.data_ptr(), ws.numel(), st), "sprc_qformer_fuse_train")
        L.check(self.lib.sprc_qformer_text_only(C.byref(self.qf), self.prompt_tokens.data_ptr(), ids.data_ptr(), mask.data_ptr(), B,
                                                tfeat.data_ptr(), None, ws.data_ptr(), ws.numel(), st), "sprc_qformer_text_only")
        sim = torch.empty((2, B, B), dtype=torch.float32, device=self.device)
        sim_max(fusion, target_feats, out=sim[0])
        sim_max(tfeat, target_feats, out=sim[1])
        for i in range(2):
            L.check(self.lib.sprc_contrastive_ce(sim[i].data_ptr(), B, B, float(temp), losses[i:].data_ptr(), st), "sprc_contrastive_ce")
        return {"loss_itc": losses[0], "loss_rtc": losses[1], "loss_align": losses[2]}
