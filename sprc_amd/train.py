"""Training step of `Blip2QformerCirAlignPrompt`: forward AND backward of
lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:95-200 as the reference's fine-tuning loop drives it
(src/blip_fine_tune_2.py:281-304: `loss = loss_itc + w_rtc loss_rtc + w_align loss_align; scaler.scale(loss).backward()`).

What trains in the reference is everything but the ViT trunk (align_prompt.py:64-69): the Q-Former (four passes share its weights:
fusion pass 1 and pass 2, the target-image pass, the text-only prompt pass), ln_vision, vision_proj / text_proj, query_tokens,
prompt_tokens and temp.  This module sequences the library's fp32 kernels -- sprc_gemm (exact-fp32 MFMA), sprc_attention,
sprc_layernorm and the backward kernels of csrc/train.hip -- into that graph; torch provides device memory and copies (slicing,
concatenation) only.  Every gradient ACCUMULATION happens inside a kernel: products accumulate through the GEMM's residual
epilogue (dW += dY^T X is `sprc_gemm(A = dY^T, W = X^T, resid = dW)`), the embedding scatters and the loss kernels add in place.
Dropout (the reference trains with the Q-Former in train mode, blip_fine_tune_2.py:290: p = 0.1 at Qformer.py:113,264,293,379) is
`TrainStep(dropout_p=...)`: counter-based masks (sprc_dropout_f32 / sprc_attention.drop_*) regenerated in backward, never stored;
dropout_p = 0 is the eval-mode graph.  The frozen ViT is always in eval mode (align_prompt.py:67-68).

    step = TrainStep(cfg, params, engine)            # params: {state-dict name: fp32 CUDA tensor}
    losses = step.forward(image, target, input_ids, attention_mask)
    grads = step.backward({"loss_itc": 1.0, "loss_rtc": 0.4, "loss_align": 0.4})     # {name: gradient}
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import engine as E

F32 = L.SPRC_F32


def _st():
    return torch.cuda.current_stream().cuda_stream


class _K:
    """fp32 kernel wrappers (one C call each)."""

    def __init__(self, device, half: bool = False):
        self.lib = L.load()
        self.dev = device
        self.half = bool(half)                   # products on fp16 OPERAND COPIES (fp32 accumulate / outputs): the reference's autocast arithmetic
        self._c16: List[tuple] = []              # the last few fp16 operand copies (x, x16): q / k / v projections read one input back to back
        self._ws: Dict[str, torch.Tensor] = {}
        self._xt: Dict[int, tuple] = {}          # transposes of SAVED activations, per step: id(x) -> (x, x^T); holding x keeps its address its own

    def transpose_saved(self, x, last=False):
        """x^T of a saved activation that SEVERAL linears read (the q / k / v projections of an attention): formed once, dropped by its
        last consumer (`last`), so that at most one attention's inputs are held transposed at a time.  Inputs with one
        consumer and gradients (dy) are transposed transiently."""
        hit = self._xt.get(id(x))
        if hit is None or hit[0] is not x:
            hit = self._xt[id(x)] = (x, self.transpose(x))
        if last:
            self._xt.pop(id(x), None)
        return hit[1]

    def empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def zeros(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.dev)

    def ws(self, key, nbytes):
        w = self._ws.get(key)
        if w is None or w.numel() < nbytes:
            w = self._ws[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.dev)
        return w

    def op16(self, x):
        """fp16 operand copy of an fp32 [R, K] tensor (sprc_cast_f32_to_16); fp16 inputs pass through"""
        if x.dtype == torch.float16:
            return x
        for src, c in self._c16:
            if src is x:
                return c
        assert x.is_contiguous()
        c = torch.empty(x.shape, dtype=torch.float16, device=self.dev)
        L.check(self.lib.sprc_cast_f32_to_16(x.data_ptr(), c.data_ptr(), x.numel(), L.SPRC_F16, _st()), "sprc_cast_f32_to_16")
        self._c16 = (self._c16 + [(x, c)])[-3:]
        return c

    # y = x . W^T (+ bias) (+ resid), fp32 out; `out` may alias `resid`.  half: the product runs on fp16 copies of x and W
    def gemm(self, x, W, bias=None, resid=None, out=None):
        if self.half:
            x, W = self.op16(x), self.op16(W)
            # `out` is about to be overwritten (in-place accumulation: resid is out): an fp16 copy cached for it is stale from here on
            # (nothing read one today only because every consumer of an accumulated dx went through ln_bwd first)
            if out is not None:
                self._c16 = [(src, c) for src, c in self._c16 if src is not out and src.data_ptr() != out.data_ptr()]
        return E.gemm(x, W, bias=bias, resid=resid, out_dtype=F32, out=out)

    def transpose(self, x, pad=32):
        """[rows, cols] -> [cols, rows padded to a multiple of `pad` with zeros] (a GEMM operand: the reduction runs over rows);
        half: an fp16 operand copy (rows padded to 64: the 16-bit kernels' K-tile)."""
        rows, cols = x.shape
        if self.half:
            rp = (rows + 63) // 64 * 64
            out = (torch.zeros if rp != rows else torch.empty)((cols, rp), dtype=torch.float16, device=self.dev)
            L.check(self.lib.sprc_transpose_f32_to16(x.data_ptr(), x.stride(0), out.data_ptr(), rp, rows, cols, L.SPRC_F16, _st()), "sprc_transpose_f32_to16")
            return out
        rp = (rows + pad - 1) // pad * pad
        out = self.zeros(cols, rp) if rp != rows else self.empty(cols, rp)
        L.check(self.lib.sprc_transpose_f32(x.data_ptr(), x.stride(0), out.data_ptr(), rp, rows, cols, _st()), "sprc_transpose_f32")
        return out

    def colsum(self, x, out, accumulate=True):
        L.check(self.lib.sprc_colsum_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), int(accumulate), _st()), "sprc_colsum_f32")

    def gelu(self, x):
        y = torch.empty_like(x)
        L.check(self.lib.sprc_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _st()), "sprc_gelu_fwd")
        return y

    def gelu_bwd(self, x, dy):
        dx = torch.empty_like(x)
        L.check(self.lib.sprc_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _st()), "sprc_gelu_bwd")
        return dx

    def ln(self, x, gamma, beta, eps):
        y, _ = E.layernorm(x, gamma, beta, eps, F32, want32=True, want16=False)
        return y

    def ln_bwd(self, x, gamma, dy, eps, dgamma, dbeta, need_dx=True):
        M, D = x.shape
        need = int(self.lib.sprc_layernorm_bwd_workspace_bytes(M, D))
        ws = self.ws("ln", need)
        dx = torch.empty_like(x) if need_dx else None
        L.check(self.lib.sprc_layernorm_bwd(x.data_ptr(), x.stride(0), gamma.data_ptr(), dy.data_ptr(), dy.stride(0), float(eps), M, D,
                                            None if dx is None else dx.data_ptr(), D, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                                            ws.numel(), _st()), "sprc_layernorm_bwd")
        return dx

    def attention(self, q, k, v, B, H, Tq, Tk, mask, scale, drop=None):
        return E.attention(q, k, v, B, H, Tq, Tk, 64, q.stride(0), k.stride(0), v.stride(0), scale, key_mask=mask, drop=drop)

    def dropout(self, x, drop, resid=None):
        """x * keep / (1 - p) (+ resid); drop = (p, seed, site) or None = identity (then x + resid when resid is given)"""
        if drop is None:
            return x if resid is None else x + resid
        y = torch.empty_like(x)
        p, seed, site = drop
        L.check(self.lib.sprc_dropout_f32(x.data_ptr(), None if resid is None else resid.data_ptr(), y.data_ptr(), x.numel(), seed, site, p, _st()),
                "sprc_dropout_f32")
        return y

    def attention_bwd(self, q, k, v, dout, B, H, Tq, Tk, mask, scale, drop=None):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        scratch = self.ws("attn", 2 * B * H * Tq * Tk * 4)
        a = L.AttentionBwdArgs()
        a.B, a.H, a.Tq, a.Tk, a.head_dim = B, H, Tq, Tk, 64
        a.q, a.k, a.v, a.dout = q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr()
        a.ldq, a.ldk, a.ldv, a.lddo = q.stride(0), k.stride(0), v.stride(0), dout.stride(0)
        a.key_mask, a.scale = (None if mask is None else mask.data_ptr()), scale
        a.dq, a.dk, a.dv, a.lddq, a.lddk, a.lddv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dq.stride(0), dk.stride(0), dv.stride(0)
        a.scratch, a.scratch_bytes = scratch.data_ptr(), scratch.numel()
        if drop is not None:
            a.drop_p, a.drop_seed, a.drop_site = drop
        L.check(self.lib.sprc_attention_bwd(C.byref(a), _st()), "sprc_attention_bwd")
        return dq, dk, dv


class _Linear:
    """y = x W^T + b with gradient accumulators; W^T is formed once per step and shared by every use of the layer."""

    def __init__(self, k: _K, P, G, wname: str, bname: Optional[str]):
        self.k, self.W, self.b = k, P[wname], (P[bname] if bname else None)
        self.gW, self.gb = G[wname], (G[bname] if bname else None)
        self._Wt = None
        self._W16 = None                         # half mode: the step's fp16 copy of the fp32 master weight

    def fwd(self, x, resid=None):
        W = self.W
        if self.k.half:
            if self._W16 is None:
                self._W16 = torch.empty(W.shape, dtype=torch.float16, device=W.device)
                L.check(self.k.lib.sprc_cast_f32_to_16(W.data_ptr(), self._W16.data_ptr(), W.numel(), L.SPRC_F16, _st()), "sprc_cast_f32_to_16")
            W = self._W16
        return self.k.gemm(x, W, bias=self.b, resid=resid)

    def bwd(self, x, dy, acc=None, need_dx=True, shared=0):
        """dW += dy^T x, db += colsum(dy); returns dx (+ acc: the running gradient of x from other paths) or None.
        shared: x is read by several linears of this step (1: cache its transpose, 2: this is its last consumer)."""
        k = self.k
        assert self.W.shape[0] % 32 == 0 or not need_dx, "dX = dY . W reduces over the N output features: the fp32 GEMM needs N % 32 == 0"
        xt = k.transpose_saved(x, last=shared == 2) if shared else k.transpose(x)
        k.gemm(k.transpose(dy), xt, resid=self.gW, out=self.gW)
        if self.gb is not None:
            k.colsum(dy, self.gb)
        if not need_dx:
            return None
        if self._Wt is None:
            self._Wt = k.transpose(self.W, pad=32)           # [K, N]: dX = dY . W reduces over the N output features
        Wt = self._Wt[:, :self.W.shape[0]] if self._Wt.shape[1] != self.W.shape[0] else self._Wt
        return k.gemm(dy, Wt, resid=acc, out=acc)


# dropout sites (the numbering of oracle/sprc_oracle.py: drop_site): pass * 256 + layer * 8 + kind, embeddings pass * 256 + 255
DROP_SELF_P, DROP_SELF_OUT, DROP_CROSS_P, DROP_CROSS_OUT, DROP_FFN_Q, DROP_FFN_T, DROP_EMB = 0, 1, 2, 3, 4, 5, 255


class TrainStep:
    def __init__(self, cfg, params: Dict[str, torch.Tensor], engine: E.Engine, dropout_p: float = 0.0, seed: int = 0, products: str = "fp32"):
        """products: "fp32" -- every product of the trainable part on the exact-fp32 MFMA (the parity mode: gradients within 1.5e-5 of the
        reference's fp32 forward + backward); "fp16" -- on fp16 operand copies with fp32 accumulation, fp32 outputs and fp32 master weights:
        the reference's training arithmetic (blip_fine_tune_2.py:290-303: fp16 autocast under GradScaler; here only the OPERANDS are rounded,
        the reference also rounds every linear's output).  Gradients of very small magnitude need the GradScaler's loss scale, as there."""
        if products not in ("fp32", "fp16"):
            raise ValueError(f"TrainStep products {products!r}")
        # `engine` runs the FROZEN ViT trunk only (vit_forward + its pre-ln_vision stream): fp32, or fp16 as under the reference's autocast
        # (blip_fine_tune_2.py:293); everything that trains -- ln_vision, the Q-Former, the heads -- is computed here on the exact-fp32 GEMM
        if engine.dt not in (L.SPRC_F32, L.SPRC_F16) or engine.fp8:
            raise L.SprcError("the training step's frozen trunk runs on an fp32 or fp16 engine")
        self.cfg, self.P, self.eng = cfg, params, engine
        if not 0.0 <= dropout_p < 1.0:
            raise ValueError("dropout_p must be in [0, 1)")
        self.drop_p, self.seed = float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF
        self.dev = engine.device
        self.k = _K(self.dev, half=products == "fp16")
        q = cfg.qformer
        self.Hd, self.H, self.Lq, self.Lt, self.eps = q.hidden, q.heads, q.num_query, cfg.max_txt_len, q.ln_eps
        self.sc = 1.0 / (q.head_dim ** 0.5)
        assert q.head_dim == 64, "sprc_attention_bwd is written for the Q-Former's head_dim 64"
        self.trainable = [n for n in params if self._trains(n)]

    @staticmethod
    def _trains(name: str) -> bool:
        return not (name.startswith("visual_encoder.") or name.startswith("itm_head.") or name.startswith("Qformer.cls.")
                    or name.endswith("position_ids"))

    def _drop(self, pass_id: int, layer: int, kind: int):
        """(p, seed, site) of one dropout call, or None in eval mode"""
        if self.drop_p <= 0.0:
            return None
        return (self.drop_p, self.seed, pass_id * 256 + (DROP_EMB if kind == DROP_EMB else layer * 8 + kind))

    # ---- building blocks with saved context -------------------------------------------------------------------------
    def _lin(self, w, b=None):
        key = w
        if key not in self._lins:
            self._lins[key] = _Linear(self.k, self.P, self.G, w, b)
        return self._lins[key]

    def _attn_fwd(self, pre, xq, xkv, B, Sq, Sk, mask, dp=None, do=None):
        """BertSelfAttention + BertSelfOutput (Qformer.py:175-295): LN(dropout(dense(dropout(softmax(q k^T / 8 + mask)) v)) + xq).
        xq [B*Sq, Hd], xkv [B*Sk, Dk]; dp / do: the dropout calls on the probabilities (:264) and on the dense output (:293)."""
        k = self.k
        lq, lk, lv = self._lin(pre + "self.query.weight", pre + "self.query.bias"), self._lin(pre + "self.key.weight", pre + "self.key.bias"), \
            self._lin(pre + "self.value.weight", pre + "self.value.bias")
        lo = self._lin(pre + "output.dense.weight", pre + "output.dense.bias")
        q, kk, v = lq.fwd(xq), lk.fwd(xkv), lv.fwd(xkv)
        ctx = k.attention(q, kk, v, B, self.H, Sq, Sk, mask, self.sc, drop=dp)
        t = lo.fwd(ctx, resid=xq) if do is None else k.dropout(lo.fwd(ctx), do, resid=xq)
        gn, bn = pre + "output.LayerNorm.weight", pre + "output.LayerNorm.bias"
        y = k.ln(t, self.P[gn], self.P[bn], self.eps)
        return y, dict(pre=pre, xq=xq, xkv=xkv, q=q, k=kk, v=v, ctx=ctx, t=t, B=B, Sq=Sq, Sk=Sk, mask=mask, same=xkv is xq, dp=dp, do=do)

    def _attn_bwd(self, c, dy, dkv_acc=None):
        """-> dxq; the key / value source gets its gradient added into dkv_acc (cross-attention) or into dxq (self-attention)."""
        k, pre = self.k, c["pre"]
        gn, bn = pre + "output.LayerNorm.weight", pre + "output.LayerNorm.bias"
        dt = k.ln_bwd(c["t"], self.P[gn], dy, self.eps, self.G[gn], self.G[bn])
        dctx = self._lin(pre + "output.dense.weight").bwd(c["ctx"], dt if c["do"] is None else k.dropout(dt, c["do"]))
        dq, dk, dv = k.attention_bwd(c["q"], c["k"], c["v"], dctx, c["B"], self.H, c["Sq"], c["Sk"], c["mask"], self.sc, drop=c["dp"])
        same = c["xq"] is c["xkv"]                                                  # self-attention: q / k / v read one input
        dxq = self._lin(pre + "self.query.weight").bwd(c["xq"], dq, acc=dt, shared=1 if same else 0)          # residual path + query path
        tgt = dxq if c["same"] else dkv_acc
        need = tgt is not None
        r = self._lin(pre + "self.key.weight").bwd(c["xkv"], dk, acc=tgt, need_dx=need, shared=1)
        r = self._lin(pre + "self.value.weight").bwd(c["xkv"], dv, acc=r if need else None, need_dx=need, shared=2)
        return r if c["same"] else dxq

    def _ffn_fwd(self, pre_i, pre_o, x, do=None):
        """LN(dropout(W2 gelu(W1 x)) + x)  (Qformer.py:482-490, dropout :379)"""
        k = self.k
        z = self._lin(pre_i + "dense.weight", pre_i + "dense.bias").fwd(x)
        h = k.gelu(z)
        lo = self._lin(pre_o + "dense.weight", pre_o + "dense.bias")
        t = lo.fwd(h, resid=x) if do is None else k.dropout(lo.fwd(h), do, resid=x)
        y = k.ln(t, self.P[pre_o + "LayerNorm.weight"], self.P[pre_o + "LayerNorm.bias"], self.eps)
        return y, dict(pre_i=pre_i, pre_o=pre_o, x=x, z=z, h=h, t=t, do=do)

    def _ffn_bwd(self, c, dy):
        k = self.k
        gn, bn = c["pre_o"] + "LayerNorm.weight", c["pre_o"] + "LayerNorm.bias"
        dt = k.ln_bwd(c["t"], self.P[gn], dy, self.eps, self.G[gn], self.G[bn])
        dh = self._lin(c["pre_o"] + "dense.weight").bwd(c["h"], dt if c["do"] is None else k.dropout(dt, c["do"]))
        dz = k.gelu_bwd(c["z"], dh)
        return self._lin(c["pre_i"] + "dense.weight").bwd(c["x"], dz, acc=dt)

    def _rows(self, x3, lo, hi):
        """[B, S, Hd] -> contiguous [B * (hi - lo), Hd] (a copy: data movement only)"""
        return x3[:, lo:hi, :].reshape(-1, x3.shape[-1]).contiguous()

    def _stack_fwd(self, x, B, S, mask, enc, Tenc, pass_id=0):
        """12 BertLayers (Qformer.py:408-480) over x [B*S, Hd]; enc [B*Tenc, Dv] or None; pass_id numbers the dropout sites."""
        Hd, Lq = self.Hd, self.Lq
        ctxs: List[dict] = []
        D = lambda l, kind: self._drop(pass_id, l, kind)                      # noqa: E731
        for l in range(self.cfg.qformer.layers):
            b = f"Qformer.bert.encoder.layer.{l}."
            a, ca = self._attn_fwd(b + "attention.", x, x, B, S, S, mask, D(l, DROP_SELF_P), D(l, DROP_SELF_OUT))
            c = dict(self_attn=ca, S=S)
            if enc is not None:
                a3 = a.view(B, S, Hd)
                qa = self._rows(a3, 0, Lq) if S > Lq else a
                if l % self.cfg.qformer.cross_freq == 0:
                    qa, c["cross"] = self._attn_fwd(b + "crossattention.", qa, enc, B, Lq, Tenc, None, D(l, DROP_CROSS_P), D(l, DROP_CROSS_OUT))
                oq, c["ffn_q"] = self._ffn_fwd(b + "intermediate_query.", b + "output_query.", qa, D(l, DROP_FFN_Q))
                if S > Lq:
                    ot, c["ffn_t"] = self._ffn_fwd(b + "intermediate.", b + "output.", self._rows(a3, Lq, S), D(l, DROP_FFN_T))
                    x = torch.cat([oq.view(B, Lq, Hd), ot.view(B, S - Lq, Hd)], dim=1).reshape(B * S, Hd)
                else:
                    x = oq
            else:
                x, c["ffn_t"] = self._ffn_fwd(b + "intermediate.", b + "output.", a, D(l, DROP_FFN_T))
            ctxs.append(c)
        return x, ctxs

    def _stack_bwd(self, ctxs, dx, B, denc):
        Hd, Lq = self.Hd, self.Lq
        for c in reversed(ctxs):
            S = c["S"]
            if "ffn_q" in c:
                d3 = dx.view(B, S, Hd)
                dqa = self._ffn_bwd(c["ffn_q"], self._rows(d3, 0, Lq) if S > Lq else dx)
                if "cross" in c:
                    dqa = self._attn_bwd(c["cross"], dqa, dkv_acc=denc)
                if S > Lq:
                    dat = self._ffn_bwd(c["ffn_t"], self._rows(d3, Lq, S))
                    da = torch.cat([dqa.view(B, Lq, Hd), dat.view(B, S - Lq, Hd)], dim=1).reshape(B * S, Hd)
                else:
                    da = dqa
            else:
                da = self._ffn_bwd(c["ffn_t"], dx)
            dx = self._attn_bwd(c["self_attn"], da)
        return dx

    def _embed_args(self, B, Lq, Lt, query, q_bstride, ids, no_img):
        a = L.QformerEmbedArgs()
        p = "Qformer.bert.embeddings."
        a.B, a.Lq, a.Lt, a.hidden, a.out_dtype, a.vocab = B, Lq, Lt, self.Hd, F32, self.cfg.qformer.vocab
        a.query_embeds, a.q_bstride = (None if query is None else query.data_ptr()), q_bstride
        a.input_ids = None if ids is None else ids.data_ptr()
        a.word_emb, a.pos_emb = self.P[p + "word_embeddings.weight"].data_ptr(), self.P[p + "position_embeddings.weight"].data_ptr()
        a.gamma, a.beta, a.eps = self.P[p + "LayerNorm.weight"].data_ptr(), self.P[p + "LayerNorm.bias"].data_ptr(), self.eps
        a.no_img = int(no_img)
        return a

    def _embed_fwd(self, B, Lq, Lt, query, q_bstride, ids, no_img=False, pass_id=0):
        """BertEmbeddings.forward (Qformer.py:78-114): rows -> LayerNorm -> dropout (:113).  query: [.., Lq, Hd] fp32 (q_bstride 0 = shared)."""
        p = "Qformer.bert.embeddings."
        a = self._embed_args(B, Lq, Lt, query, q_bstride, ids, no_img)
        pre = self.k.empty(B * (Lq + Lt), self.Hd)
        L.check(self.k.lib.sprc_qformer_embed_rows(C.byref(a), pre.data_ptr(), _st()), "sprc_qformer_embed_rows")
        y = self.k.ln(pre, self.P[p + "LayerNorm.weight"], self.P[p + "LayerNorm.bias"], self.eps)
        do = self._drop(pass_id, 0, DROP_EMB)
        if do is not None:
            y = self.k.dropout(y, do)
        return y, dict(args=a, pre=pre, keep=(query, ids), do=do)

    def _embed_bwd(self, c, dy, dquery, dq_bstride):
        p = "Qformer.bert.embeddings."
        if c["do"] is not None:
            dy = self.k.dropout(dy, c["do"])
        dpre = self.k.ln_bwd(c["pre"], self.P[p + "LayerNorm.weight"], dy, self.eps, self.G[p + "LayerNorm.weight"], self.G[p + "LayerNorm.bias"])
        L.check(self.k.lib.sprc_qformer_embed_bwd(C.byref(c["args"]), dpre.data_ptr(), None if dquery is None else dquery.data_ptr(), dq_bstride,
                                                  self.G[p + "word_embeddings.weight"].data_ptr(), self.G[p + "position_embeddings.weight"].data_ptr(),
                                                  _st()), "sprc_qformer_embed_bwd")

    def _head_fwd(self, wname, x):
        """normalize(proj(x)) -> (unit rows, context)"""
        z = self._lin(wname + ".weight", wname + ".bias").fwd(x)
        y = torch.empty_like(z)
        L.check(self.k.lib.sprc_l2norm_rows(z.data_ptr(), z.stride(0), y.data_ptr(), None, y.stride(0), z.shape[0], z.shape[1], F32, _st()), "sprc_l2norm_rows")
        return y, dict(w=wname, x=x, z=z)

    def _head_bwd(self, c, dy):
        z = c["z"]
        dz = torch.empty_like(z)
        L.check(self.k.lib.sprc_l2norm_bwd(z.data_ptr(), z.stride(0), dy.data_ptr(), dy.stride(0), dz.data_ptr(), dz.stride(0), z.shape[0], z.shape[1], _st()),
                "sprc_l2norm_bwd")
        return self._lin(c["w"] + ".weight").bwd(c["x"], dz)

    # ---- the step -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, image, target, input_ids, attention_mask) -> Dict[str, torch.Tensor]:
        k, P, eng = self.k, self.P, self.eng
        B = image.shape[0]
        Hd, Lq, Lt = self.Hd, self.Lq, self.Lt
        S = Lq + Lt
        self.G = {n: torch.zeros_like(P[n]) for n in self.trainable}
        self._lins: Dict[str, _Linear] = {}
        self.k._xt.clear()
        ids = input_ids.to(device=self.dev, dtype=torch.int64).contiguous()
        am = attention_mask.to(device=self.dev, dtype=torch.int64).contiguous()
        # frozen ViT (align_prompt.py:64-69) + ln_vision; its INPUT is kept: ln_vision trains
        T, Dv = self.cfg.vit.tokens, self.cfg.vit.width
        self.pre_ref, self.pre_tgt = k.empty(B * T, Dv), k.empty(B * T, Dv)
        # (the engine's own ln_vision output is not used: its packed copy of ln_vision is stale once the optimizer has stepped)
        eng.vit_forward(image, pre_ln_out=self.pre_ref)
        eng.vit_forward(target, pre_ln_out=self.pre_tgt)
        raw_ref = k.ln(self.pre_ref, P["ln_vision.weight"], P["ln_vision.bias"], self.cfg.ln_vision_eps)
        raw_tgt = k.ln(self.pre_tgt, P["ln_vision.weight"], P["ln_vision.bias"], self.cfg.ln_vision_eps)
        mask = k.empty(B, S)
        L.check(k.lib.sprc_qformer_mask(am.data_ptr(), mask.data_ptr(), B, Lq, Lt, _st()), "sprc_qformer_mask")
        qt = P["query_tokens"].view(Lq, Hd)
        # P1: fusion pass 1 (:120-127)      P2: pass 2 on its query rows (:129-134)
        x1, self.e1 = self._embed_fwd(B, Lq, Lt, qt, 0, ids, pass_id=0)
        h1, self.c1 = self._stack_fwd(x1, B, S, mask, raw_ref, T, pass_id=0)
        x2, self.e2 = self._embed_fwd(B, Lq, Lt, h1, S * Hd, ids, pass_id=1)
        h2, self.c2 = self._stack_fwd(x2, B, S, mask, None, 0, pass_id=1)
        fusion, self.hf = self._head_fwd("text_proj", h2.view(B, S, Hd)[:, Lq, :].contiguous())
        # P3: target image pass (:141-155)
        x3, self.e3 = self._embed_fwd(B, Lq, 0, qt, 0, None, pass_id=2)
        h3, self.c3 = self._stack_fwd(x3, B, Lq, None, raw_tgt, T, pass_id=2)
        tfeat, self.ht = self._head_fwd("vision_proj", h3)
        # P4: text-only prompt pass (:170-179)
        pt = P["prompt_tokens"].view(Lq, Hd)
        x4, self.e4 = self._embed_fwd(B, Lq, Lt, pt, 0, ids, no_img=True, pass_id=3)
        h4, self.c4 = self._stack_fwd(x4, B, S, mask, None, 0, pass_id=3)
        tonly, self.ho = self._head_fwd("text_proj", h4.view(B, S, Hd)[:, 0, :].contiguous())
        # losses (:157-167, :181-193)
        temp = float(P["temp"])
        tf3 = tfeat.view(B, Lq, -1).contiguous()
        self.sim = torch.empty((2, B, B), dtype=torch.float32, device=self.dev)
        E.sim_max(fusion, tf3, out=self.sim[0])
        E.sim_max(tonly, tf3, out=self.sim[1])
        losses = k.zeros(3)
        for i in range(2):
            L.check(k.lib.sprc_contrastive_ce(self.sim[i].data_ptr(), B, B, temp, losses[i:].data_ptr(), _st()), "sprc_contrastive_ce")
        L.check(k.lib.sprc_align_mse(h1.data_ptr(), S * Hd, Lq, Hd, pt.data_ptr(), B, losses[2:].data_ptr(), _st()), "sprc_align_mse")
        self.saved = dict(B=B, S=S, T=T, ids=ids, mask=mask, h1=h1, fusion=fusion, tonly=tonly, tf3=tf3, temp=temp, pt=pt,
                          raw_ref=raw_ref, raw_tgt=raw_tgt)
        return {"loss_itc": losses[0], "loss_rtc": losses[1], "loss_align": losses[2]}

    @torch.no_grad()
    def backward(self, weights: Dict[str, float]) -> Dict[str, torch.Tensor]:
        """Gradient of sum_k weights[k] * loss_k with respect to every trainable tensor (call once per forward)."""
        k, G, s = self.k, self.G, self.saved
        B, S, T, Hd, Lq = s["B"], s["S"], s["T"], self.Hd, self.Lq
        E_ = s["fusion"].shape[1]
        lib = k.lib
        dfusion, dtonly, dtf3 = k.zeros(B, E_), k.zeros(B, E_), k.zeros(B, Lq, E_)
        jstar = torch.empty((B, B), dtype=torch.int32, device=self.dev)
        dsim = k.empty(B, B)
        dtemp = k.zeros(1)
        for i, (name, q_feat, dq) in enumerate((("loss_itc", s["fusion"], dfusion), ("loss_rtc", s["tonly"], dtonly))):
            L.check(lib.sprc_contrastive_ce_bwd(self.sim[i].data_ptr(), B, B, s["temp"], float(weights.get(name, 0.0)), dsim.data_ptr(), dtemp.data_ptr(),
                                                _st()), "sprc_contrastive_ce_bwd")
            L.check(lib.sprc_sim_max_bwd(q_feat.data_ptr(), s["tf3"].data_ptr(), dsim.data_ptr(), B, B, Lq, E_, dq.data_ptr(), dtf3.data_ptr(),
                                         jstar.data_ptr(), _st()), "sprc_sim_max_bwd")
        G["temp"].copy_(dtemp.view(()))
        denc_ref, denc_tgt = k.zeros(B * T, self.cfg.vit.width), k.zeros(B * T, self.cfg.vit.width)
        # P4: text-only prompt pass
        dh4 = k.zeros(B, S, Hd)
        dh4[:, 0, :] = self._head_bwd(self.ho, dtonly)
        dx4 = self._stack_bwd(self.c4, dh4.view(B * S, Hd), B, None)
        self._embed_bwd(self.e4, dx4, G["prompt_tokens"], 0)
        # P3: target image pass
        dh3 = self._head_bwd(self.ht, dtf3.view(B * Lq, E_))
        dx3 = self._stack_bwd(self.c3, dh3, B, denc_tgt)
        self._embed_bwd(self.e3, dx3, G["query_tokens"], 0)
        # P2 -> P1: pass 2's query rows ARE pass 1's output rows; loss_align adds to the same rows
        dh2 = k.zeros(B, S, Hd)
        dh2[:, Lq, :] = self._head_bwd(self.hf, dfusion)
        dx2 = self._stack_bwd(self.c2, dh2.view(B * S, Hd), B, None)
        dh1 = k.zeros(B, S, Hd)
        self._embed_bwd(self.e2, dx2, dh1, S * Hd)
        L.check(lib.sprc_align_mse_bwd(s["h1"].data_ptr(), S * Hd, Lq, Hd, s["pt"].data_ptr(), B, float(weights.get("loss_align", 0.0)),
                                       dh1.data_ptr(), S * Hd, _st()), "sprc_align_mse_bwd")
        dx1 = self._stack_bwd(self.c1, dh1.view(B * S, Hd), B, denc_ref)
        self._embed_bwd(self.e1, dx1, G["query_tokens"], 0)
        # ln_vision (blip2.py:81,193-199) trains: image tokens = LN(frozen ViT output)
        gv, bv = self.P["ln_vision.weight"], G["ln_vision.bias"]
        for pre, d in ((self.pre_ref, denc_ref), (self.pre_tgt, denc_tgt)):
            k.ln_bwd(pre, gv, d, self.cfg.ln_vision_eps, G["ln_vision.weight"], bv, need_dx=False)
        k._xt.clear()
        return G
