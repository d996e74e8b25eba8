"""CPU oracle for the SPRC retrieval hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain fp32 restatement (torch CPU ops for the floating-point stages, numpy for the
integer ranking / metric stages) of the reference algorithm, written from the
reference's behaviour; every function cites the reference lines it follows (paths
relative to /root/reference/src).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path (sprc_amd/) never does and
fails loudly if its HIP library is missing.

Pinning: the reference is Python, so it is imported and executed in the build
container by oracle/gen_golden.py (shims listed there), and this restatement is checked
against the outputs it produced (tests/golden/*.npz, tests/test_oracle_golden.py).
R9 (WordPiece tokenisation) lives in the third-party `transformers` package
(requirements.txt:9 pins 4.36.2; vocab `bert-base-uncased` is a network fetch,
lavis/models/blip2_models/blip2.py:32) -> tokeniser parity is pinned only against the
installed transformers implementation on a synthetic vocabulary; "parity unpinned"
against the real vocabulary.

State-dict keys are the reference's own.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# R3: EVA ViT-g trunk  (lavis/models/eva_vit.py)
# --------------------------------------------------------------------------------------
def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def eva_vit_forward(sd: SD, cfg, image: Tensor, prefix: str = "visual_encoder.",
                    taps: Optional[dict] = None) -> Tensor:
    """`VisionTransformer.forward_features` eva_vit.py:324-340; Block :173-180; Attention :117-148."""
    v = cfg.vit
    p = prefix
    B = image.shape[0]
    # PatchEmbed eva_vit.py:196,203: conv 14x14 stride 14 (+bias) -> flatten(2).transpose(1,2)
    x = F.conv2d(image.float(), sd[p + "patch_embed.proj.weight"].float(),
                 sd[p + "patch_embed.proj.bias"].float(), stride=v.patch)
    x = x.flatten(2).transpose(1, 2)
    # cls token + absolute position embedding :328-331
    x = torch.cat([sd[p + "cls_token"].float().expand(B, -1, -1), x], dim=1) + sd[p + "pos_embed"].float()
    if taps is not None:
        taps["patch_embed"] = x.clone()
    H, dh = v.heads, v.head_dim
    scale = dh ** -0.5                                                       # :74
    for i in range(v.depth):
        b = f"{p}blocks.{i}."
        h = _ln(x, sd[b + "norm1.weight"], sd[b + "norm1.bias"], v.ln_eps)   # eps 1e-6 :439
        qkv_bias = torch.cat([sd[b + "attn.q_bias"].float(),
                              torch.zeros_like(sd[b + "attn.v_bias"]).float(),
                              sd[b + "attn.v_bias"].float()])                # :120-122 (k bias == 0)
        qkv = F.linear(h, sd[b + "attn.qkv.weight"].float(), qkv_bias)
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)             # :125
        q, k, vv = qkv[0] * scale, qkv[1], qkv[2]                            # :128
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                     # :129,142
        ctx = (attn @ vv).transpose(1, 2).reshape(B, T, H * dh)              # :145
        x = x + F.linear(ctx, sd[b + "attn.proj.weight"].float(), sd[b + "attn.proj.bias"].float())
        h = _ln(x, sd[b + "norm2.weight"], sd[b + "norm2.bias"], v.ln_eps)
        h = F.gelu(F.linear(h, sd[b + "mlp.fc1.weight"].float(), sd[b + "mlp.fc1.bias"].float()))  # erf GELU :45
        x = x + F.linear(h, sd[b + "mlp.fc2.weight"].float(), sd[b + "mlp.fc2.bias"].float())
        if taps is not None and i == 0:
            taps["block0"] = x.clone()
    return x                                                                 # no final norm (:341-347 commented)


# --------------------------------------------------------------------------------------
# R3L: CLIP ViT-L trunk  (lavis/models/clip_vit.py)
# --------------------------------------------------------------------------------------
def clip_vit_forward(sd: SD, cfg, image: Tensor, prefix: str = "visual_encoder.",
                     taps: Optional[dict] = None) -> Tensor:
    """`VisionTransformer.forward` clip_vit.py:171-185; ResidualAttentionBlock :132-139."""
    v = cfg.vit
    p = prefix
    B = image.shape[0]
    x = F.conv2d(image.float(), sd[p + "conv1.weight"].float(), None, stride=v.patch)   # no bias :160
    x = x.reshape(B, v.width, -1).permute(0, 2, 1)
    cls = sd[p + "class_embedding"].float().reshape(1, 1, -1).expand(B, -1, -1)
    x = torch.cat([cls, x], dim=1) + sd[p + "positional_embedding"].float()              # :176-177
    x = _ln(x, sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"], v.ln_eps)                 # :178
    if taps is not None:
        taps["patch_embed"] = x.clone()
    H, dh = v.heads, v.head_dim
    for i in range(v.depth):
        b = f"{p}transformer.resblocks.{i}."
        h = _ln(x, sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], v.ln_eps)
        qkv = F.linear(h, sd[b + "attn.in_proj_weight"].float(), sd[b + "attn.in_proj_bias"].float())
        T = qkv.shape[1]
        q, k, vv = qkv.split(v.width, dim=-1)                                # packed [q;k;v] nn.MultiheadAttention
        q = q.reshape(B, T, H, dh).transpose(1, 2) * (dh ** -0.5)
        k = k.reshape(B, T, H, dh).transpose(1, 2)
        vv = vv.reshape(B, T, H, dh).transpose(1, 2)
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        ctx = (attn @ vv).transpose(1, 2).reshape(B, T, H * dh)
        x = x + F.linear(ctx, sd[b + "attn.out_proj.weight"].float(), sd[b + "attn.out_proj.bias"].float())
        h = _ln(x, sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], v.ln_eps)
        h = F.linear(h, sd[b + "mlp.c_fc.weight"].float(), sd[b + "mlp.c_fc.bias"].float())
        h = h * torch.sigmoid(1.702 * h)                                     # QuickGELU :109-111
        x = x + F.linear(h, sd[b + "mlp.c_proj.weight"].float(), sd[b + "mlp.c_proj.bias"].float())
        if taps is not None and i == 0:
            taps["block0"] = x.clone()
    return x                                                                 # no ln_final (:184)


def vit_forward(sd: SD, cfg, image: Tensor, taps: Optional[dict] = None) -> Tensor:
    if cfg.vit.kind == "eva_g":
        return eva_vit_forward(sd, cfg, image, taps=taps)
    return clip_vit_forward(sd, cfg, image, taps=taps)


def encode_image_tokens(sd: SD, cfg, image: Tensor, taps: Optional[dict] = None) -> Tensor:
    """`ln_vision(visual_encoder(image)).float()` align_prompt.py:366-368; LayerNorm blip2.py:193-199."""
    x = vit_forward(sd, cfg, image, taps=taps)
    return _ln(x, sd["ln_vision.weight"], sd["ln_vision.bias"], cfg.ln_vision_eps)


# --------------------------------------------------------------------------------------
# R5: Q-Former  (lavis/models/blip2_models/Qformer.py)
# --------------------------------------------------------------------------------------
# ---- training-mode dropout (Qformer.py:113,264,293,379 under blip_fine_tune_2.py:290 `.train()`) with REPRODUCIBLE masks ----------------
# nn.Dropout draws from torch's global generator; a golden needs masks both sides can regenerate.  The mask of element i of dropout
# site `site` under `seed` is a counter-based hash (the same integer arithmetic as csrc/common.hpp: drop_keep), injected into the
# reference by oracle/gen_golden.py and regenerated by the HIP kernels.
DROP_SELF_P, DROP_SELF_OUT, DROP_CROSS_P, DROP_CROSS_OUT, DROP_FFN_Q, DROP_FFN_T, DROP_EMB = 0, 1, 2, 3, 4, 5, 255


def drop_site(pass_id: int, layer: int, kind: int) -> int:
    """site id of a dropout call: pass 0 = fusion pass 1, 1 = pass 2, 2 = target-image pass, 3 = text-only prompt pass (the order of
    align_prompt.py:120-179); kind = DROP_*; the embedding dropout of a pass is (pass, 0, DROP_EMB)."""
    return pass_id * 256 + (DROP_EMB if kind == DROP_EMB else layer * 8 + kind)


def drop_keep(seed: int, site: int, n: int, p: float) -> np.ndarray:
    """keep mask [n] (bool) of dropout site `site`: z = seed + site * 0x9E3779B97F4A7C15 + i * 0xD1B54A32D192ED03 (mod 2^64),
    SplitMix64 finaliser, keep <=> (z >> 32) >= floor(fl32(p) * 2^32) -- p is taken as the FLOAT32 the C ABI carries (sprc.h: drop_p is a
    float; csrc/common.hpp: drop_thresh), so that both sides form the same threshold: 0.1 -> 429496736 (the double 0.1 gives 429496729)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed % (1 << 64)) + np.uint64(site) * np.uint64(0x9E3779B97F4A7C15) + np.arange(n, dtype=np.uint64) * np.uint64(0xD1B54A32D192ED03)
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(32)) >= np.uint64(int(float(np.float32(p)) * 4294967296.0))


def dropout(x: Tensor, drop, site: int) -> Tensor:
    """x * keep / (1 - p) with the reproducible mask; drop = None (eval) or (seed, p, pass_id)."""
    if drop is None or drop[1] <= 0.0:
        return x
    seed, p = drop[0], drop[1]
    keep = torch.from_numpy(drop_keep(seed, site, x.numel(), p)).view(x.shape)
    return x * keep.to(x.dtype) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))      # the kernels' fp32 1 / (1 - p)


def _bert_attention(sd: SD, pre: str, x_q: Tensor, x_kv: Tensor, add_mask: Optional[Tensor],
                    heads: int, eps: float, drop=None, layer: int = 0, cross: bool = False) -> Tensor:
    """BertSelfAttention.forward Qformer.py:175-281 + BertSelfOutput :291-295 (drop None = eval: dropout = identity)."""
    B, Sq, Hd = x_q.shape
    dh = Hd // heads
    q = F.linear(x_q, sd[pre + "self.query.weight"].float(), sd[pre + "self.query.bias"].float())
    k = F.linear(x_kv, sd[pre + "self.key.weight"].float(), sd[pre + "self.key.bias"].float())
    v = F.linear(x_kv, sd[pre + "self.value.weight"].float(), sd[pre + "self.value.bias"].float())
    q = q.view(B, Sq, heads, dh).permute(0, 2, 1, 3)
    k = k.view(B, -1, heads, dh).permute(0, 2, 1, 3)
    v = v.view(B, -1, heads, dh).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)                 # :250
    if add_mask is not None:
        s = s + add_mask                                                     # :253
    pr = torch.softmax(s, dim=-1)
    if drop is not None:
        pr = dropout(pr, drop, drop_site(drop[2], layer, DROP_CROSS_P if cross else DROP_SELF_P))            # :264
    ctx = torch.matmul(pr, v).permute(0, 2, 1, 3).contiguous().view(B, Sq, Hd)
    out = F.linear(ctx, sd[pre + "output.dense.weight"].float(), sd[pre + "output.dense.bias"].float())
    if drop is not None:
        out = dropout(out, drop, drop_site(drop[2], layer, DROP_CROSS_OUT if cross else DROP_SELF_OUT))      # :293
    return _ln(out + x_q, sd[pre + "output.LayerNorm.weight"], sd[pre + "output.LayerNorm.bias"], eps)


def _bert_ffn(sd: SD, pre_i: str, pre_o: str, x: Tensor, eps: float, drop=None, layer: int = 0, query: bool = False) -> Tensor:
    """feed_forward_chunk(_query) Qformer.py:482-490: LN(dropout(W2 . GELU_erf(W1 x)) + x)."""
    h = F.gelu(F.linear(x, sd[pre_i + "dense.weight"].float(), sd[pre_i + "dense.bias"].float()))
    h = F.linear(h, sd[pre_o + "dense.weight"].float(), sd[pre_o + "dense.bias"].float())
    if drop is not None:
        h = dropout(h, drop, drop_site(drop[2], layer, DROP_FFN_Q if query else DROP_FFN_T))                 # :379
    return _ln(h + x, sd[pre_o + "LayerNorm.weight"], sd[pre_o + "LayerNorm.bias"], eps)


def qformer_forward(sd: SD, cfg, query_embeds: Tensor, input_ids: Optional[Tensor] = None,
                    attention_mask: Optional[Tensor] = None, encoder_hidden_states: Optional[Tensor] = None,
                    taps: Optional[dict] = None, drop=None) -> Tensor:
    """`BertModel.forward` Qformer.py:810-973 in the three call shapes of the retrieval path.

    (i)  image-only: input_ids None, encoder_hidden_states given  (align_prompt.py:376-381)
    (ii) fusion pass 1: ids + mask[B,64] + encoder_hidden_states  (align_prompt.py:332-339)
    (iii) pass 2: ids + mask, NO encoder states -> no cross-attention, text FFN on all rows
          (align_prompt.py:341-346; Qformer.py:434-435,469-475)
    """
    qc = cfg.qformer
    p = "Qformer.bert."
    Lq = query_embeds.shape[1]
    # BertEmbeddings.forward :98-114 -- text rows get word+position (positions from 0), query rows
    # are used as given; ONE LayerNorm over all rows (including the query rows).
    if input_ids is not None:
        S = input_ids.shape[1]
        emb = sd[p + "embeddings.word_embeddings.weight"].float()[input_ids] \
            + sd[p + "embeddings.position_embeddings.weight"].float()[:S].unsqueeze(0)
        emb = torch.cat([query_embeds.float(), emb], dim=1)
    else:
        emb = query_embeds.float()
    x = _ln(emb, sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], qc.ln_eps)
    if drop is not None:
        x = dropout(x, drop, drop_site(drop[2], 0, DROP_EMB))                                                # :113
    B, S_all, _ = x.shape
    if attention_mask is None:
        attention_mask = torch.ones((B, S_all))                              # :887-890
    add_mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0   # :806-807
    if taps is not None:
        taps["emb"] = x.clone()
    for l in range(qc.layers):
        b = f"{p}encoder.layer.{l}."
        a = _bert_attention(sd, b + "attention.", x, x, add_mask, qc.heads, qc.ln_eps, drop, l)
        if encoder_hidden_states is not None:                                # :434
            qa = a[:, :Lq, :]
            if l % qc.cross_freq == 0:                                       # :392-399, :438-450
                # encoder mask is all ones -> additive 0 (:925-934)
                qa = _bert_attention(sd, b + "crossattention.", qa, encoder_hidden_states.float(), None,
                                     qc.heads, qc.ln_eps, drop, l, cross=True)
            out = _bert_ffn(sd, b + "intermediate_query.", b + "output_query.", qa, qc.ln_eps, drop, l, query=True)
            if a.shape[1] > Lq:                                              # :461-468
                out_t = _bert_ffn(sd, b + "intermediate.", b + "output.", a[:, Lq:, :], qc.ln_eps, drop, l)
                out = torch.cat([out, out_t], dim=1)
        else:                                                                # :469-475
            out = _bert_ffn(sd, b + "intermediate.", b + "output.", a, qc.ln_eps, drop, l)
        x = out
        if taps is not None and l == 0:
            taps["layer0"] = x.clone()
    return x


# --------------------------------------------------------------------------------------
# R2 / R6: model protocol
# --------------------------------------------------------------------------------------
def _normalize(x: Tensor) -> Tensor:
    return F.normalize(x, dim=-1)                                            # x / max(||x||, 1e-12)


def extract_target_features(sd: SD, cfg, image: Tensor, taps: Optional[dict] = None, drop=None) -> Tuple[Tensor, Tensor]:
    """`Blip2QformerCirAlignPrompt.extract_target_features` align_prompt.py:364-386 (CPU path: all fp32)."""
    raw = encode_image_tokens(sd, cfg, image, taps=taps)
    B = raw.shape[0]
    q = qformer_forward(sd, cfg, sd["query_tokens"].float().expand(B, -1, -1), encoder_hidden_states=raw, drop=drop)
    feats = _normalize(F.linear(q, sd["vision_proj.weight"].float(), sd["vision_proj.bias"].float()))
    return feats, raw


def fuse_queries(sd: SD, cfg, reference_embeds: Tensor, input_ids: Tensor, attention_mask: Tensor,
                 taps: Optional[dict] = None) -> Tensor:
    """The fusion half of `inference` align_prompt.py:313-350 -> fusion_feats [B,256]."""
    B = reference_embeds.shape[0]
    Lq = cfg.qformer.num_query
    qt = sd["query_tokens"].float().expand(B, -1, -1)
    mask = torch.cat([torch.ones((B, Lq), dtype=attention_mask.dtype), attention_mask], dim=1)   # :331
    p1 = qformer_forward(sd, cfg, qt, input_ids, mask, encoder_hidden_states=reference_embeds)
    p2 = qformer_forward(sd, cfg, p1[:, :Lq, :], input_ids, mask)
    if taps is not None:
        taps["pass1"], taps["pass2"] = p1.clone(), p2.clone()
    return _normalize(F.linear(p2[:, 32, :], sd["text_proj.weight"].float(), sd["text_proj.bias"].float()))  # :348-350


def similarity(fusion: Tensor, target_feats: Tensor) -> Tensor:
    """sim[b,n] = max_j <fusion_b, target_feats[n,j,:]>  (align_prompt.py:353-358), as a plain GEMM."""
    N, J, E = target_feats.shape
    s = fusion.float() @ target_feats.float().reshape(N * J, E).t()          # [B, N*J]
    return s.view(fusion.shape[0], N, J).max(dim=-1).values


def inference(sd: SD, cfg, reference_embeds: Tensor, target_feats: Tensor, input_ids: Tensor,
              attention_mask: Tensor, taps: Optional[dict] = None) -> Tensor:
    """`Blip2QformerCirAlignPrompt.inference` align_prompt.py:312-361 with pre-tokenised text."""
    fusion = fuse_queries(sd, cfg, reference_embeds, input_ids, attention_mask, taps=taps)
    if taps is not None:
        taps["fusion"] = fusion.clone()
    return similarity(fusion, target_feats)


def text_feature(sd: SD, cfg, input_ids: Tensor, attention_mask: Tensor) -> Tensor:
    """The query side of `Blip2QformerCirRerank.inference` (blip2_qformer_cir_rerank.py:373-390): `Qformer.bert(ids, mask)` with no
    query tokens and no image -- word + position embeddings (positions from 0, Qformer.py:93-105), LayerNorm, 12 layers of
    self-attention + TEXT FFN on every row (:469-475) -- then normalize(text_proj(h[:, 0, :]))."""
    qc = cfg.qformer
    p = "Qformer.bert."
    S = input_ids.shape[1]
    emb = sd[p + "embeddings.word_embeddings.weight"].float()[input_ids] \
        + sd[p + "embeddings.position_embeddings.weight"].float()[:S].unsqueeze(0)
    x = _ln(emb, sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], qc.ln_eps)
    add_mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
    for l in range(qc.layers):
        b = f"{p}encoder.layer.{l}."
        a = _bert_attention(sd, b + "attention.", x, x, add_mask, qc.heads, qc.ln_eps)
        x = _bert_ffn(sd, b + "intermediate.", b + "output.", a, qc.ln_eps)
    return _normalize(F.linear(x[:, 0, :], sd["text_proj.weight"].float(), sd["text_proj.bias"].float()))


def inference_rerank_stage1(sd: SD, cfg, target_feats: Tensor, input_ids: Tensor, attention_mask: Tensor) -> Tensor:
    """`Blip2QformerCirRerank.inference` (:373-397): sim[b, n] = max_j <text_feature_b, target_feats[n, j]>."""
    return similarity(text_feature(sd, cfg, input_ids, attention_mask), target_feats)


def qformer_text_only(sd: SD, cfg, prompt_embeds: Tensor, input_ids: Tensor, attention_mask: Tensor, drop=None) -> Tensor:
    """`BertModel.forward(..., no_img=True)` (Qformer.py:88-104, used by align_prompt.py:173-179): the embedding rows are
    [text[0] ([CLS]) ; the 32 prompt rows ; text[1:]], EVERY row gets its absolute position (0..63), one LayerNorm; no
    encoder states -> no cross-attention, text FFN on all rows.  attention_mask [B,64] is used as given."""
    qc = cfg.qformer
    p = "Qformer.bert."
    we = sd[p + "embeddings.word_embeddings.weight"].float()[input_ids]                       # [B,32,H]
    emb = torch.cat([we[:, :1, :], prompt_embeds.float(), we[:, 1:, :]], dim=1)
    emb = emb + sd[p + "embeddings.position_embeddings.weight"].float()[:emb.shape[1]].unsqueeze(0)
    x = _ln(emb, sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], qc.ln_eps)
    if drop is not None:
        x = dropout(x, drop, drop_site(drop[2], 0, DROP_EMB))
    add_mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
    for l in range(qc.layers):
        b = f"{p}encoder.layer.{l}."
        a = _bert_attention(sd, b + "attention.", x, x, add_mask, qc.heads, qc.ln_eps, drop, l)
        x = _bert_ffn(sd, b + "intermediate.", b + "output.", a, qc.ln_eps, drop, l)
    return x


def training_losses(sd: SD, cfg, image: Tensor, target: Tensor, input_ids: Tensor, attention_mask: Tensor,
                    drop: Optional[Tuple[int, float]] = None) -> Dict[str, Tensor]:
    """`Blip2QformerCirAlignPrompt.forward` align_prompt.py:95-200, pre-tokenised text: loss_itc (fusion -> target contrastive),
    loss_rtc (text-only prompt -> target contrastive), loss_align (MSE between the mean fused query token and the mean prompt token).
    drop None: eval mode (dropout = identity); drop = (seed, p): train mode as blip_fine_tune_2.py:290 runs it, the Q-Former's dropout
    (p = 0.1 in the reference) with the reproducible masks of `drop_keep` (the ViT stays in eval: align_prompt.py:67-68)."""
    dr = (lambda pass_id: None) if drop is None else (lambda pass_id: (drop[0], drop[1], pass_id))
    B = image.shape[0]
    Lq = cfg.qformer.num_query
    temp = sd["temp"].float() if "temp" in sd else torch.tensor(0.07)
    raw = encode_image_tokens(sd, cfg, image)
    qt = sd["query_tokens"].float().expand(B, -1, -1)
    mask = torch.cat([torch.ones((B, Lq), dtype=attention_mask.dtype), attention_mask], dim=1)
    p1 = qformer_forward(sd, cfg, qt, input_ids, mask, encoder_hidden_states=raw, drop=dr(0))             # :120-127
    p2 = qformer_forward(sd, cfg, p1[:, :Lq, :], input_ids, mask, drop=dr(1))                             # :129-134
    fusion = _normalize(F.linear(p2[:, 32, :], sd["text_proj.weight"].float(), sd["text_proj.bias"].float()))
    target_feats, _ = extract_target_features(sd, cfg, target, drop=dr(2))                    # :141-155
    targets = torch.arange(B)
    loss_itc = F.cross_entropy(similarity(fusion, target_feats) / temp, targets)              # :157-167
    prompt = sd["prompt_tokens"].float().expand(B, -1, -1)
    t_only = qformer_text_only(sd, cfg, prompt, input_ids, mask, drop=dr(3))                  # :170-179
    t_feat = _normalize(F.linear(t_only[:, 0, :], sd["text_proj.weight"].float(), sd["text_proj.bias"].float()))
    loss_rtc = F.cross_entropy(similarity(t_feat, target_feats) / temp, targets)              # :181-190
    loss_align = F.mse_loss(p1[:, :Lq, :].mean(1), prompt.clone().detach().mean(1))           # :192-193 (the prompt side is detached)
    return {"loss_itc": loss_itc, "loss_rtc": loss_rtc, "loss_align": loss_align}


TRAIN_LOSS_WEIGHTS = {"loss_itc": 1.0, "loss_rtc": 0.4, "loss_align": 0.4}    # blip_fine_tune_2.py:293-299 with its defaults (:379-380)


def training_gradients(sd: SD, cfg, image: Tensor, target: Tensor, input_ids: Tensor, attention_mask: Tensor,
                       weights: Optional[Dict[str, float]] = None, drop: Optional[Tuple[int, float]] = None) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """-> (losses, {name: d(loss_itc + w_rtc loss_rtc + w_align loss_align) / d tensor}) for every tensor the reference trains
    (blip_fine_tune_2.py:257-262: everything with requires_grad, i.e. all but the ViT trunk, align_prompt.py:64-69): torch autograd
    over `training_losses`, the restatement of `forward` above.  Tensors `forward` does not touch (itm_head, the LM head) get none."""
    weights = weights or TRAIN_LOSS_WEIGHTS
    leaf = {k: (v.detach().float().clone().requires_grad_(not k.startswith("visual_encoder.")) if v.is_floating_point() else v)
            for k, v in sd.items()}
    losses = training_losses(leaf, cfg, image, target, input_ids, attention_mask, drop=drop)
    total = sum(weights[k] * v for k, v in losses.items())
    total.backward()
    grads = {k: v.grad for k, v in leaf.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    return {k: v.detach() for k, v in losses.items()}, grads


def inference_rerank(sd: SD, cfg, reference_embeds: Tensor, target_embeds: Tensor, input_ids: Tensor,
                     attention_mask: Tensor) -> Tensor:
    """`Blip2QformerCirRerank.inference_rerank` blip2_qformer_cir_rerank.py:399-445 with pre-tokenised text (N2).

    reference_embeds [B,257,D]; target_embeds [B*T,257,D] (T candidates per query, query-major); ids/mask [B,32].
    Every (query, candidate) pair runs the Q-Former ONCE in call shape (ii) with the 514 encoder tokens
    cat(reference, candidate) (:430-437); itm_head on the 32 query rows, mean over them, softmax over the two
    classes, probability of class 1 ("match") -> [B*T]."""
    B, BT = reference_embeds.shape[0], target_embeds.shape[0]
    T = BT // B if B > 1 else BT                                             # :404-407
    ref = reference_embeds.repeat_interleave(T, dim=0)                       # 'b l d -> (b t) l d'
    ids = input_ids.repeat_interleave(T, dim=0)
    atts = attention_mask.repeat_interleave(T, dim=0)
    Lq = cfg.qformer.num_query
    qt = sd["query_tokens"].float().expand(BT, -1, -1)
    mask = torch.cat([torch.ones((BT, Lq), dtype=atts.dtype), atts], dim=1)  # :424
    h = qformer_forward(sd, cfg, qt, ids, mask, encoder_hidden_states=torch.cat([ref.float(), target_embeds.float()], dim=1))
    vl = F.linear(h[:, :Lq, :], sd["itm_head.weight"].float(), sd["itm_head.bias"].float())      # :440-441
    return torch.softmax(vl.mean(dim=1), dim=-1)[:, -1]                      # :442-445


# --------------------------------------------------------------------------------------
# R7: ranking + metrics (integer work, numpy)
# --------------------------------------------------------------------------------------
def distances(sim: np.ndarray) -> np.ndarray:
    """`distances = 1 - pred_sim` in fp32 (validate_blip.py:253, :44; cirr_test_submission.py:82)."""
    return (np.float32(1.0) - np.asarray(sim, dtype=np.float32)).astype(np.float32)


def rank_stable(sim: np.ndarray) -> np.ndarray:
    """The ranking contract: sort by key (fl32(1 - sim), index) ascending.

    torch.argsort on the reference path (validate_blip.py:254) is not stable; SURVEY.md
    section 7 ("tie semantics") fixes the contract as the stable order, which is one of the
    orders the reference may produce.
    """
    return np.argsort(distances(sim), axis=-1, kind="stable").astype(np.int64)


def topk_stable(sim: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    idx = rank_stable(sim)[:, :k]
    return np.take_along_axis(np.asarray(sim, dtype=np.float32), idx, axis=1), idx


def rank_of(sim: np.ndarray, listed: np.ndarray) -> np.ndarray:
    """rank_of[q,l] = #{n : (d[q,n], n) < (d[q,listed[q,l]], listed[q,l])}  (position in rank_stable)."""
    d = distances(sim)
    nq, N = d.shape
    out = np.zeros(listed.shape, dtype=np.int64)
    ar = np.arange(N)
    for q in range(nq):
        for l in range(listed.shape[1]):
            t = int(listed[q, l])
            if t < 0:
                out[q, l] = -1
                continue
            out[q, l] = int(np.sum((d[q] < d[q, t]) | ((d[q] == d[q, t]) & (ar < t))))
    return out


def cirr_metrics(sim: np.ndarray, ref_idx: np.ndarray, tgt_idx: np.ndarray, group_idx: np.ndarray) -> Tuple[float, ...]:
    """`compute_cirr_val_metrics` validate_blip.py:253-285 on gallery indices instead of names.

    ref_idx/tgt_idx [nq]; group_idx [nq, G] (the 6 subset members incl. reference and target).
    Returns (group_recall@1,2,3, recall@1,5,10,50) in percent, the reference's return order (:285).
    """
    order = rank_stable(sim)
    nq, N = order.shape
    keep = order != ref_idx[:, None]                                          # :258-261 drop the reference image
    order = order[keep].reshape(nq, N - 1)
    labels = order == tgt_idx[:, None]                                        # :264-265
    gmask = (order[..., None] == group_idx[:, None, :]).sum(-1).astype(bool)  # :268-271
    glabels = labels[gmask].reshape(nq, -1)
    assert (labels.sum(-1) == 1).all() and (glabels.sum(-1) == 1).all()       # :273-274

    def rec(l, k):  # 100 * torch.sum(labels[:, :k]) / len(labels), computed in fp32 like torch
        return float(np.float32(l[:, :k].sum()) / np.float32(len(l))) * 100

    return (rec(glabels, 1), rec(glabels, 2), rec(glabels, 3),
            rec(labels, 1), rec(labels, 5), rec(labels, 10), rec(labels, 50))


def fiq_metrics(sim: np.ndarray, tgt_idx: np.ndarray) -> Tuple[float, float]:
    """`compute_fiq_val_metrics` validate_blip.py:43-57 (reference image NOT removed)."""
    order = rank_stable(sim)
    labels = order == tgt_idx[:, None]
    assert (labels.sum(-1) == 1).all()                                        # :51
    r10 = float(np.float32(labels[:, :10].sum()) / np.float32(len(labels))) * 100
    r50 = float(np.float32(labels[:, :50].sum()) / np.float32(len(labels))) * 100
    return r10, r50


def cirr_test_dicts(sim: np.ndarray, ref_idx: np.ndarray, group_idx: np.ndarray, pair_ids: Sequence[int],
                    names: Sequence[str], rerank_scores: Optional[np.ndarray] = None):
    """`generate_cirr_test_dicts` cirr_test_submission.py:81-130.  rerank_scores [nq, top] (optional): the stage-2
    probabilities of the first `top` entries of every row, in stage-1 order; those entries are re-sorted by
    (fl32(1 - score), stage-1 position) before the reference image is removed (:88-112)."""
    order = rank_stable(sim)
    nq, N = order.shape
    if rerank_scores is not None:
        top = rerank_scores.shape[1]
        perm = np.argsort(distances(rerank_scores), axis=1, kind="stable")
        order = order.copy()
        order[:, :top] = np.take_along_axis(order[:, :top], perm, axis=1)
    order = order[order != ref_idx[:, None]].reshape(nq, N - 1)               # :116-120
    gmask = (order[..., None] == group_idx[:, None, :]).sum(-1).astype(bool)  # :122-124
    gorder = order[gmask].reshape(nq, -1)
    names = np.asarray(names)
    top = {str(int(p)): names[o[:50]].tolist() for p, o in zip(pair_ids, order)}      # :127-128
    sub = {str(int(p)): names[o[:3]].tolist() for p, o in zip(pair_ids, gorder)}      # :129-130
    return top, sub


# --------------------------------------------------------------------------------------
# R8 / R10: host-side caption handling
# --------------------------------------------------------------------------------------
def pre_caption(caption: str, max_words: int = 50) -> str:
    """`BlipCaptionProcessor.pre_caption` lavis/processors/blip_processors.py:49-68."""
    caption = re.sub(r"([.!\"()*#:;~])", " ", caption.lower())
    caption = re.sub(r"\s{2,}", " ", caption)
    caption = caption.rstrip("\n").strip(" ")
    words = caption.split(" ")
    if len(words) > max_words:
        caption = " ".join(words[:max_words])
    return caption


def fiq_caption(c1: str, c2: str) -> str:
    """validate_blip.py:180-184: '{Cap1} and {cap2}' before the text processor."""
    return f"{c1.strip('.?, ').capitalize()} and {c2.strip('.?, ')}"
