"""Deterministic synthetic checkpoints and inputs.

The reference's weights (eva_vit_g.pth, blip2_pretrained.pth, sprc_cirr.pt) are
network fetches (models/eva_vit.py:442-446, configs/models/blip2/blip2_pretrain.yaml:10,
README.md:123-128) and are not available offline, so parity and benchmark runs use
seeded random state dicts with the reference's own key names and shapes
(SURVEY.md section 8(b)).  Each tensor is drawn from its own torch CPU generator
seeded by crc32(name) so that any subset of tensors can be produced independently
and identically in the golden-vector script, the oracle and the HIP engine.

Synthetic inputs follow SURVEY.md section 8(d): images ~ N(0,1) (the distribution of
CLIP-normalised pixels), captions as ``input_ids[nq,32]`` with [CLS]=101 at column 0,
uniform ids in [1000,30000), length U{4..32}, [SEP]=102 at the last real column,
0 padding; the reference image of query i is gallery index (7919*i) mod N.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, List, Tuple

import torch

from .config import SprcConfig

Spec = Tuple[str, Tuple[int, ...], str]


def _vit_specs(cfg: SprcConfig) -> List[Spec]:
    v = cfg.vit
    D, F, T = v.width, v.mlp, v.tokens
    p = "visual_encoder."
    out: List[Spec] = []
    if v.kind == "eva_g":
        out += [(p + "cls_token", (1, 1, D), "emb"), (p + "pos_embed", (1, T, D), "emb"),
                (p + "patch_embed.proj.weight", (D, 3, v.patch, v.patch), "w"),
                (p + "patch_embed.proj.bias", (D,), "b")]
        for i in range(v.depth):
            b = f"{p}blocks.{i}."
            out += [(b + "norm1.weight", (D,), "ln_w"), (b + "norm1.bias", (D,), "ln_b"),
                    (b + "attn.q_bias", (D,), "b"), (b + "attn.v_bias", (D,), "b"),
                    (b + "attn.qkv.weight", (3 * D, D), "w"),
                    (b + "attn.proj.weight", (D, D), f"w_res{i + 1}"), (b + "attn.proj.bias", (D,), "b"),
                    (b + "norm2.weight", (D,), "ln_w"), (b + "norm2.bias", (D,), "ln_b"),
                    (b + "mlp.fc1.weight", (F, D), "w"), (b + "mlp.fc1.bias", (F,), "b"),
                    (b + "mlp.fc2.weight", (D, F), f"w_res{i + 1}"), (b + "mlp.fc2.bias", (D,), "b")]
    elif v.kind == "clip_L":
        out += [(p + "class_embedding", (D,), "emb"), (p + "positional_embedding", (T, D), "emb"),
                (p + "conv1.weight", (D, 3, v.patch, v.patch), "w"),
                (p + "ln_pre.weight", (D,), "ln_w"), (p + "ln_pre.bias", (D,), "ln_b")]
        for i in range(v.depth):
            b = f"{p}transformer.resblocks.{i}."
            out += [(b + "ln_1.weight", (D,), "ln_w"), (b + "ln_1.bias", (D,), "ln_b"),
                    (b + "attn.in_proj_weight", (3 * D, D), "w"), (b + "attn.in_proj_bias", (3 * D,), "b"),
                    (b + "attn.out_proj.weight", (D, D), f"w_res{i + 1}"), (b + "attn.out_proj.bias", (D,), "b"),
                    (b + "ln_2.weight", (D,), "ln_w"), (b + "ln_2.bias", (D,), "ln_b"),
                    (b + "mlp.c_fc.weight", (F, D), "w"), (b + "mlp.c_fc.bias", (F,), "b"),
                    (b + "mlp.c_proj.weight", (D, F), f"w_res{i + 1}"), (b + "mlp.c_proj.bias", (D,), "b")]
    else:
        raise ValueError(v.kind)
    return out


def _qformer_specs(cfg: SprcConfig) -> List[Spec]:
    q = cfg.qformer
    H, F, Dv = q.hidden, q.ffn, cfg.vit.width
    p = "Qformer.bert."
    out: List[Spec] = [
        (p + "embeddings.word_embeddings.weight", (q.vocab, H), "emb"),
        (p + "embeddings.position_embeddings.weight", (q.max_pos, H), "emb"),
        (p + "embeddings.LayerNorm.weight", (H,), "ln_w"), (p + "embeddings.LayerNorm.bias", (H,), "ln_b"),
    ]

    def attn(prefix: str, kv_in: int) -> List[Spec]:
        return [(prefix + "self.query.weight", (H, H), "w"), (prefix + "self.query.bias", (H,), "b"),
                (prefix + "self.key.weight", (H, kv_in), "w"), (prefix + "self.key.bias", (H,), "b"),
                (prefix + "self.value.weight", (H, kv_in), "w"), (prefix + "self.value.bias", (H,), "b"),
                (prefix + "output.dense.weight", (H, H), "w"), (prefix + "output.dense.bias", (H,), "b"),
                (prefix + "output.LayerNorm.weight", (H,), "ln_w"), (prefix + "output.LayerNorm.bias", (H,), "ln_b")]

    def ffn(prefix_i: str, prefix_o: str) -> List[Spec]:
        return [(prefix_i + "dense.weight", (F, H), "w"), (prefix_i + "dense.bias", (F,), "b"),
                (prefix_o + "dense.weight", (H, F), "w"), (prefix_o + "dense.bias", (H,), "b"),
                (prefix_o + "LayerNorm.weight", (H,), "ln_w"), (prefix_o + "LayerNorm.bias", (H,), "ln_b")]

    for l in range(q.layers):
        b = f"{p}encoder.layer.{l}."
        out += attn(b + "attention.", H)
        if l % q.cross_freq == 0:                       # Qformer.py:392-399
            out += attn(b + "crossattention.", Dv)
        out += ffn(b + "intermediate.", b + "output.")
        out += ffn(b + "intermediate_query.", b + "output_query.")
    return out


def param_specs(cfg: SprcConfig) -> List[Spec]:
    """(state-dict key, shape, init kind) for every tensor the retrieval path reads."""
    H, E, Dv = cfg.qformer.hidden, cfg.embed_dim, cfg.vit.width
    out = _vit_specs(cfg)
    out += [("ln_vision.weight", (Dv,), "ln_w"), ("ln_vision.bias", (Dv,), "ln_b"),
            ("query_tokens", (1, cfg.qformer.num_query, H), "emb"),
            ("prompt_tokens", (1, cfg.qformer.num_query, H), "emb")]
    out += _qformer_specs(cfg)
    out += [("vision_proj.weight", (E, H), "w_head"), ("vision_proj.bias", (E,), "b"),
            ("text_proj.weight", (E, H), "w_head"), ("text_proj.bias", (E,), "b"),
            # image-text-matching head: unused by `inference` (align_prompt.py:92), the classifier of the stage-2 rerank
            # (blip2_qformer_cir_rerank.py:88, :441-445)
            ("itm_head.weight", (2, H), "w_head"), ("itm_head.bias", (2,), "b")]
    return out


def _draw(name: str, shape: Tuple[int, ...], kind: str, seed: int, device: str) -> torch.Tensor:
    if device == "cpu":
        g = torch.Generator(device="cpu")
        g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        x = torch.randn(shape, generator=g, dtype=torch.float32)
    else:  # throughput runs only: fast on-device fill, not reproducible against the CPU draw
        g = torch.Generator(device=device)
        g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        x = torch.randn(shape, generator=g, dtype=torch.float32, device=device)
    if kind == "w" or kind == "emb":
        return x.mul_(0.02)
    if kind == "w_head":
        return x.mul_(0.05)
    if kind.startswith("w_res"):           # eva_vit.py:297-303 rescales proj/fc2 by 1/sqrt(2*layer_id)
        return x.mul_(0.02 / (2.0 * int(kind[5:])) ** 0.5)
    if kind == "b" or kind == "ln_b":
        return x.mul_(0.02)
    if kind == "ln_w":
        return x.mul_(0.1).add_(1.0)
    raise ValueError(kind)


def iter_state_dict(cfg: SprcConfig, seed: int = 0, device: str = "cpu") -> Iterator[Tuple[str, torch.Tensor]]:
    for name, shape, kind in param_specs(cfg):
        yield name, _draw(name, shape, kind, seed, device)


PLANT_RANK = 8          # rank of the planted ITC heads: cosines live in an 8-dim subspace -> scores spread over ~ +-0.8


def plant_structure(sd: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Planted-structure variant of a random state dict.  Plain random-init weights give nearly
    uniform similarities (0.02 .. 0.09, neighbour gaps of 1e-4: Recall@K is decided by noise).  Here
      * vision_proj / text_proj share a rank-8 output subspace (W = U A with U [256,8] orthonormal), so the max-cosine
        scores of a gallery row spread over more than 1.0 with a median neighbour gap of 1e-2;
      * the paths that carry image content and text into the features are amplified MILDLY (cross-attention x2, word
        embeddings x2).  Larger gains (x6 cross-attention, x3 on every Q-Former linear) saturate the softmaxes and make
        the map chaotic: a 2e-7 relative perturbation of the weights then moves scores by 3e-2, so "the reference's
        order" stops being defined at fp32 precision; with these gains the same perturbation moves them by 2.5e-6
        (plain random weights: 4e-7), measured with the oracle.
    In place; CPU tensors only (parity runs)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(777 + seed)
    E, H = sd["vision_proj.weight"].shape
    U = torch.linalg.qr(torch.randn((E, PLANT_RANK), generator=g, dtype=torch.float32))[0]
    for k in list(sd):
        if not k.startswith("Qformer.") or not k.endswith("weight") or "LayerNorm" in k:
            continue
        if "crossattention" in k or "word_embeddings" in k:
            sd[k] = sd[k] * 2.0
    sd["vision_proj.weight"] = (U @ (torch.randn((PLANT_RANK, H), generator=g) * 0.2)).contiguous()
    sd["text_proj.weight"] = (U @ (torch.randn((PLANT_RANK, H), generator=g) * 0.2)).contiguous()
    sd["vision_proj.bias"] = sd["vision_proj.bias"] * 0.1
    sd["text_proj.bias"] = sd["text_proj.bias"] * 0.1
    return sd


# tensors of the trunk that the reference's convert_weights_to_fp16 (eva_vit.py:410-425; clip_vit.py:12 imports the same function)
# turns into fp16: weight and bias of every nn.Conv2d / nn.Linear MODULE.  EVA's q_bias / v_bias and nn.MultiheadAttention's
# in_proj_* are bare Parameters and stay fp32, like cls / pos embeddings and the LayerNorms.
_TRUNK_FP16_SUFFIXES = ("patch_embed.proj.weight", "patch_embed.proj.bias", "attn.qkv.weight", "attn.proj.weight", "attn.proj.bias",
                        "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias",
                        "conv1.weight", "attn.out_proj.weight", "attn.out_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                        "mlp.c_proj.weight", "mlp.c_proj.bias")


def round_trunk_to_fp16(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The state dict as a GPU-trained reference checkpoint holds it ("ViT Linear/Conv weights are fp16 in
    GPU-trained checkpoints"; the released SPRC checkpoint is the full state_dict() of a model built with vit_precision="fp16",
    README.md:123-128, utils.py:219-222): the trunk's Conv / Linear tensors take fp16 VALUES (kept as fp32 tensors here, which is
    what the reference's CPU path makes of them: models/__init__.py:246-247 `model.float()`).  In place."""
    for k in sd:
        if k.startswith("visual_encoder.") and k.endswith(_TRUNK_FP16_SUFFIXES):
            sd[k] = sd[k].to(torch.float16).to(torch.float32)
    return sd


def make_state_dict(cfg: SprcConfig, seed: int = 0, device: str = "cpu", planted: bool = False, trunk_fp16: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded random state dict with the reference's key names (fp32); planted=True: see `plant_structure`; trunk_fp16=True:
    the trunk's Conv / Linear tensors hold fp16-representable values, as in a GPU-trained checkpoint (`round_trunk_to_fp16`)."""
    sd = dict(iter_state_dict(cfg, seed, device))
    sd["temp"] = torch.tensor(0.07, device=device)          # align_prompt.py:84 (unused by inference)
    if planted:
        if device != "cpu":
            raise ValueError("planted weights are drawn on the CPU (reproducible against the golden fixtures)")
        plant_structure(sd, seed)
    if trunk_fp16:
        round_trunk_to_fp16(sd)
    return sd


def make_images(n: int, seed: int = 0, image: int = 224, planted: bool = False) -> torch.Tensor:
    """N(0,1) pixels; planted=True: 0.8 * (random mixture of 8 basis images) + 0.4 * noise, so that images differ along a
    few shared directions instead of being 150k-dimensional white noise."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    if not planted:
        return torch.randn((n, 3, image, image), generator=g, dtype=torch.float32)
    basis = torch.randn((8, 3, image, image), generator=g, dtype=torch.float32)
    coef = torch.randn((n, 8), generator=g, dtype=torch.float32)
    noise = torch.randn((n, 3, image, image), generator=g, dtype=torch.float32)
    return torch.einsum("nk,kchw->nchw", coef, basis).mul_(0.8).add_(noise.mul_(0.4))


def make_queries(nq: int, n_gallery: int, seed: int = 1, max_len: int = 32,
                 vocab_lo: int = 1000, vocab_hi: int = 30000):
    """Synthetic tokenised captions + the gallery index of each query's reference image."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ids = torch.randint(vocab_lo, vocab_hi, (nq, max_len), generator=g, dtype=torch.int64)
    lens = torch.randint(4, max_len + 1, (nq,), generator=g, dtype=torch.int64)
    col = torch.arange(max_len).unsqueeze(0)
    mask = (col < lens.unsqueeze(1)).to(torch.int64)
    ids = ids * mask
    ids[:, 0] = 101
    ids[torch.arange(nq), lens - 1] = 102
    ref_index = (7919 * torch.arange(nq, dtype=torch.int64)) % max(n_gallery, 1)
    return ids, mask, ref_index
