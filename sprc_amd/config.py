"""Model hyper-parameters of the SPRC retrieval path.

Every constant cites the reference line it restates (paths relative to
/root/reference/src/lavis):

* EVA ViT-g/14: models/eva_vit.py:428-441 (patch 14, width 1408, depth 39,
  16 heads x 88, mlp_ratio 4.3637 -> int(1408*4.3637) = 6144, LN eps 1e-6,
  exact-erf GELU, q/v bias only).
* CLIP ViT-L/14 trunk: models/clip_vit.py:242-250 (width 1024, 23 layers,
  16 heads x 64, MLP 4096, QuickGELU, ln_pre, conv without bias, LN eps 1e-5).
* Q-Former: models/blip2_models/blip2.py:46-61 + configs/models/bert_config.json
  (hidden 768, 12 layers, 12 heads x 64, FFN 3072, LN eps 1e-12, cross-attention
  every 2nd layer, 32 query tokens, vocab 30522 + "[DEC]" = 30523).
* ITC heads: models/blip2_models/blip2_qformer_cir_align_prompt.py:80-92
  (embed_dim 256, max_txt_len 32).
"""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class VitConfig:
    kind: str            # "eva_g" | "clip_L"
    width: int
    depth: int
    heads: int
    head_dim: int
    mlp: int
    act: str             # "gelu" (erf) | "quick_gelu"
    ln_eps: float
    patch: int = 14
    image: int = 224
    patch_bias: bool = True
    ln_pre: bool = False

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def patch_k(self) -> int:
        return 3 * self.patch * self.patch


@dataclass(frozen=True)
class QformerConfig:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    head_dim: int = 64
    ffn: int = 3072
    ln_eps: float = 1e-12
    vocab: int = 30523
    max_pos: int = 512
    num_query: int = 32
    cross_freq: int = 2


@dataclass(frozen=True)
class SprcConfig:
    vit: VitConfig
    qformer: QformerConfig
    embed_dim: int = 256
    max_txt_len: int = 32
    ln_vision_eps: float = 1e-5    # blip2.py:81,193-199 (nn.LayerNorm default)

    def with_depth(self, vit_depth: int | None = None, q_layers: int | None = None) -> "SprcConfig":
        vit = self.vit if vit_depth is None else replace(self.vit, depth=vit_depth)
        qf = self.qformer if q_layers is None else replace(self.qformer, layers=q_layers)
        return replace(self, vit=vit, qformer=qf)


EVA_G = VitConfig(kind="eva_g", width=1408, depth=39, heads=16, head_dim=88, mlp=6144,
                  act="gelu", ln_eps=1e-6, patch_bias=True, ln_pre=False)
CLIP_L = VitConfig(kind="clip_L", width=1024, depth=23, heads=16, head_dim=64, mlp=4096,
                   act="quick_gelu", ln_eps=1e-5, patch_bias=False, ln_pre=True)

# model_type names of load_model_and_preprocess (align_prompt.py:38-42)
MODEL_TYPES = {
    "pretrain": SprcConfig(vit=EVA_G, qformer=QformerConfig()),
    "pretrain_vitL": SprcConfig(vit=CLIP_L, qformer=QformerConfig()),
}


def get_config(model_type: str = "pretrain", vit_depth: int | None = None,
               q_layers: int | None = None) -> SprcConfig:
    if model_type not in MODEL_TYPES:
        raise AssertionError("Unknown model type {}".format(model_type))  # base_model.py:76-79
    return MODEL_TYPES[model_type].with_depth(vit_depth, q_layers)
