#!/usr/bin/env python3
"""Fake-quantisation study of the EVA ViT-g trunk (CPU, torch): which 16-bit rounding sites of the engine's ViT carry the error of
the ViT output?  The fp32 restatement follows oracle/sprc_oracle.py (eva_vit_forward) with a rounding hook q(site, tensor) at every
place the 16-bit engine stores or reads a 16-bit value (weights; LayerNorm outputs = GEMM operands; the qkv GEMM's output; the
softmax probabilities; the attention output; the GELU output); the residual stream, LayerNorm statistics, softmax and accumulation
stay fp32 as in the engine.  One site class at a time, then all.
    python tools/fq_vit.py [fp16|bf16] [n_images] [depth]        -> relative rms error of the ViT output (after ln_vision)
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 4
depth = int(sys.argv[3]) if len(sys.argv) > 3 else None
torch.set_num_threads(16)
cfg = get_config("pretrain", vit_depth=depth)
sd = synth.make_state_dict(cfg, seed=0, planted=True)
images = synth.make_images(n_img, seed=0, planted=True)
ACTIVE = set()


def q(site, t):
    return t.to(dt).float() if (site in ACTIVE or "all" in ACTIVE) else t


def ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w.float(), b.float(), eps)


@torch.no_grad()
def vit(image):
    v, p = cfg.vit, "visual_encoder."
    B = image.shape[0]
    x = F.conv2d(image.float(), q("w_patch", sd[p + "patch_embed.proj.weight"].float()), sd[p + "patch_embed.proj.bias"].float(), stride=v.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[p + "cls_token"].float().expand(B, -1, -1), x], dim=1) + sd[p + "pos_embed"].float()
    H, dh = v.heads, v.head_dim
    scale = dh ** -0.5
    for i in range(v.depth):
        b = f"{p}blocks.{i}."
        h = q("ln1_out", ln(x, sd[b + "norm1.weight"], sd[b + "norm1.bias"], v.ln_eps))
        qkv_bias = torch.cat([sd[b + "attn.q_bias"].float(), torch.zeros_like(sd[b + "attn.v_bias"]).float(), sd[b + "attn.v_bias"].float()])
        qkv = q("qkv_out", F.linear(h, q("w_qkv", sd[b + "attn.qkv.weight"].float()), qkv_bias))
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
        qq, k, vv = qkv[0], qkv[1], qkv[2]
        attn = q("probs", ((qq @ k.transpose(-2, -1)) * scale).softmax(dim=-1))
        ctx = q("ctx", (attn @ vv).transpose(1, 2).reshape(B, T, H * dh))
        x = x + F.linear(ctx, q("w_proj", sd[b + "attn.proj.weight"].float()), sd[b + "attn.proj.bias"].float())
        h = q("ln2_out", ln(x, sd[b + "norm2.weight"], sd[b + "norm2.bias"], v.ln_eps))
        h = q("gelu_out", F.gelu(F.linear(h, q("w_fc1", sd[b + "mlp.fc1.weight"].float()), sd[b + "mlp.fc1.bias"].float())))
        x = x + F.linear(h, q("w_fc2", sd[b + "mlp.fc2.weight"].float()), sd[b + "mlp.fc2.bias"].float())
    return ln(x, sd["ln_vision.weight"], sd["ln_vision.bias"], cfg.ln_vision_eps)


ref = vit(images)
SITES = ["ln1_out", "w_qkv", "qkv_out", "probs", "ctx", "w_proj", "ln2_out", "w_fc1", "gelu_out", "w_fc2", "w_patch"]
tot = 0.0
res = {}
for s in SITES + ["all"]:
    ACTIVE.clear()
    ACTIVE.add(s)
    out = vit(images)
    rel = float((out - ref).norm() / ref.norm())
    res[s] = rel
    if s != "all":
        tot += rel * rel
    print(f"{s:9s} rel-rms error of the ViT output {rel:.3e}", flush=True)
print(f"root of the sum of squares of the single sites {tot ** 0.5:.3e}; variance shares: " +
      ", ".join(f"{s} {100 * res[s] ** 2 / tot:.0f} %" for s in SITES))
groups = {"attention branch (ln1_out, w_qkv, qkv_out, probs, ctx, w_proj)": ["ln1_out", "w_qkv", "qkv_out", "probs", "ctx", "w_proj"],
          "MLP branch (ln2_out, w_fc1, gelu_out, w_fc2)": ["ln2_out", "w_fc1", "gelu_out", "w_fc2"],
          "weights": ["w_qkv", "w_proj", "w_fc1", "w_fc2", "w_patch"], "activations": ["ln1_out", "qkv_out", "probs", "ctx", "ln2_out", "gelu_out"]}
for name, ss in groups.items():
    print(f"  {name}: {100 * sum(res[s] ** 2 for s in ss) / tot:.0f} % of the variance")
