#!/usr/bin/env python3
"""(Not a test: a study script kept under tests/ because it imports the oracle, which only tests/, smoke() and bench.py's cpu_baseline may.)
Fake-quantisation study of the Q-Former image pass (CPU, torch): which 16-bit rounding sites of the engine's Q-Former carry the
feature error?  The fp32 restatement below follows oracle/sprc_oracle.py (qformer_forward, call shape (i)) with a rounding hook
q(site, tensor) at every place the 16-bit engine stores or reads a 16-bit value; one site class is switched on at a time.
    python tests/study_fq_qformer.py [fp16|bf16] [n_images]
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import sprc_oracle as O  # noqa: E402
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402

dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.set_num_threads(8)
cfg = get_config("pretrain", vit_depth=4)
sd = synth.make_state_dict(cfg, seed=0, planted=True)
images = synth.make_images(n_img, seed=0, planted=True)
with torch.no_grad():
    raw = O.encode_image_tokens(sd, cfg, images)
ACTIVE = set()


def q(site, t):
    return t.to(dt).float() if (site in ACTIVE or "all" in ACTIVE) else t


def lin(x, w, b, wsite):
    return F.linear(x, q(wsite, sd[w + ".weight"].float()), sd[w + ".bias"].float())


def ln(x, pre, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"].float(), sd[pre + ".bias"].float(), eps)


def attn(pre, x16, kv16, H, cross, l):
    B, S, D = x16.shape
    dh = D // H
    tag = "x" if cross else "s"
    qh = q(f"O_q{tag}", lin(x16, pre + "self.query", None, f"W_q{tag}")).view(B, S, H, dh).transpose(1, 2)
    kh = q(f"O_k{tag}", lin(kv16, pre + "self.key", None, f"W_k{tag}")).view(B, -1, H, dh).transpose(1, 2)
    vh = q(f"O_v{tag}", lin(kv16, pre + "self.value", None, f"W_v{tag}")).view(B, -1, H, dh).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, dim=-1)
    p = q(f"P_{tag}", p)
    ctx = q(f"O_ctx{tag}", (p @ vh).transpose(1, 2).reshape(B, S, D))
    return lin(ctx, pre + "output.dense", None, f"W_o{tag}")


def forward():
    qc = cfg.qformer
    p = "Qformer.bert."
    B = raw.shape[0]
    x = ln(sd["query_tokens"].float().expand(B, -1, -1), p + "embeddings.LayerNorm", qc.ln_eps)
    raw16 = q("A_raw", raw)
    for l in range(qc.layers):
        b = f"{p}encoder.layer.{l}."
        x16 = q("A_ln", x)
        a = ln(attn(b + "attention.", x16, x16, qc.heads, False, l) + x, b + "attention.output.LayerNorm", qc.ln_eps)
        if l % qc.cross_freq == 0:
            a16 = q("A_ln", a)
            a = ln(attn(b + "crossattention.", a16, raw16, qc.heads, True, l) + a, b + "crossattention.output.LayerNorm", qc.ln_eps)
        a16 = q("A_ln", a)
        h = q("O_ffn", F.gelu(lin(a16, b + "intermediate_query.dense", None, "W_f1")))
        x = ln(lin(h, b + "output_query.dense", None, "W_f2") + a, b + "output_query.LayerNorm", qc.ln_eps)
    f = lin(q("A_ln", x), "vision_proj", None, "W_head")
    return F.normalize(f, dim=-1)


SITES = ["A_ln", "A_raw", "W_qs", "W_ks", "W_vs", "W_os", "W_qx", "W_kx", "W_vx", "W_ox", "W_f1", "W_f2", "W_head",
         "O_qs", "O_ks", "O_vs", "P_s", "O_ctxs", "O_qx", "O_kx", "O_vx", "P_x", "O_ctxx", "O_ffn"]
with torch.no_grad():
    f0 = forward()
    chk, _ = O.extract_target_features(sd, cfg, images)
    print(f"restatement vs oracle: {float((f0 - chk).abs().max()):.1e}")
    tot = 0.0
    for s in SITES + ["all"]:
        ACTIVE.clear()
        ACTIVE.add(s)
        f = forward()
        e = float((f - f0).norm() / f0.norm())
        if s != "all":
            tot += e * e
        print(f"{s:8s} feats rel err {e:.2e}   max abs {float((f - f0).abs().max()):.2e}")
    print(f"quadrature sum of the single-site errors: {tot ** 0.5:.2e}")
    groups = {"weights": [s for s in SITES if s.startswith("W_")], "ln copies + raw": ["A_ln", "A_raw"],
              "self-attn q,k": ["O_qs", "O_ks"], "cross-attn q,k": ["O_qx", "O_kx"], "v, P, ctx": ["O_vs", "O_vx", "P_s", "P_x", "O_ctxs", "O_ctxx"],
              "ffn hidden": ["O_ffn"]}
    for name, ss in groups.items():
        ACTIVE.clear()
        ACTIVE.update(ss)
        f = forward()
        print(f"group {name:18s}: {float((f - f0).norm() / f0.norm()):.2e}")
    # what a split-operand (hi + lo fp16, ~22 bits) Q-Former leaves: GEMM operands exact, attention internals 16-bit
    for name, ss in {"split GEMM operands; attention q,k,v,P 16-bit": ["O_qs", "O_ks", "O_vs", "P_s", "O_qx", "O_kx", "O_vx", "P_x"],
                     "... and cross-attention q,k split too": ["O_qs", "O_ks", "O_vs", "P_s", "O_vx", "P_x"]}.items():
        ACTIVE.clear()
        ACTIVE.update(ss)
        f = forward()
        print(f"{name}: {float((f - f0).norm() / f0.norm()):.2e}  max abs {float((f - f0).abs().max()):.2e}")
