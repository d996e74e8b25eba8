#!/usr/bin/env python3
"""(Not a test: a measurement script kept under tests/ because its CPU leg times the oracle, which only tests/, smoke() and bench.py's
cpu_baseline may import.)  One training step of the reference's loop (blip_fine_tune_2.py:281-304: forward of the three losses, `scaler.scale(loss).backward()`,
AdamW) on the HIP training path (sprc_amd/train.py: fp32 kernels, full-depth frozen ViT + trainable Q-Former / heads / ln_vision),
timed on the GPU, next to the same step of the CPU oracle (torch autograd over oracle.training_losses) on a smaller batch.
    python tests/bench_train_step.py [batch=32] [steps=5] [cpu_batch=4] [fp32|fp16 trunk] [fp32|fp16 products]
SURVEY.md section 8(f) N4: measurement of the training row (not the headline metric)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sprc_amd import synth  # noqa: E402
from sprc_amd.config import get_config  # noqa: E402
from sprc_amd.model import Blip2QformerCirAlignPrompt  # noqa: E402
from sprc_amd.tokenizer import TokenBatch  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
CPU_B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
VIT_DT = sys.argv[4] if len(sys.argv) > 4 else "fp32"          # dtype of the frozen trunk inside the step: fp32 | fp16 (the reference's autocast)
PROD = sys.argv[5] if len(sys.argv) > 5 else "fp32"            # products of the trainable part: fp32 (parity mode) | fp16 (operand copies: autocast arithmetic)


class _Tok:
    def __init__(self, ids, mask):
        self.ids, self.mask = ids, mask

    def __call__(self, text, **kw):
        return TokenBatch(self.ids[:len(text)], self.mask[:len(text)])


cfg = get_config("pretrain")
sd = synth.make_state_dict(cfg, seed=0)
ids, mask, _ = synth.make_queries(B, B, seed=1)
images = synth.make_images(2 * B, seed=0)
model = Blip2QformerCirAlignPrompt(cfg=cfg, compute_dtype="fp32", max_batch=B, train_vit_dtype=VIT_DT, train_products=PROD)
model.load_state_dict(sd, strict=False)
model = model.to(DEV)
model.tokenizer = _Tok(ids, mask)
batch = {"image": images[:B].to(DEV), "target": images[B:].to(DEV), "text_input": ["caption"] * B}
opt = torch.optim.AdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 2e-5, "betas": (0.9, 0.98), "eps": 1e-7,
                          "weight_decay": 0.05}])
scaler = torch.cuda.amp.GradScaler()


def step():
    opt.zero_grad()
    with torch.cuda.amp.autocast():
        d = model(batch)
        loss = d["loss_itc"] + 0.4 * d["loss_rtc"] + 0.4 * d["loss_align"]
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    return float(loss)


hist = [step() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
hist += [step() for _ in range(STEPS)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
# 2 B images through the frozen ViT (forward only) + four Q-Former passes forward and backward
print(f"[train step, HIP path: {VIT_DT} frozen trunk, {PROD} products in the Q-Former's forward + backward] batch {B} (2 x {B} images through ViT-g): {dt * 1e3:.0f} ms per step = {B / dt:.1f} triplets/s; "
      f"loss {hist[0]:.4f} -> {hist[-1]:.4f} over {len(hist)} AdamW steps")

if CPU_B > 0:
    from oracle import sprc_oracle as O  # noqa: E402  (the CPU baseline leg: the oracle as the thing timed against, not shipped)
    torch.set_num_threads(torch.get_num_threads())
    sdc = {k: v.detach().clone().float().requires_grad_(not k.startswith("visual_encoder.") and v.dtype.is_floating_point) for k, v in sd.items()}
    t0 = time.perf_counter()
    losses = O.training_losses(sdc, cfg, images[:CPU_B], images[B:B + CPU_B], ids[:CPU_B], mask[:CPU_B])
    total = losses["loss_itc"] + 0.4 * losses["loss_rtc"] + 0.4 * losses["loss_align"]
    total.backward()
    dc = time.perf_counter() - t0
    print(f"[train step, CPU oracle (torch autograd, fp32, {torch.get_num_threads()} threads)] batch {CPU_B}: {dc:.1f} s = {CPU_B / dc:.2f} triplets/s "
          f"(forward + backward, no optimiser step)")
