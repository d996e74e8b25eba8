#!/usr/bin/env python3
"""bench.py -- throughput of the SPRC retrieval hot path on MI355X.

metric  : gallery images encoded+ranked / s   (BASELINE.json)
workload: BASELINE.json configs[1] -- "CIRR-val full gallery (~2k), ViT-g bf16, 1xMI355X, batch 128":
          a 2297-image gallery and 4181 composed queries (CIRR-val sizes).
step    : one pass of the hot path over one batch of synthetic input =
            encode 128 gallery images   (ViT-g/14, 39 blocks -> ln_vision -> Q-Former -> vision_proj -> L2 norm),
            fuse   233 composed queries (ceil(128 * 4181/2297): Q-Former pass 1 + pass 2 -> text_proj -> L2 norm),
            rank   them against the FULL resident 2297-image gallery (max-over-32 cosine similarity, top-51).
          value = images processed by all ranks / wall time; inputs are resident in HBM before the timed region.
N > 1   : weak scaling, one rank per GPU: every rank owns a 2297-image gallery shard and its own queries; the only
          exchanges are the all_gather of fused query vectors and of per-shard top-k (sprc_amd/dist.py).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel class = the 16-bit MFMA GEMM -- fp16 operands by default, the
reference's own GPU precision -- measured live with HIP events on the launch stream inside the timed region; `frac` prices the
class's ALGORITHMIC flops, `executed_tflop_per_step` says what the split-precision products add on top) and `cpu_baseline`
(the fp32 CPU oracle on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP's runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (ROCm default 4; streams beyond that share them, which ones depends on the
# order the runtime saw the streams).  The pipelined step has three streams in flight (the ViT, the gallery-side and the query-side Q-Former
# passes): with TWO hardware queues it is 0.8-1.0 % faster than with 3, 4 or 8 on every box measured, and ONE queue reproduces the slow mode some
# boxes fell into with the default (step / GEMM-class time 1.16 instead of 1.11: profiles/r06_hwq_sweep.txt).  Set before the HIP runtime starts,
# for the single-process run only (ranks of an N > 1 job keep the runtime's default: RCCL's streams have never been measured under it); a value in
# the caller's environment wins.
def _gpus_arg(argv) -> int:
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv) and argv[i + 1].isdigit():
            return int(argv[i + 1])
        if a.startswith("--gpus=") and a[7:].isdigit():
            return int(a[7:])
    return 1


if int(os.environ.get("WORLD_SIZE", "1")) == 1 and _gpus_arg(sys.argv[1:]) == 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GALLERY, QUERIES, BATCH, TOPK = 2297, 4181, 128, 51
Q_PER_STEP = (BATCH * QUERIES + GALLERY - 1) // GALLERY          # 233 -> keep CIRR-val's query:image ratio
MFMA_BF16_PEAK_TFLOPS = 2500.0                                    # dense bf16, MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0                                     # dense fp8 (MX-scaled K = 128 / 64 instructions), same table
HBM_PEAK_GBS = 8000.0
# ALGORITHMIC work per unit (BASELINE.md section 4 / SURVEY.md section 8(d); 1 MAC = 2 FLOP): what `roofline.step_frac` prices a
# whole step with -- extra MFMA work an implementation chooses to do (the split-precision Q-Former's lo terms) does not count
GFLOP_PER_IMAGE = {"pretrain": 533.4, "pretrain_vitL": 166.2}
GFLOP_PER_QUERY = {"pretrain": 29.3, "pretrain_vitL": 27.5}
FLOP_PER_PAIR = 16384.0
# the GEMM class = every launch of these kernels (sprc_amd/csrc/gemm.hip); the 256x256 anti-phase kernel carries > 95 % of
# the class time at the bench shapes, the 128x128 kernel the remainder rows and the small Q-Former products
GEMM_KERNELS = {"fp16": "sprc::gemm_anti_kernel<f16,...> (256x256 anti-phase, v_mfma_f32_32x32x16_f16, dominant) + sprc::gemm_kernel<f16,...> (128x128 / 64x64)",
                "bf16": "sprc::gemm_anti_kernel<...> (256x256 anti-phase, dominant) + sprc::gemm_kernel<bf16,...> (128x128)",
                "fp8": "sprc::gemm_anti_kernel<..., FP8> (256x256 anti-phase, MX-scaled e4m3 MFMA: ViT qkv / fc1 / fc2) + the bf16 kernels (proj, Q-Former)",
                "fp32": "sprc::gemm_kernel<float,...> (exact fp32 MFMA)"}


def kernel_source_sha() -> str:
    """sha256 over the HIP sources: ties a committed counter profile to the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sprc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed steps (the default N = 1 run also spends ~40 s AFTER the timed region on the `recall` object "
                    "and ~15 s on `cpu_baseline`: --no-recall / --no-cpu-baseline skip them)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32", "fp8"], help="fp16: fp16 MFMA operands (the reference's GPU autocast precision; the bf16 rate); fp8: bf16 engine whose ViT qkv / fc1 / fc2 GEMMs run on e4m3fn operands (BASELINE config C5)")
    ap.add_argument("--fp8-base", default="bf16", choices=["bf16", "fp16"], help="--dtype fp8: the 16-bit dtype of everything but the fp8 GEMMs (fp16: with the split-precision Q-Former)")
    ap.add_argument("--fp8-layers", default="all", choices=["all", "mlp"], help="--dtype fp8: which ViT GEMMs take e4m3 operands: qkv + fc1 + fc2, or fc1 + fc2 only")
    ap.add_argument("--backbone", default="pretrain", choices=["pretrain", "pretrain_vitL"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-every", type=int, default=10, help="record per-launch HIP events on every Nth timed step, starting with the first (0 = never)")
    ap.add_argument("--vit-streams", type=int, default=1, help="2 = pipeline the two halves of a batch on two streams inside sprc_vit_forward (+4 %% images/s; per-kernel timings then overlap)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c5-slice"], help="c5-slice: ONE GPU's share of BASELINE config C5 (1 M-image gallery / 8 = "
                    "a 125 000-image shard, 10 000 queries of which 1 250 are fused here): ViT-L encode steps of 128 images are timed, then the shard's fusion and "
                    "ranking passes once; value = shard images / (encode time extrapolated over the shard + fusion + ranking).  Use with --backbone pretrain_vitL --dtype fp8")
    ap.add_argument("--qf-streams", type=int, default=2, help="2 = the gallery-side and the query-side Q-Former passes of a step run on two streams")
    ap.add_argument("--pipeline", type=int, default=1, help="1 = batch i's Q-Former passes and ranking overlap batch i+1's ViT (side streams, raw embeddings "
                    "double-buffered); instrumented steps (--prof-every) stay serialised")
    ap.add_argument("--qf-group", type=int, default=1, help="G: the Q-Former stage (gallery-side pass, query fusion, ranking) of G consecutive steps runs once, "
                    "on their G x 128 images and G x 233 queries; the ViT still runs per step on batches of 128 (1 = every step on its own)")
    ap.add_argument("--no-recall", action="store_true", help="skip the `recall` object (default ON at N = 1: after the timed region, Recall@1/5/10/50 + subset "
                    "recalls of the benchmarked engine on the planted-structure CIRR-val-sized case next to the UNMODIFIED REFERENCE's own scores for "
                    "every 22nd query of that case -- 191 queries x 2297 images, tests/golden/planted_c2_subset_eva*.npz; ~40 s; synthetic weights: "
                    "the real checkpoint is a network fetch)")
    ap.add_argument("--no-power", action="store_true", help="skip roofline.power (socket power + shader clock from rocm-smi over ~6 s of extra un-timed steps "
                    "after the timed region, N = 1)")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` object (default ON for the default N = 1 fp16 ViT-g line: config C5's per-GPU encode step -- "
                    "ViT-L backbone, fp8 MFMA, 20 timed steps -- measured by a second invocation of this script after the timed region, ~15 s)")
    ap.add_argument("--recall", action="store_true", help="(accepted for compatibility: the recall object is on by default)")
    ap.add_argument("--cpu-images", type=int, default=32, help="size of the bounded CPU-baseline sample (~15 s of CPU work on 16 cores)")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full"], help="full: ONLY run SURVEY.md section 8(d)'s CPU-baseline procedure -- config C1's "
                    "sizes (256 gallery images, 64 queries), one warm-up + best of 3, all host cores (~10 min) -- print it as one JSON line and write "
                    "gpurun_out/cpu_baseline_c1.json; no GPU work, no bench line (once per round -> profiles/rNN_cpu_baseline_c1.json)")
    return ap.parse_args()


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota and at 64 threads
    (torch's intra-op scaling on 257x1408 GEMMs is flat or negative beyond that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, n_img: int):
    """The fp32 CPU oracle (a port of the reference's CPU path; oracle/sprc_oracle.py) on a bounded sample of the
    same workload: encode n_img images, fuse 2*n_img queries, rank them; all host cores."""
    from oracle import sprc_oracle as O
    from sprc_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(cfg, seed=0)
    images = synth.make_images(n_img, seed=0)
    nq = 2 * n_img
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    ids, mask, ref = synth.make_queries(nq, n_img, seed=1)
    with torch.no_grad():
        O.extract_target_features(sd, cfg, images[:1])             # warm-up (thread pools, page-in)
        t0 = time.perf_counter()
        feats, raw = O.extract_target_features(sd, cfg, images)
        t1 = time.perf_counter()
        sim = O.inference(sd, cfg, raw[ref], feats, ids, mask)
        O.topk_stable(sim.numpy(), min(TOPK, n_img))
        t2 = time.perf_counter()
    # scale the query side to the workload's ratio (QUERIES/GALLERY queries per image)
    per_img = (t1 - t0) / n_img + (t2 - t1) / nq * (QUERIES / GALLERY)
    return {"value": round(1.0 / per_img, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "encode_images_per_s": round(n_img / (t1 - t0), 4), "fuse_rank_queries_per_s": round(nq / (t2 - t1), 3),
            "sample": f"BOUNDED sample, one pass after a one-image warm-up (SURVEY 8(d) plans 256 images / 64 queries, best of 3: ~11 min "
                      f"of CPU time, outside the bench's budget): oracle fp32 (torch CPU, {cores} threads on {cpu_model}): encode {n_img} "
                      f"images {t1 - t0:.1f}s + fuse/rank {nq} queries vs {n_img} images {t2 - t1:.1f}s; value = 1 / (s per image + "
                      f"{QUERIES}/{GALLERY} x s per query)"}


def cpu_baseline_full(cfg, n_img: int = 256, nq: int = 64, reps: int = 3):
    """SURVEY.md section 8(d)'s procedure as written: config C1's sizes, `torch.set_num_threads(usable cores)`, one warm-up pass over a
    16-image batch, then `reps` timed passes; best-of-`reps` for the encode and the fuse + rank halves separately and combined."""
    from oracle import sprc_oracle as O
    from sprc_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    cpu_model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.lower().startswith("model name")), "unknown")
    sd = synth.make_state_dict(cfg, seed=0)
    images = synth.make_images(n_img, seed=0)
    ids, mask, ref = synth.make_queries(nq, n_img, seed=1)
    enc, fr = [], []
    with torch.no_grad():
        O.extract_target_features(sd, cfg, images[:16])
        for _ in range(reps):
            t0 = time.perf_counter()
            parts = [O.extract_target_features(sd, cfg, images[s:s + 32]) for s in range(0, n_img, 32)]
            feats, raw = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
            t1 = time.perf_counter()
            sim = O.inference(sd, cfg, raw[ref], feats, ids, mask)
            O.topk_stable(sim.numpy(), min(TOPK, n_img))
            t2 = time.perf_counter()
            enc.append(t1 - t0)
            fr.append(t2 - t1)
            print(f"  pass: encode {n_img} images {t1 - t0:.1f}s, fuse + rank {nq} queries {t2 - t1:.1f}s", file=sys.stderr, flush=True)
    e, f = min(enc), min(fr)
    per_img = e / n_img + f / nq * (QUERIES / GALLERY)
    return {"procedure": "SURVEY.md section 8(d): config C1 sizes, one warm-up, best of %d" % reps, "kind": "port", "cores": cores, "cpu_model": cpu_model,
            "gallery_images": n_img, "queries": nq, "encode_s_per_pass": [round(x, 2) for x in enc], "fuse_rank_s_per_pass": [round(x, 2) for x in fr],
            "encode_images_per_s": round(n_img / e, 4), "fuse_rank_queries_per_s": round(nq / f, 3),
            "value": round(1.0 / per_img, 4), "unit": "images/s",
            "combined": f"1 / (s per image + {QUERIES}/{GALLERY} x s per query): the workload's query : image ratio"}


def respawn(n: int) -> int:
    """`bench.py --gpus N` outside a launcher: start N ranks of this script under torch.distributed.run (one per GPU;
    rank r on GPU r, or r % visible GPUs when fewer are visible -- see `backend` in main)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd)


def c5_slice(a, dev, rank, world):
    """One GPU's share of config C5 (see --workload): K timed encode steps, then the shard's 1250 fusions and the 10k x 125k ranking."""
    from sprc_amd import engine as E
    from sprc_amd import synth
    from sprc_amd.config import get_config
    if world != 1:
        raise SystemExit("--workload c5-slice is a single-GPU line (the per-GPU share of the 8-GPU configuration)")
    N_SHARD, NQ, NQ_LOCAL, QB = 125000, 10000, 1250, 2048
    cfg = get_config(a.backbone)
    sd = synth.make_state_dict(cfg, seed=0, device=str(dev))
    g = torch.Generator(device=dev).manual_seed(99)
    images = torch.randn((BATCH, 3, 224, 224), generator=g, device=dev)
    if a.dtype == "fp8":
        cal = E.Engine(cfg, sd, dev, dtype=a.fp8_base, max_batch=BATCH, qformer_x3=0)
        amax = cal.calibrate_fp8(images)
        del cal
        torch.cuda.empty_cache()
        eng = E.Engine(cfg, sd, dev, dtype="fp8", max_batch=250, fp8_amax=amax, fp8_margin=1.1, fp8_base=a.fp8_base, fp8_layers=a.fp8_layers)
    else:
        eng = E.Engine(cfg, sd, dev, dtype=a.dtype, max_batch=250)
    del sd
    raw = torch.empty((BATCH, cfg.vit.tokens, cfg.vit.width), dtype=torch.float32, device=dev)
    shard = torch.nn.functional.normalize(torch.randn((N_SHARD, 32, cfg.embed_dim), generator=g, device=dev), dim=-1).to(torch.bfloat16)
    fusion_all = torch.nn.functional.normalize(torch.randn((NQ, cfg.embed_dim), generator=g, device=dev), dim=-1)
    ids, mask, _ = synth.make_queries(NQ_LOCAL, BATCH, seed=5)
    ids, mask = ids.to(dev), mask.to(dev)
    ref_slot = (7919 * torch.arange(NQ_LOCAL, device=dev)) % BATCH
    from sprc_amd.dist import ShardedRanker
    # the library's ranking path: local scores in blocks of QB query rows (a 1-GB budget), never the 5-GB [10 000, 125 000] matrix
    ranker = ShardedRanker(shard, index_base=0, sim_budget_bytes=QB * N_SHARD * 4)

    def step(i):
        eng.vit_forward(images, out=raw)
        feats, f16 = eng.qformer_image(raw)
        lo = (i * BATCH) % (N_SHARD - BATCH)
        shard[lo:lo + BATCH].copy_(f16 if f16 is not None and f16.dtype == torch.bfloat16 else feats)

    def fuse_and_rank():
        for s in range(0, NQ_LOCAL, 250):
            f, _ = eng.qformer_fuse(raw.index_select(0, ref_slot[s:s + 250]), ids[s:s + 250], mask[s:s + 250])
            fusion_all[s:s + 250].copy_(f)
        return ranker.rank(fusion_all.to(torch.bfloat16), TOPK)

    for i in range(a.warmup):
        step(i)
    fuse_and_rank()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / a.steps
    t0 = time.perf_counter()
    fuse_and_rank()
    torch.cuda.synchronize()
    t_fr = time.perf_counter() - t0
    total = N_SHARD / BATCH * t_step + t_fr
    peak = MFMA_FP8_PEAK_TFLOPS if a.dtype == "fp8" else MFMA_BF16_PEAK_TFLOPS
    enc_tflop = BATCH * GFLOP_PER_IMAGE[a.backbone] * 1e-3
    out = {"metric": "gallery images encoded+ranked/sec", "value": round(N_SHARD / total, 2), "unit": "images/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(t_step * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": f"C5 slice (one GPU of eight): {N_SHARD}-image shard of a 1 M gallery, {NQ} queries ranked, {NQ_LOCAL} fused here; "
                                  f"{'ViT-L' if a.backbone == 'pretrain_vitL' else 'ViT-g'} {a.dtype} encode timed over {a.steps} steps of {BATCH} images and "
                                  f"EXTRAPOLATED over the shard ({N_SHARD / BATCH * t_step:.1f} s), + the shard's fusion and bf16 ranking passes measured once "
                                  f"({t_fr * 1e3:.0f} ms)", "backbone": a.backbone, "batch": BATCH, "shard": N_SHARD, "queries": NQ,
                      "fuse_rank_ms": round(t_fr * 1e3, 1), "rank_dtype": "bf16", "topk": TOPK},
           "roofline": {"bound": "mfma", "kernel": GEMM_KERNELS.get(a.dtype, a.dtype), "achieved": round(enc_tflop / t_step, 1), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(enc_tflop / t_step / peak, 4), "traffic": None,
                        "note": "whole encode step (algorithmic flops of 128 images / step time), not a per-kernel figure"}}
    print(json.dumps(out), flush=True)


def power_sample(run_steps, seconds: float = 6.0):
    """Socket power and shader clock (rocm-smi, ~4 Hz) while `run_steps(n)` repeats the benchmarked step AFTER the timed region: the
    2.5 PFLOP/s the roofline prices against is the 2.4-GHz figure, and under this step the part sits at its power cap well below that clock
    (1400 W, 1.55 GHz under the GEMM alone).  None when rocm-smi is missing or prints something else."""
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if exe is None:
        return None
    stop, acc = threading.Event(), []

    def sampler():
        while not stop.is_set():
            try:
                d = json.loads(subprocess.run([exe, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout)
                c = d[sorted(k for k in d if k.startswith("card"))[0]]
                pw = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k and v not in ("N/A", "")), None)
                ck = next((v for k, v in c.items() if k.startswith("sclk clock speed")), None)
                mhz = float("".join(ch for ch in ck if ch.isdigit() or ch == ".")) if ck else None
                if pw is not None and mhz is not None:
                    acc.append((pw, mhz))
            except Exception:
                pass
            stop.wait(0.2)

    th = threading.Thread(target=sampler, daemon=True)
    t0, n = time.perf_counter(), 0
    run_steps(3)                                             # the clock settles within a few steps
    th.start()
    while time.perf_counter() - t0 < seconds:
        run_steps(4)
        n += 4
    stop.set()
    th.join(timeout=30)
    acc = acc[1:]                                            # (the first sample straddles the start)
    if len(acc) < 3:
        return None
    pw, ck = [x[0] for x in acc], [x[1] for x in acc]
    return {"socket_w_mean": round(sum(pw) / len(pw), 1), "socket_w_max": round(max(pw), 1), "sclk_mhz_mean": round(sum(ck) / len(ck), 1),
            "samples": len(acc), "steps_run": n,
            "source": "rocm-smi --showpower --showclocks, ~4 Hz, over extra un-timed steps of the same pipelined schedule after the timed region"}


def c5_extra(dev) -> dict:
    """BASELINE config C5's per-GPU number on the box that measures the headline: the C2-shaped step on the ViT-L backbone with
    the ViT's qkv / fc1 / fc2 products on e4m3 operands (fp8 MFMA), 20 timed steps, by a second invocation of this script (its own process: its own
    engine, calibration pass and streams; the parent's engine is idle by now).  `value` / `config` of the line stay the headline's."""
    import subprocess
    torch.cuda.synchronize()
    cmd = [sys.executable, os.path.abspath(__file__), "--backbone", "pretrain_vitL", "--dtype", "fp8", "--steps", "20", "--warmup", "3",
           "--no-recall", "--no-cpu-baseline", "--no-extra", "--no-power", "--prof-every", "0"]
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        line = next(ln for ln in p.stdout.splitlines() if ln.startswith("{"))
        d = json.loads(line)
    except Exception as e:                                   # the headline line must not die with its appendix
        return {"c5_per_gpu_step": None, "error": repr(e)[:300]}
    return {"c5_per_gpu_step": {"workload": d["config"]["workload"], "dtype": d["dtype"], "backbone": d["config"]["backbone"],
                                "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                                "step_frac": d["roofline"]["step_frac"], "peak_tflops": d["roofline"]["peak"], "step_alg_tflop": d["roofline"]["step_alg_tflop"],
                                "precision": d["config"]["precision"],
                                "note": "config C5's backbone and dtype on ONE GPU, C2-shaped step (128 images + 233 queries vs 2297); whole step priced "
                                        "against the 5 PF dense fp8 peak although attention, proj and the Q-Former run on 16-bit operands"}}


def main():
    a = parse()
    if a.cpu_baseline == "full":
        from sprc_amd.config import get_config
        out = cpu_baseline_full(get_config(a.backbone))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "cpu_baseline_c1.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out), flush=True)
        return
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(respawn(a.gpus))
    if a.vit_streams != 1:
        os.environ["SPRC_VIT_STREAMS"] = str(a.vit_streams)      # read once by the library at its first ViT forward
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only compute path (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    # one rank per GPU over RCCL ("nccl" IS RCCL on ROCm).  With fewer visible GPUs than ranks (a 1-GPU box running
    # `--gpus 2` as a plumbing check) the ranks share devices and the two tiny all_gathers are staged through gloo:
    # RCCL refuses two ranks on one device.  The line then says so in config.backend -- it is not a scaling number.
    backend = "nccl" if world <= ndev else "gloo"
    # SPRC_BENCH_FORCE_DIST=1: a ONE-rank process group anyway (with SPRC_DIST_ALWAYS_EXCHANGE=1 both all_gathers run): the code an
    # N-GPU launch executes -- RCCL init with device_id, the device check, the exchanges, the per-rank record -- on a 1-GPU box
    use_dist = world > 1 or os.environ.get("SPRC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:                   # forced one-rank group outside a launcher
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT=str(sk.getsockname()[1]))
            sk.close()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus
        if backend == "nccl":
            # one rank per DEVICE over RCCL, or the line is not a scaling number: every rank reports its device index and bus id
            assert dist.get_backend() == "nccl"
            mine = (torch.cuda.current_device(), str(getattr(torch.cuda.get_device_properties(dev), "uuid", local)))
            seen = [None] * world
            dist.all_gather_object(seen, mine)
            assert len(set(seen)) == world, f"bench.py --gpus {a.gpus}: ranks share devices under the nccl backend: {seen}"

    from sprc_amd import _lib as L
    from sprc_amd import engine as E
    from sprc_amd import synth
    from sprc_amd.config import get_config
    from sprc_amd.dist import ShardedRanker, owner_of, shard_bounds

    lib = L.load()
    if a.workload == "c5-slice":
        return c5_slice(a, dev, rank, world)
    cfg = get_config(a.backbone)
    sd = synth.make_state_dict(cfg, seed=0, device=str(dev))      # random-init weights of the named architecture
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((BATCH, 3, 224, 224), generator=g, device=dev)           # synthetic, already "normalised"
    if a.dtype == "fp8":
        # static activation scales: one calibration pass of the 16-bit engine over a batch of the synthetic images
        # (outside the timed region, like packing the weights); 10 % head-room over the observed maxima
        cal = E.Engine(cfg, sd, dev, dtype=a.fp8_base, max_batch=BATCH, qformer_x3=0)
        amax = cal.calibrate_fp8(images)
        del cal
        torch.cuda.empty_cache()
        eng = E.Engine(cfg, sd, dev, dtype="fp8", max_batch=max(1, a.qf_group) * max(BATCH, Q_PER_STEP), fp8_amax=amax, fp8_margin=1.1,
                       fp8_base=a.fp8_base, fp8_layers=a.fp8_layers)
    else:
        eng = E.Engine(cfg, sd, dev, dtype=a.dtype, max_batch=max(1, a.qf_group) * max(BATCH, Q_PER_STEP),
                       qformer_x3=(0 if os.environ.get("SPRC_X3_OFF") else None))       # SPRC_X3_OFF=1: A/B line without the split-precision Q-Former
    del sd
    torch.cuda.empty_cache()
    ids, mask, _ = synth.make_queries(Q_PER_STEP, GALLERY, seed=1 + rank)
    ids, mask = ids.to(dev), mask.to(dev)
    ref_slot = (7919 * torch.arange(Q_PER_STEP, device=dev)) % BATCH             # references come from the batch's raw embeds
    gallery = torch.nn.functional.normalize(torch.randn((GALLERY, 32, cfg.embed_dim), generator=g, device=dev), dim=-1)
    # weak scaling: the global gallery has world x 2297 images in contiguous, balanced shards; a query is fused on the rank
    # that owns its reference image (sprc_amd/dist.py: owner_of) -- here every rank draws references from its own batch
    lo_g, hi_g = shard_bounds(world * GALLERY, world, rank)
    assert (lo_g, hi_g) == (rank * GALLERY, (rank + 1) * GALLERY)
    assert bool((owner_of(lo_g + ref_slot.cpu(), world * GALLERY, world) == rank).all())
    ranker = ShardedRanker(gallery, index_base=lo_g)
    # --qf-group G: the ViT runs on every step's batch of 128 images; the Q-Former stage (gallery-side pass, query fusion, ranking) of G
    # consecutive steps runs ONCE, on the G x 128 images / G x 233 queries of those steps.  The Q-Former's products are small at one
    # step's size (233 queries = 14 912 rows: 177 tiles of 256 x 256 on 256 CUs, 128 images = 4 096 rows: latency-bound launches); at
    # G = 4 the same launches fill 91 % of their rounds (tools/qf_group_probe.py: image pass 4.96 -> 3.61 ms per 128 images, fusion
    # 15.8 -> 14.0 ms per 233 queries).  Every step's work is done inside the timed region (a last partial group is flushed before the
    # closing barrier); G = 1 is the step-by-step schedule of rounds 1-4.
    G = max(1, a.qf_group)
    NQG = G * Q_PER_STEP
    ids_g, mask_g = ids.repeat(G, 1), mask.repeat(G, 1)
    ref_slot_g = torch.cat([j * BATCH + ref_slot for j in range(G)])             # step j of a group draws its references from its own batch

    # The gallery-side Q-Former pass (small, latency-bound launches that leave most CUs idle) and the query-side fusion
    # passes are independent until the ranking: they run on two streams (--qf-streams 1 serialises them for A/B runs).
    side = torch.cuda.Stream(device=dev) if a.qf_streams >= 2 else None
    main = torch.cuda.current_stream(dev)
    # --pipeline 1: the Q-Former passes + ranking of group g run on side streams WHILE the ViT of group g+1 runs on the main stream (raw
    # embeddings double-buffered): the ViT's GEMMs own every CU while they run, but their partial last rounds, the small remainder
    # launches and the bandwidth-bound LayerNorms leave CUs idle that the Q-Former's launches can take.  Groups whose launches
    # are timed with HIP events (--prof-every) run on ONE stream, so per-kernel durations stay those of a kernel that has the chip
    # (and agree with a `rocprofv3 --kernel-trace` of `--pipeline 0 --qf-streams 1`, profiles/).
    pipe = a.pipeline and side is not None
    s_img, s_fuse = (side, torch.cuda.Stream(device=dev)) if pipe else (None, None)
    vit_streams = [main]           # (two ViT batches in flight on two streams, half a pass apart, were measured: 89.7 -> 90.3 ms per step -- DESIGN.md 5.4)
    NBUF = 2 if pipe else 1
    raws = torch.empty((NBUF, G * BATCH, cfg.vit.tokens, cfg.vit.width), dtype=torch.float32, device=dev)
    done = [torch.cuda.Event() for _ in range(NBUF)] if pipe else None
    groups = [0]                                                                  # groups run so far: the parity of the raw buffers

    def qformer_stage(buf, first: int, n: int, serial: bool):
        """gallery-side pass for the n x 128 images in buf, fusion of their n x 233 queries, ranking (on the current stream(s))"""
        nq = n * Q_PER_STEP
        refs = buf.index_select(0, ref_slot_g[:nq])                               # references come from the group's raw embeddings

        def image_side():
            feats, _ = eng.qformer_image(buf)                                     # R5(i) + vision_proj
            for j in range(n):                                                    # resident gallery slices of these batches
                lo = ((first + j) * BATCH) % (GALLERY - BATCH)
                gallery[lo:lo + BATCH].copy_(feats[j * BATCH:(j + 1) * BATCH])
            return feats
        if side is not None and not serial:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                image_side().record_stream(side)
            fusion, _ = eng.qformer_fuse(refs, ids_g[:nq], mask_g[:nq])           # R6 fusion half
            main.wait_stream(side)
        else:
            image_side()
            fusion, _ = eng.qformer_fuse(refs, ids_g[:nq], mask_g[:nq])
        return ranker.rank(fusion, TOPK)                                          # R6 similarity + R7 top-k (+ exchanges)

    def group_pipelined(first: int, n: int):
        g = groups[0]
        groups[0] += 1
        buf = raws[g % NBUF][:n * BATCH]
        vst = vit_streams[g % len(vit_streams)]
        vst.wait_event(done[g % NBUF])                                            # group g-NBUF's Q-Former passes have read this buffer
        with torch.cuda.stream(vst):
            for j in range(n):
                eng.vit_forward(images, out=buf[j * BATCH:(j + 1) * BATCH], slot=g % len(vit_streams))     # R3/R4
        s_img.wait_stream(vst)
        s_fuse.wait_stream(vst)
        s_img.wait_event(done[(g - 1) % NBUF])                                    # group g-1's ranking has read the whole gallery: the slice
        nq = n * Q_PER_STEP                                                       # copies below must not land under it
        with torch.cuda.stream(s_img):
            feats, _ = eng.qformer_image(buf)
            for j in range(n):
                lo = ((first + j) * BATCH) % (GALLERY - BATCH)
                gallery[lo:lo + BATCH].copy_(feats[j * BATCH:(j + 1) * BATCH])
        with torch.cuda.stream(s_fuse):
            fusion, _ = eng.qformer_fuse(buf.index_select(0, ref_slot_g[:nq]), ids_g[:nq], mask_g[:nq])
            s_fuse.wait_stream(s_img)                                             # the ranking reads the gallery slices of this group
            out = ranker.rank(fusion, TOPK)
            done[g % NBUF].record(s_fuse)
        return out

    def drain():
        if pipe:
            for st in vit_streams[1:]:
                main.wait_stream(st)
            main.wait_stream(s_img)
            main.wait_stream(s_fuse)

    def group(first: int, n: int, serial: bool = False):
        buf = raws[0][:n * BATCH]
        for j in range(n):
            eng.vit_forward(images, out=buf[j * BATCH:(j + 1) * BATCH])
        return qformer_stage(buf, first, n, serial)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(0, a.warmup, G):
        (group_pipelined if pipe else group)(i, min(G, a.warmup - i))
    drain()
    barrier()
    # Per-launch HIP events cost two stream markers per launch (~7 us of pipeline bubble: 4.5 ms on a 100-ms step when every
    # launch is recorded), so they are recorded on part of the timed region only -- G = 1: every `prof_every`-th step, G > 1: the
    # first group; `value` is the throughput of the WHOLE region, instrumented steps included.
    n_prof = 0
    t0 = time.perf_counter()
    for i in range(0, a.steps, G):
        n = min(G, a.steps - i)
        rec = a.prof_every > 0 and (i % a.prof_every == 0 if G == 1 else i == 0)
        if rec:
            drain()
            lib.sprc_prof_enable(1 if n_prof == 0 else 2)
            n_prof += n                                                           # instrumented STEPS
            group(a.warmup + i, n, serial=True)                                   # instrumented groups: ONE stream (see `pipe`)
            lib.sprc_prof_enable(0)
        else:
            (group_pipelined if pipe else group)(a.warmup + i, n)
    drain()
    barrier()
    dt = time.perf_counter() - t0
    n_prof = max(n_prof, 1)
    power = None
    if world == 1 and not use_dist and not a.no_power:
        def more(k):
            for i in range(0, k, G):
                (group_pipelined if pipe else group)(a.warmup + a.steps + i, min(G, k - i))
            drain()
            torch.cuda.synchronize()
        try:
            power = power_sample(more)
        except Exception:
            power = None
    prof = (L.ProfEntry * len(L.K_CLASSES))()
    L.check(lib.sprc_prof_collect(prof), "sprc_prof_collect")
    lib.sprc_prof_enable(0)
    per_rank_ms = [round(dt / a.steps * 1e3, 3)]
    if use_dist:
        # value uses the MAX over ranks (the contract); every rank's own step time rides along so that a straggler shows in the record
        times = [None] * world
        torch.distributed.all_gather_object(times, dt)
        per_rank_ms = [round(float(x) / a.steps * 1e3, 3) for x in times]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = world * BATCH * a.steps / dt
        kidx = 1 if a.dtype == "fp32" else 0
        pe = prof[kidx]
        # the library pipelines the two halves of a batch on two streams, so launches of one class can overlap in time: the
        # class time is the UNION of the launches' HIP-event intervals (busy_ms); the plain sum is reported next to it
        ach = pe.flops / (pe.busy_ms * 1e-3) / 1e12 if pe.busy_ms > 0 else 0.0
        # fp8: the ViT's qkv / fc1 / fc2 products (85-92 % of the class flops) issue the MX-scaled e4m3 MFMA
        # (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales): the class is priced against the 5 PF dense fp8 peak,
        # although proj and the Q-Former products in the same class run on bf16 operands
        peak = 157.3 if a.dtype == "fp32" else MFMA_FP8_PEAK_TFLOPS if a.dtype == "fp8" else MFMA_BF16_PEAK_TFLOPS
        kernels = {n: {"ms_per_step": round(prof[j].busy_ms / n_prof, 3), "sum_launch_ms_per_step": round(prof[j].ms / n_prof, 3),
                       "launches_per_step": prof[j].launches // n_prof,
                       "tflops": round(prof[j].flops / max(prof[j].busy_ms, 1e-9) / 1e9, 1),
                       "alg_GBs": round(prof[j].bytes / max(prof[j].busy_ms, 1e-9) / 1e6, 1)}
                   for j, n in enumerate(L.K_CLASSES) if prof[j].launches}
        # HBM-side bytes per GEMM launch: PMC counters cannot be read from inside this process, so the figure comes from the
        # committed rocprofv3 passes over this same command (tools/profile_bench.sh -> profiles/r01_traffic.json)
        # -- and only when that profile was taken on THIS build (it records the hash of the kernel sources): otherwise null
        traffic, traffic_src = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_traffic.json")), reverse=True)       # the newest round's first
        if a.dtype == "fp16" and a.backbone == "pretrain" and a.vit_streams == 1 and a.qf_group == 1 and cands and pe.launches and not os.environ.get("SPRC_X3_OFF"):
            sha = kernel_source_sha()
            for tj in cands:
                with open(tj) as f:
                    tr = json.load(f)
                if tr.get("kernel_source_sha") == sha:
                    traffic = round(tr["gemm_bytes_per_step"]["total"] / (pe.launches / n_prof), 1)
                    traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate counter-only "
                                   "passes over this command on this build: kernel_source_sha %s)" % (os.path.basename(tj), sha[:12]))
                    break
        # whole-step utilisation: ALGORITHMIC flops of one step over the step time, against the dtype's peak
        step_tflop = (BATCH * GFLOP_PER_IMAGE[a.backbone] + Q_PER_STEP * GFLOP_PER_QUERY[a.backbone]) * 1e-3 \
            + Q_PER_STEP * GALLERY * FLOP_PER_PAIR * 1e-12
        precision = {"fp16": "fp16 MFMA operands, fp32 accumulate / residual stream / LayerNorm / softmax; ViT fp16 as under the reference's "
                             "autocast (blip2.py:36-44), Q-Former at split-precision (hi + lo fp16 operands, masks image %d / query %d) as the "
                             "reference keeps it in fp32 (align_prompt.py:366-368)" % (eng.x3_image, eng.x3_fuse),
                     "bf16": "bf16 MFMA operands, fp32 accumulate / residual stream / LayerNorm / softmax",
                     "fp8": f"{a.fp8_base} engine, ViT {'qkv / ' if a.fp8_layers == 'all' else ''}fc1 / fc2 on e4m3 MX MFMA", "fp32": "exact fp32 MFMA"}[a.dtype]
        out = {
            "metric": "gallery images encoded+ranked/sec", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"CIRR-val sizes: {GALLERY}-image gallery x {QUERIES} queries, "
                                   f"{'ViT-g' if a.backbone == 'pretrain' else 'ViT-L'} {a.dtype}, batch {BATCH}; step = encode {BATCH} images + "
                                   f"fuse {Q_PER_STEP} queries + rank vs {GALLERY} (top-{TOPK})",
                       "backbone": a.backbone, "batch": BATCH, "queries_per_step": Q_PER_STEP, "gallery": GALLERY,
                       "topk": TOPK, "rank_dtype": "fp32", "sharding": f"gallery-sharded x{world}", "vit_streams": a.vit_streams, "qformer_group": a.qf_group, "qformer_streams": a.qf_streams, "pipeline": int(bool(a.pipeline)), "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                       "precision": precision, "rccl_ranks": (torch.distributed.get_world_size() if use_dist and backend == "nccl" else None),
                       "per_rank_ms_per_step": per_rank_ms,
                       "backend": ("rccl" if backend == "nccl" else f"gloo ({world} ranks sharing {ndev} GPU: plumbing check, not a scaling number)") if use_dist else None},
            "roofline": {"bound": "mfma", "kernel": GEMM_KERNELS[a.dtype], "achieved": round(ach, 1), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "step_frac": round(step_tflop / (dt / a.steps) / peak, 4), "step_alg_tflop": round(step_tflop, 2),
                         "traffic": traffic,
                         "traffic_unit": "HBM-side bytes per launch", "traffic_source": traffic_src,
                         "alg_bytes_per_launch": round(pe.bytes / max(pe.launches, 1), 1),
                         "launches": int(pe.launches), "avg_launch_ms": round(pe.busy_ms / max(pe.launches, 1), 4),
                         "timing": "HIP events on the launch stream(s), recorded on %d of the %d timed steps (every %dth; recording all "
                                   "of them costs 4.5 %% of the step); class time = union of the launch intervals, sum of launch "
                                   "durations = %.4f ms per launch" % (n_prof, a.steps, a.prof_every, pe.ms / max(pe.launches, 1)),
                         "alg_flops_per_launch": round(pe.flops / max(pe.launches, 1), 1),
                         # launches x alg_flops_per_launch (<= step_alg_tflop: the class is part of the step), and what the launches
                         # executed on top of it (split-precision products reduce over 3 K, the patch embedding over zero padding)
                         "class_alg_tflop_per_step": round(pe.flops / n_prof * 1e-12, 3),
                         "executed_tflop_per_step": round(pe.exec_flops / n_prof * 1e-12, 3),
                         "executed_tflops": round(pe.exec_flops / max(pe.busy_ms, 1e-9) / 1e9, 1),
                         # the peak above is the 2.4-GHz figure; `power` says what clock the part's power cap allows under THIS step, and
                         # what the matrix pipes could deliver at that clock (peak x sclk / 2400 MHz)
                         "power": (dict(power, peak_at_sclk_tflops=round(peak * power["sclk_mhz_mean"] / 2400.0, 1),
                                        step_frac_of_peak_at_sclk=round(step_tflop / (dt / a.steps) / (peak * power["sclk_mhz_mean"] / 2400.0), 4))
                                   if power else None)},
            "kernels": kernels,
        }
        if not a.no_recall and world == 1 and a.backbone == "pretrain" and a.dtype in ("fp16", "bf16", "fp32"):
            # BASELINE.json's metric names Recall@K next to the throughput ("Recall@1/5/10 equal to reference on CIRR-val").  No checkpoint and no
            # dataset offline: planted-structure weights / images at CIRR-val's sizes; the yardstick is the scores the unmodified reference
            # produced itself (CPU fp32) for every 22nd query of that case, on fp32-valued weights and on fp16-valued trunk weights (what a
            # GPU-trained reference checkpoint holds) -- fixtures under tests/golden/, generated by oracle/gen_c2_subset.py
            from sprc_amd import planted as P
            del eng
            torch.cuda.empty_cache()
            imgs = list(P.planted_images())
            out["recall"] = {"case": f"planted-structure synthetic weights and images, {P.N_GALLERY} gallery images (CIRR-val's size), every 22nd of "
                                     f"{P.N_QUERIES} composed queries, full depth, engine dtype {a.dtype}; targets at planned ranks of the REFERENCE's ordering; "
                                     "reference = the unmodified reference's CPU fp32 scores (tests/golden/planted_c2_subset_eva*.npz)"}
            for key, fn in (("fp32_weights", "planted_c2_subset_eva.npz"), ("fp16_valued_trunk", "planted_c2_subset_eva_h16.npz")):
                gp = os.path.join(ROOT, "tests", "golden", fn)
                if os.path.exists(gp):
                    out["recall"][key] = P.reference_subset_report(cfg, dev, a.dtype, gp, images=imgs)
            subs = {k: v for k, v in out["recall"].items() if isinstance(v, dict)}
            out["recall"]["fixtures_evaluated"] = sorted(subs)
            # the aggregate is a statement about BOTH checkpoint kinds: null when a fixture is missing (a line from a tree without
            # tests/golden/ must not read "equal" off one file)
            out["recall"]["equal_recall_at_1_5_10"] = all(v["equal_recall_at_1_5_10"] for v in subs.values()) if len(subs) == 2 else None
            del imgs
        if (world == 1 and not use_dist and not a.no_extra and a.backbone == "pretrain" and a.dtype == "fp16" and a.workload == "c2"
                and a.qf_group == 1 and a.vit_streams == 1):
            out["extra"] = c5_extra(dev)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_images)
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
