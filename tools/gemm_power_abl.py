#!/usr/bin/env python3
"""Where does the 256 x 256 GEMM's power go?  The product library next to ABLATION builds of its fp16 translation unit (tools/gemm_power_abl.patch applied
to gemm_impl.hpp, -DSPRC_ANTI_ABL=n: 1 no fragment reads after K-tile 1, 2 no MFMAs, 4 no staging loads after K-tile 1; results are WRONG in those
builds): time per launch, socket power and shader clock (rocm-smi) for the 32768 x 4096 x 1408 product.  Energy per launch = power x time; the idle
socket draws ~310 W.
    python tools/gemm_power_abl.py            (gpurun; needs sprc_amd/libsprc_hip_abl{1,2,3,4}.so)
    python tools/gemm_power_abl.py --worker   (one library: SPRC_LIB_PATH)"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    try:
        d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout)
        c = d[sorted(k for k in d if k.startswith("card"))[0]]
        pw = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k and v not in ("N/A", "")), None)
        ck = next((v for k, v in c.items() if k.startswith("sclk clock speed")), None)
        return pw, (float("".join(ch for ch in ck if ch.isdigit() or ch == ".")) if ck else None)
    except Exception:
        return None, None


def worker():
    import torch
    from sprc_amd import engine as E
    dev = torch.device("cuda", 0)
    A = torch.randn((32768, 1408), device=dev).to(torch.float16)
    W = torch.randn((4096, 1408), device=dev).to(torch.float16)
    Cc = torch.empty((32768, 4096), device=dev, dtype=torch.float16)
    for _ in range(5):
        E.gemm(A, W, out=Cc)
    torch.cuda.synchronize()
    stop, acc = threading.Event(), []

    def sample():
        while not stop.is_set():
            acc.append(smi())
            stop.wait(0.2)
    th = threading.Thread(target=sample)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 10.0:
        for _ in range(50):
            E.gemm(A, W, out=Cc)
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    acc = [(p, c) for p, c in acc[2:] if p is not None and c is not None]
    pw, ck = sum(p for p, _ in acc) / len(acc), sum(c for _, c in acc) / len(acc)
    ms = dt / n * 1e3
    print(f"{os.environ.get('ABL_NAME', 'product library'):46s} {ms:7.4f} ms per launch | {pw:7.1f} W | {ck:6.0f} MHz | {pw * ms:7.1f} mJ per launch "
          f"({(pw - 310.0) * ms:6.1f} mJ above the idle socket) | {len(acc)} samples", flush=True)


if "--worker" in sys.argv:
    worker()
else:
    variants = [("product library", None), ("no fragment reads (ds_read) after K-tile 1", 1), ("no MFMAs", 2), ("no staging loads (LDS DMA) after K-tile 1", 4),
                ("no reads and no MFMAs: staging + barriers + epilogue", 3), ("product library (again)", None)]
    for name, n in variants:
        env = dict(os.environ, ABL_NAME=name)
        if n is not None:
            env["SPRC_LIB_PATH"] = os.path.join(ROOT, "sprc_amd", f"libsprc_hip_abl{n}.so")
        else:
            env.pop("SPRC_LIB_PATH", None)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, timeout=300)
        print((p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1], flush=True)
