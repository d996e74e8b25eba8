#!/usr/bin/env python3
"""Is the step power-bound?  Socket power and shader clock (rocm-smi, sampled while the work runs) for: the 256 x 256 GEMM on the whole chip, on
ONE CU partition (half the chip; the other half idle), on BOTH partitions side by side, the LayerNorm-shaped memory-bound kernel, and the bench step.
    python tools/power_probe.py            (gpurun; writes gpurun_out/r06/power_probe.txt)"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sprc_amd import _lib as L
from sprc_amd import engine as E

dev = torch.device("cuda", 0)
lib = L.load()
main = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
parts = []
for i in range(2):
    h = C.c_void_p()
    L.check(lib.sprc_stream_create_partition(i, 2, C.byref(h)), "partition")
    parts.append(torch.cuda.ExternalStream(h.value, device=dev))


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d[sorted(k for k in d if k.startswith("card"))[0]]
        pw = next((float(v) for k, v in c.items() if "Power" in k and "W" in k and v not in ("N/A", "")), None)
        sclk = next((v for k, v in c.items() if k.startswith("sclk")), None)
        return pw, sclk
    except Exception as e:
        return None, repr(e)[:80]


def sample(stop, acc):
    while not stop.is_set():
        acc.append(smi())
        time.sleep(0.25)


def run(name, fn, secs=12.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc))
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    pw = [p for p, _ in acc[2:] if p is not None]
    clk = [c for _, c in acc[2:] if c]
    print(f"{name:62s} {dt / n * 1e3:8.3f} ms per call | power W: mean {sum(pw) / max(len(pw), 1):7.1f} max {max(pw, default=0):7.1f} ({len(pw)} samples) | sclk samples {clk[:3]} ... {clk[-2:]}", flush=True)


A = torch.randn((32768, 1408), device=dev).to(torch.float16)
W = torch.randn((4096, 1408), device=dev).to(torch.float16)
Cs = [torch.empty((32768, 4096), device=dev, dtype=torch.float16) for _ in range(2)]
x = torch.randn((32896, 1408), device=dev)
g = torch.ones(1408, device=dev)
b = torch.zeros(1408, device=dev)
print("idle:", smi())


def full():
    E.gemm(A, W, out=Cs[0])


def one():
    parts[0].wait_stream(main)
    with torch.cuda.stream(parts[0]):
        E.gemm(A, W, out=Cs[0])
    main.wait_stream(parts[0])


def both():
    for i, st in enumerate(parts):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            E.gemm(A, W, out=Cs[i])
    for st in parts:
        main.wait_stream(st)


def ln():
    E.layernorm(x, g, b, 1e-6, L.SPRC_F16, want32=False, want16=True)


run("256x256 GEMM 32768 x 4096 x 1408, whole chip", full)
run("the same product on ONE partition (128 CUs), the other half idle", one)
run("the same product on BOTH partitions at once (2 products per call)", both)
try:
    run("LayerNorm 32896 x 1408 fp32 -> fp16 (memory-bound)", ln)
except Exception as e:
    print("layernorm probe skipped:", repr(e)[:200])
print("idle:", smi())
