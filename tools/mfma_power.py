#!/usr/bin/env python3
"""What a flop costs per MFMA kind when NOTHING else runs: tools/bin/mfma_power (register-resident MFMA loops on every SIMD, tools/mfma_power.hip) under
rocm-smi.  Prints TFLOP/s, socket power, shader clock and the energy per flop above the idle socket.
    hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_power tools/mfma_power.hip ; python tools/mfma_power.py      (gpurun)"""
import json
import os
import subprocess
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ["v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_f16", "v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3)",
         "v_mfma_scale_f32_32x32x64_f8f6f4 (e2m3 x e2m3: fp6)", "v_mfma_f32_32x32x2_f32"]


def smi():
    try:
        d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout)
        c = d[sorted(k for k in d if k.startswith("card"))[0]]
        pw = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k and v not in ("N/A", "")), None)
        ck = next((v for k, v in c.items() if k.startswith("sclk clock speed")), None)
        return pw, (float("".join(ch for ch in ck if ch.isdigit() or ch == ".")) if ck else None)
    except Exception:
        return None, None


idle = smi()
print(f"idle socket: {idle[0]} W")
for kind, name in enumerate(KINDS):
    stop, acc = threading.Event(), []

    def sample():
        time.sleep(2.0)                                     # (the clock settles)
        while not stop.is_set():
            acc.append(smi())
            stop.wait(0.2)
    th = threading.Thread(target=sample)
    th.start()
    p = subprocess.run([os.path.join(ROOT, "tools", "bin", "mfma_power"), str(kind), "9"], capture_output=True, text=True, timeout=120)
    stop.set()
    th.join()
    acc = [(a, b) for a, b in acc if a is not None and b is not None]
    tf = float(p.stdout.split(":")[1].split("TFLOP")[0]) if "TFLOP" in p.stdout else float("nan")
    pw, ck = (sum(a for a, _ in acc) / len(acc), sum(b for _, b in acc) / len(acc)) if acc else (float("nan"), float("nan"))
    print(f"{name:52s} {tf:8.1f} TFLOP/s | {pw:7.1f} W | {ck:6.0f} MHz | {(pw - 310.0) / tf:6.3f} pJ per flop above the idle socket ({pw / tf:6.3f} all in) | "
          f"{tf / (ck / 2400.0):8.1f} TFLOP/s scaled to 2.4 GHz", flush=True)
