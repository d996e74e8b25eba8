#!/usr/bin/env python3
"""Steady-state K loop of the 256 x 256 GEMM kernels in a `hipcc -S --cuda-device-only` listing: length, MFMAs, v_readfirstlane / waterfall loops
(a buffer descriptor that ended up in VGPRs: guide T20), v_readlane (spilled SGPRs), SALU / VALU counts.  A healthy loop: 207 instructions, 64 MFMAs,
0 waterfalls, 0 readlanes, 0 VALU.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o x.s tools/kloop_probe.hip; python tools/kloop_stat.py x.s
(tests/test_kloop_listing.py runs exactly this on every CPU test pass: the regression it guards against is silent -- all parity tests stay green.)"""
import re
import sys

KERNELS = {"qkv-like (fp16 -> fp16)": "_ZN4sprc16gemm_anti_kernelIDF16_DF16_Li0ELb0ELb0ELb0EEEvNS_10GemmParamsE",
           "fc1-like (fp16 -> fp16, GELU)": "_ZN4sprc16gemm_anti_kernelIDF16_DF16_Li1ELb0ELb0ELb0EEEvNS_10GemmParamsE",
           "proj / fc2-like (fp16 -> fp32 + residual)": "_ZN4sprc16gemm_anti_kernelIDF16_fLi0ELb0ELb0ELb0EEEvNS_10GemmParamsE"}


def steady_loop(lines, name):
    """-> dict of counts over the innermost back-edge loop that holds >= 32 MFMAs in the kernel `name` (None if the kernel is not in the listing)"""
    idx = [k for k, l in enumerate(lines) if l.startswith(name + ":")]
    if not idx:
        return None
    i = j = idx[0]
    while "s_endpgm" not in lines[j]:
        j += 1
    body = [l for l in lines[i:j] if l.strip() and not l.strip().startswith(";")]
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    best = None
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), k) < k:
            a = labels[m.group(1)]
            nm = sum("v_mfma" in x for x in body[a:k])
            if nm >= 32 and (best is None or (nm, k - a) < (best[0], best[2] - best[1])):      # fewest MFMAs, then shortest: the innermost loop
                best = (nm, a, k)
    if best is None:
        return {"kernel_lines": len(body), "len": 0, "mfma": 0, "waterfall": -1, "readfirstlane": -1, "readlane": -1, "salu": -1, "valu": -1}
    nm, a, k = best
    seg = [x.strip() for x in body[a:k]]
    return {"kernel_lines": len(body), "len": k - a, "mfma": nm, "readfirstlane": sum(x.startswith("v_readfirstlane") for x in seg),
            "waterfall": sum(x.startswith("s_cbranch_execnz") for x in seg), "readlane": sum(x.startswith("v_readlane") for x in seg),
            "salu": sum(x.startswith("s_") for x in seg), "valu": sum(x.startswith("v_") and "mfma" not in x for x in seg)}


if __name__ == "__main__":
    for path in sys.argv[1:]:
        lines = open(path).read().split("\n")
        for what, name in KERNELS.items():
            print(path.split("/")[-1], what, steady_loop(lines, name))
