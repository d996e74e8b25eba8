#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr log) per kernel: registers, spills, scratch, occupancy.
    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> log; python tools/kres.py log"""
import re
import sys

txt = open(sys.argv[1]).read()
K = {"v": "VGPRs", "a": "AGPRs", "sp": "VGPR Spill", "ss": "SGPRs Spill", "sc": r"ScratchSize \[bytes/lane\]", "oc": r"Occupancy \[waves/SIMD\]"}
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]
    r = {k: (re.search(v + r": (\d+)", b) or [None, "?"])[1] for k, v in K.items()}
    print("VGPR %4s AGPR %4s vspill %4s sspill %4s scratch %5s occ %2s  %s" % (r["v"], r["a"], r["sp"], r["ss"], r["sc"], r["oc"], name[:170]))
