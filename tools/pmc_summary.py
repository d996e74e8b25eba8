#!/usr/bin/env python3
"""Summarise the rocprofv3 counter passes of tools/pmc_kernel.sh: pmc_summary.py OUTDIR MATCH [command text].
MATCH is one kernel-name substring, or several classes "name=substr,name=substr" (each summarised separately from the same
passes).  Writes OUTDIR/summary.json = {class: {match, counters: {name: {n, mean, sum}}, derived: {...}}}."""
import collections
import csv
import glob
import json
import sys

O, match_arg = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
classes = dict(c.split("=", 1) for c in match_arg.split(",")) if "=" in match_arg else {match_arg: match_arg}
rows_c = [r for f in glob.glob(f"{O}/p*/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
rows_t = [r for f in glob.glob(f"{O}/kt/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]


def summarise(match):
    acc = collections.defaultdict(list)
    for r in rows_c:
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows_t if match in r["Kernel_Name"]]
    s = {k: {"n": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for k, v in sorted(acc.items())}
    d = {}

    def m(k):
        return s[k]["mean"] if k in s else None

    if dur:
        d["mean_duration_us"] = sum(dur) / len(dur) / 1e3
        d["total_duration_ms"] = sum(dur) / 1e6
        d["dispatches_timed"] = len(dur)
    # GRBM_GUI_ACTIVE (summed over the 8 XCDs) spans the counter-collection window of a dispatch, which is longer than the kernel:
    # "GRBM_GUI_ACTIVE / 8 / kernel time" came out at 3.3 GHz for 12-us launches on a 2.4-GHz part.  The ratio
    # is only reported for dispatches long enough for the window's edges not to matter (>= 150 us), and flagged as an upper bound.
    mean_us = sum(dur) / len(dur) / 1e3 if dur else 0.0
    if m("GRBM_GUI_ACTIVE") and dur and mean_us >= 150.0:
        d["effective_clock_GHz_upper_bound"] = m("GRBM_GUI_ACTIVE") / 8 / (sum(dur) / len(dur))
    if m("SQ_VALU_MFMA_BUSY_CYCLES") and dur:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 MFMA), summed over the chip's 1024 SIMDs.  Against the kernel's
        # WALL time at the part's maximum clock: the matrix pipe's share of the time the launch took, whatever clock it ran at
        d["mfma_busy_frac_of_wall_at_2p4GHz"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (sum(dur) / len(dur) * 2.4)
    if m("SQ_VALU_MFMA_BUSY_CYCLES") and m("GRBM_GUI_ACTIVE"):
        d["mfma_busy_frac"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (m("GRBM_GUI_ACTIVE") / 8)       # of the counter window's cycles
    if m("SQ_LDS_BANK_CONFLICT") is not None and m("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = m("SQ_LDS_BANK_CONFLICT") / m("SQ_LDS_IDX_ACTIVE")
    if m("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if m(k) is not None:
                d[k.lower() + "_frac_of_wave_cycles"] = m(k) / m("SQ_WAVE_CYCLES")
    if m("FETCH_SIZE") is not None:
        d["fetch_bytes_per_dispatch_x2"] = 2.0 * 1024.0 * m("FETCH_SIZE")                # KiB; doubled: MI355X_MICROARCH.md HBM section
    if m("WRITE_SIZE") is not None:
        d["write_bytes_per_dispatch"] = 1024.0 * m("WRITE_SIZE")
    if dur and "fetch_bytes_per_dispatch_x2" in d:
        d["hbm_side_GBs"] = (d["fetch_bytes_per_dispatch_x2"] + d.get("write_bytes_per_dispatch", 0.0)) / (sum(dur) / len(dur))
    return {"match": match, "counters": s, "derived": d}


out = {"command": cmd, "classes": {c: summarise(mt) for c, mt in classes.items()}}
json.dump(out, open(f"{O}/summary.json", "w"), indent=1)
for c, v in out["classes"].items():
    print(f"== {c} ({v['match']})")
    for k, x in v["counters"].items():
        print(f"  {k:30s} n={x['n']:5d} mean={x['mean']:16.1f}")
    print(json.dumps(v["derived"], indent=1))
