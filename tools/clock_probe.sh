#!/bin/bash
# effective shader clock of the GEMM kernel = GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel duration.
# Usage: tools/clock_probe.sh M N K   (honours SPRC_GEMM_* env)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/clock_$$
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O -o out --output-format csv -- python $R/tools/gemm_one.py $1 $2 $3 6 > /dev/null 2>&1
python - <<PY
import csv, glob
cyc = [float(r["Counter_Value"]) for f in glob.glob("$O/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
dur = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
if cyc and dur:
    c, d = sum(cyc[2:]) / len(cyc[2:]), sum(dur[2:]) / len(dur[2:])
    print(f"GRBM_GUI_ACTIVE/8 = {c/8:.0f} cycles, duration {d/1e3:.1f} us -> {c/8/d:.3f} GHz")
else:
    print("no data", len(cyc), len(dur))
PY
