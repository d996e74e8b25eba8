#!/bin/bash
# One-stop profile of `bench.py` on the GPU box; writes gpurun_out/<tag>/ :
#   bench.json                 the bench line (un-profiled run)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats summary (per-kernel calls / total / average)
#   traffic.json               per-kernel-class HBM-side bytes per launch from two --pmc passes (FETCH_SIZE, WRITE_SIZE)
# Counter passes carry no tracing options (the pool refuses --pmc together with API traces).
# Usage: tools/profile_bench.sh <tag> [bench args...]
tag=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 10 --warmup 2 $* > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o out --output-format csv -- python $R/bench.py $ARGS > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d $O/pmc_$c -o out --output-format csv -- python $R/bench.py $ARGS > $O/pmc_$c.log 2>&1
done
python - <<EOF
import csv, glob, json, collections
def cls(name):
    if "gemm" in name and "sprc" in name: return "gemm"
    if "attn" in name: return "attention"
    if "sprc::" in name: return "rowops+rank"
    return None
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = cls(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    out[c] = {k: {"kb_total": v[0], "dispatches": v[1], "kb_per_dispatch": v[0] / max(v[1], 1)} for k, v in acc.items()}
g_f = out["FETCH_SIZE"].get("gemm", {}); g_w = out["WRITE_SIZE"].get("gemm", {})
import sys
sys.path.insert(0, "$R")
import bench
summary = {
    "kernel_source_sha": bench.kernel_source_sha(),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over 'bench.py $ARGS' (separate passes, counters only)",
    "units": "counters are KiB; FETCH_SIZE doubled (gfx950 counts the 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated)",
    "raw": out,
    "steps_profiled": 4,
    "gemm_bytes_per_step": {
        "fetch_corrected": 2.0 * 1024.0 * g_f.get("kb_total", 0.0) / 4,
        "write": 1024.0 * g_w.get("kb_total", 0.0) / 4,
    },
}
summary["gemm_bytes_per_step"]["total"] = summary["gemm_bytes_per_step"]["fetch_corrected"] + summary["gemm_bytes_per_step"]["write"]
json.dump(summary, open("$O/traffic.json", "w"), indent=1)
print(json.dumps(summary["gemm_bytes_per_step"]))
EOF
head -12 $O/kernel_stats.csv | cut -c1-200
cat $O/bench.json | cut -c1-300
