#!/bin/bash
# does the GEMM's memory-side traffic matter?  Same build, same box: the n-tile group size of the 256x256 kernel's tile order
# (SPRC_GEMM_ORDER: W-resident groups of 2 / 4 (default) / 8 n-tiles) changes the A-panel re-reads and nothing else.  Per setting: GEMM-class
# time of the single-stream bench step (HIP events) and FETCH_SIZE (counter-only pass, x2: gfx950 correction) per step.
# Writes gpurun_out/traffic_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/traffic_ab; mkdir -p $O
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-recall --pipeline 0 --qf-streams 1"
for ord in 4 8 2 4; do
  export SPRC_GEMM_ORDER=$ord
  python bench.py --steps 10 --warmup 3 --prof-every 2 $ARGS 2>/dev/null | tail -1 > $O/bench_$ord.json
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_$ord -o out --output-format csv -- python $R/bench.py --steps 3 --warmup 1 $ARGS > $O/pmc_$ord.log 2>&1)
  python - <<PY
import csv, glob, json
d = json.loads(open("$O/bench_$ord.json").read())
kb = 0.0; n = 0
for f in glob.glob("$O/pmc_$ord/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] and "sprc" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            kb += float(r["Counter_Value"]); n += 1
print("SPRC_GEMM_ORDER=$ord  step %.2f ms  GEMM class %.2f ms (%s TFLOP/s)  FETCH_SIZE x2 = %.1f GB per step over %d GEMM dispatches per step" % (
      d["ms_per_step"], d["kernels"]["gemm_bf16"]["ms_per_step"], d["kernels"]["gemm_bf16"]["tflops"], 2 * 1024 * kb / 4 / 1e9, n // 4))
PY
  rm -rf $O/pmc_$ord
done > gpurun_out/traffic_ab.txt 2>&1
cat gpurun_out/traffic_ab.txt
