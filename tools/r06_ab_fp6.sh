#!/bin/bash
# What would fp6 (e2m3) correction segments buy?  Same box, alternating: the product library against a TIMING-ONLY variant whose correction-segment
# MFMAs are issued in the fp6 format (half the passes) on the existing bytes -- scores are WRONG in the variant, step time and power are what is read.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/fp6_ab.txt
python tools/gemm_split_bench.py 14912,768,3072 14912,3072,768 2>&1 | grep -v amdgpu | sed 's/^/product lib: /' | tee -a $O/fp6_ab.txt
SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_fp6t.so python tools/gemm_split_bench.py 14912,768,3072 14912,3072,768 2>&1 | grep -v amdgpu | sed 's/^/fp6-timing variant: /' | tee -a $O/fp6_ab.txt
for v in prod fp6t prod fp6t prod fp6t; do
  echo "library: $v" | tee -a $O/fp6_ab.txt
  if [ $v = fp6t ]; then export SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_fp6t.so; else unset SPRC_LIB_PATH; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'])" | tee -a $O/fp6_ab.txt
done
