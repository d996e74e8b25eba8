#!/bin/bash
# The round's kernel changes together, same box, alternating: SPRC_GEMM_DEAD=0 SPRC_ATTN_CROSS=0 (= round 5's code paths in this build) vs the defaults
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/total_ab.txt
for v in r6 r5 r6 r5 r6 r5; do
  echo "paths: $v" | tee -a $O/total_ab.txt
  if [ $v = r5 ]; then export SPRC_GEMM_DEAD=0 SPRC_ATTN_CROSS=0; else unset SPRC_GEMM_DEAD SPRC_ATTN_CROSS; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['kernels']['attention']['ms_per_step'])" | tee -a $O/total_ab.txt
done
