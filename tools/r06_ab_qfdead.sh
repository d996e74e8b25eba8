#!/bin/bash
# Last-layer row pruning of the Q-Former passes (SPRC_QF_DEAD 0 -> 1), same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/qf_dead_ab.txt
for v in 1 0 1 0 1 0; do
  echo "SPRC_QF_DEAD=$v" | tee -a $O/qf_dead_ab.txt
  SPRC_QF_DEAD=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['kernels']['attention']['ms_per_step'])" | tee -a $O/qf_dead_ab.txt
done
