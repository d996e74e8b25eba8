#!/bin/bash
# Epilogue bias wait on the straight-line path (new library) vs the waits hipcc repeated in every row's block (libsprc_hip_old.so), same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/epi_wait_ab.txt
for v in new old new old new old; do
  echo "library: $v" | tee -a $O/epi_wait_ab.txt
  if [ $v = old ]; then export SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_old.so; else unset SPRC_LIB_PATH; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['kernels']['attention']['ms_per_step'])" | tee -a $O/epi_wait_ab.txt
done
