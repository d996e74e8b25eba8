#!/bin/bash
# Product library vs an A/B variant library (sprc_amd/libsprc_hip_<tag>.so), same box, alternating.  Usage: tools/r06_ab_lib.sh <tag> [bench args]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/lib_ab_$tag.txt
for v in $tag prod $tag prod $tag prod; do
  echo "library: $v" | tee -a $O/lib_ab_$tag.txt
  if [ $v = prod ]; then unset SPRC_LIB_PATH; else export SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_$tag.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['kernels']['attention']['ms_per_step'])" | tee -a $O/lib_ab_$tag.txt
done
