#!/bin/bash
# Several A/B variant libraries (sprc_amd/libsprc_hip_<tag>.so) against the product library, same box, round-robin, three rounds.
# Usage: tools/r06_ab_multi.sh <outname> <tag> [<tag> ...]
out=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/$out.txt
for round in 1 2 3; do
for v in prod "$@"; do
  if [ $v = prod ]; then unset SPRC_LIB_PATH; else export SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % '$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['kernels']['attention']['ms_per_step'], [ (k, v['ms_per_step']) for k, v in d['kernels'].items() if k not in ('gemm_bf16','attention')])" | tee -a $O/$out.txt
done; done
