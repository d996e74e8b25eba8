#!/bin/bash
# Environment-switch sweep on one box, round-robin, two rounds.  Usage: tools/r06_env_sweep.sh <outname> "VAR=val [VAR=val]" ...   ("-" = defaults)
out=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/$out.txt
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['kernels']['attention']['ms_per_step'])" | tee -a $O/$out.txt
done; done
