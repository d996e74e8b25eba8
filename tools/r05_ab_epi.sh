#!/bin/bash
# A/B of the wide (16-B) stores in the 16-bit GEMM epilogues: same box, same build, SPRC_EPI_WIDE=0 vs 1
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_epi; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
for w in 0 1 0 1; do
  echo "SPRC_EPI_WIDE=$w" | tee -a $O/ab.txt
  SPRC_EPI_WIDE=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])" | tee -a $O/ab.txt
done
bash tools/r05_base.sh ab_epi/wide
