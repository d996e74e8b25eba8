#!/bin/bash
# A/B of the dead-wave skeleton in the 256 x 256 GEMM (waves whose 64 columns lie beyond N skip fragment reads, MFMAs and the epilogue):
# same box, same build, SPRC_GEMM_DEAD=0 vs 1; parity tests + race screen first
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/dead_tests.txt
timeout 300 python tools/race_screen.py 10 2>&1 | grep -v amdgpu.ids | tail -8 >> $O/dead_tests.txt
cat $O/dead_tests.txt
: > $O/dead_ab.txt
for w in 1 0 1 0 1 0; do
  echo "SPRC_GEMM_DEAD=$w" | tee -a $O/dead_ab.txt
  SPRC_GEMM_DEAD=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'])" | tee -a $O/dead_ab.txt
done
