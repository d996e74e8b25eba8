#!/bin/bash
# A/B of the persistent tile loop of the 256 x 256 GEMM kernel: previous build / this build one workgroup per tile / this build one per CU
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_persist2; mkdir -p $O; cd $R
run() { python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"; }
for rep in 1 2; do
  echo "== prev build" | tee -a $O/ab.txt; SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_prev.so run | tee -a $O/ab.txt
  echo "== this build, SPRC_GEMM_PERSIST=0" | tee -a $O/ab.txt; SPRC_GEMM_PERSIST=0 run | tee -a $O/ab.txt
  echo "== this build, SPRC_GEMM_PERSIST=1" | tee -a $O/ab.txt; SPRC_GEMM_PERSIST=1 run | tee -a $O/ab.txt
done
SPRC_GEMM_PERSIST=1 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_benchshape_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/ab.txt
