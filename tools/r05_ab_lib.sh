#!/bin/bash
# A/B of two builds of the library on one box: tools/r05_ab_lib.sh <tag of sprc_amd/libsprc_hip_<tag>.so>   (B = the current build)
tag=${1:-prev}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_lib_$tag; mkdir -p $O; cd $R
for rep in 1 2; do
for which in $tag main; do
  if [ $which = main ]; then unset SPRC_LIB_PATH; else export SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_$which.so; fi
  echo "== $which (rep $rep)" | tee -a $O/ab.txt
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})" | tee -a $O/ab.txt
  if [ $rep = 1 ]; then
    python tools/gemm_shapes.py 32768,6144,1408,bf16,gelu 32768,4224,1408 32768,1408,1408,f32,res 32768,1408,6144,f32,res 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
    python tools/qf_shapes.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  fi
done
done
unset SPRC_LIB_PATH
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_e2e_gpu.py tests/test_benchshape_gpu.py tests/test_fp8_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
