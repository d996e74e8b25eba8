#!/bin/bash
# A/B of the Q-Former cross-attention on one-wave DMA workgroups (SPRC_ATTN_CROSS 0 = resident kernel, 2 / 3 = ring depth): parity tests, the
# launches alone, the bench step alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_e2e_gpu.py tests/test_rerank_gpu.py tests/test_planted_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/cross_tests.txt
cat $O/cross_tests.txt
: > $O/cross_ab.txt
for c in 0 2 3; do SPRC_ATTN_CROSS=$c timeout 120 python tools/qf_attn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/cross_ab.txt; done
for c in 3 0 3 0 2 0; do
  echo "SPRC_ATTN_CROSS=$c" | tee -a $O/cross_ab.txt
  SPRC_ATTN_CROSS=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels'].get('attention', d['kernels']))" | tee -a $O/cross_ab.txt
done
