cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 0 2 0 2; do
  SPRC_FUSE_ADD=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('FUSE_ADD=$v', d['value'], 'img/s', d['ms_per_step'],'ms', 'gemm', k['gemm_bf16']['ms_per_step'], k['gemm_bf16']['tflops'], 'attn', k['attention']['ms_per_step'], 'rowops', k['rowops']['ms_per_step'])"
done
SPRC_FUSE_ADD=2 timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -s -k "bf16_engine" 2>&1 | grep -E "^\[|passed|failed"
