# A/B of the LayerNorm-fused residual add (SPRC_FUSE_ADD) on the bench workload + the parity tests that cover it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "f16_delta or fused_residual or layernorm or gemm_bf16" 2>&1 | tail -4 > gpurun_out/fuse_tests.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -s 2>&1 | grep -E "bf16\]|fp32\]|passed|failed|Error" >> gpurun_out/fuse_tests.log
for v in 0 1 0 1; do
  SPRC_FUSE_ADD=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('FUSE_ADD=$v', d['value'], 'img/s', d['ms_per_step'],'ms', 'gemm', k['gemm_bf16']['ms_per_step'], k['gemm_bf16']['tflops'], 'attn', k['attention']['ms_per_step'], 'rowops', k['rowops']['ms_per_step'], k['rowops']['alg_GBs'])" >> gpurun_out/fuse_ab.log
done
cat gpurun_out/fuse_tests.log gpurun_out/fuse_ab.log
