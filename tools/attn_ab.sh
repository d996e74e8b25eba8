cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/attn_ab.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5 > gpurun_out/attn_tests.log
for r in 1 2; do for s in 0 1; do
  SPRC_ATTN_STREAM=$s python tools/attn_one.py 128 16 257 88 20 2>/dev/null >> gpurun_out/attn_ab.log
  SPRC_ATTN_STREAM=$s python tools/attn_one.py 128 16 257 64 20 2>/dev/null >> gpurun_out/attn_ab.log
done; done
SPRC_ATTN_STREAM=1 bash tools/pmc_kernel.sh pmc_attn_v2 attn -- python $GRAFT_REPO_ROOT/tools/attn_one.py 128 16 257 88 6 > gpurun_out/pmc_attn_v2.txt 2>&1
cat gpurun_out/attn_tests.log gpurun_out/attn_ab.log; tail -32 gpurun_out/pmc_attn_v2.txt
