set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5 > gpurun_out/attn_tests.log
for s in 0 1; do
  SPRC_ATTN_STREAM=$s python tools/attn_one.py 128 16 257 88 20 >> gpurun_out/attn_ab.log 2>&1
  SPRC_ATTN_STREAM=$s python tools/attn_one.py 128 16 257 64 20 >> gpurun_out/attn_ab.log 2>&1
done
for s in 0 1; do
  SPRC_ATTN_STREAM=$s python tools/attn_one.py 128 16 257 88 20 >> gpurun_out/attn_ab.log 2>&1
done
cat gpurun_out/attn_tests.log gpurun_out/attn_ab.log
