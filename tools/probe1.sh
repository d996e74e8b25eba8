# round-2 probe: attention counters (resident vs streaming kernel), GEMM per-shape table, tile-order sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in 0 1; do
  SPRC_ATTN_STREAM=$s bash tools/pmc_kernel.sh pmc_attn_s$s attn -- python $GRAFT_REPO_ROOT/tools/attn_one.py 128 16 257 88 6 > gpurun_out/pmc_attn_s$s.txt 2>&1
done
SH="32896,4224,1408 32896,1408,1408,f32,res 32896,6144,1408,gelu 32896,1408,6144,f32,res 32896,9216,1408 14912,2304,768 14912,768,768,f32,res 14912,3072,768,gelu 14912,768,3072,f32,res 8192,8192,8192"
python tools/gemm_shapes.py $SH > gpurun_out/shapes_base.txt 2>&1
for o in 2 3 5 6 8; do SPRC_GEMM_ORDER=$o python tools/gemm_shapes.py 32896,4224,1408 32896,6144,1408,gelu 32896,1408,6144,f32,res 32896,9216,1408 > gpurun_out/shapes_order$o.txt 2>&1; done
SPRC_GEMM_TILE=2 python tools/gemm_shapes.py 32896,1408,1408,f32,res 14912,2304,768 14912,768,768,f32,res 14912,3072,768,gelu 14912,768,3072,f32,res > gpurun_out/shapes_tile2.txt 2>&1
SPRC_GEMM_TILE=4 python tools/gemm_shapes.py 32896,1408,1408,f32,res 14912,2304,768 14912,768,768,f32,res 14912,3072,768,gelu 14912,768,3072,f32,res > gpurun_out/shapes_tile4.txt 2>&1
tail -n 30 gpurun_out/pmc_attn_s0.txt gpurun_out/pmc_attn_s1.txt; head -20 gpurun_out/shapes_*.txt
