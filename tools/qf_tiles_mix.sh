# the Q-Former's launches (split-precision ones included) under each forced tile configuration: tools/qf_tiles_mix.sh (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/qf_tiles; mkdir -p $O
for cfg in "0 1" "1 1" "1 0" "2 1" "2 0" "4 1"; do
  set -- $cfg
  echo "== SPRC_GEMM_TILE=$1 SPRC_GEMM_RING=$2" | tee -a $O/out.txt
  SPRC_GEMM_TILE=$1 SPRC_GEMM_RING=$2 timeout 300 python tools/qf_shapes.py 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
done
