# s_memtime phase timeline of the anti-phase GEMM on the ViT-g shapes: tools/stamp_sweep.sh  (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SPRC_GEMM_DEBUG=64 SPRC_GEMM_TILE=4
for shape in "32768 6144 1408" "8192 8192 8192"; do
  for m in 0xfff 0x3 0x6 0xc 0x18 0x9 0x28 0x120 0x900 0x801; do
    echo "== $shape mask $m"
    SPRC_GEMM_STAMP_MASK=$m timeout 120 python tools/gemm_stamp.py $shape 2>&1 | tail -4
  done
done
