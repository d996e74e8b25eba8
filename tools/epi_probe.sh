# epilogue cost vs number of active CUs (is the fp32+residual epilogue an HBM burst or issue-bound?): through gpurun
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SPRC_GEMM_DEBUG=64 SPRC_GEMM_TILE=4 SPRC_GEMM_STAMP_MASK=0x801
for cfg in "2048 1024 1408 8" "4096 4096 1408 100" "8192 8192 1408 520" "32768 1408 1408 520"; do
  set -- $cfg
  echo "== M=$1 N=$2 K=$3 timed WG $4"
  SPRC_GEMM_STAMP_WG=$4 timeout 120 python tools/gemm_stamp.py $1 $2 $3 2>&1 | tail -2
done
