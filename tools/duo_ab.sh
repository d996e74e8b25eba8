# A/B of the duo GEMM kernel against the anti-phase kernel + timing ablations: tools/duo_ab.sh   (writes gpurun_out/duo_ab.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
SPRC_GEMM_DUO=1 timeout 300 python tools/duo_check.py check 2>&1 | grep -v "Warning\|amdgpu.ids"
SPRC_GEMM_DUO=0 timeout 300 python tools/duo_check.py time 2>&1 | grep -v "Warning\|amdgpu.ids"
SPRC_GEMM_DUO=1 timeout 300 python tools/duo_check.py time 2>&1 | grep -v "Warning\|amdgpu.ids"
for v in 1 2 3 4; do
  echo "== ablation $v (1 = no refill loads, 2 = no barrier, 4 = one workgroup per CU)"
  SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_abl$v.so SPRC_GEMM_DUO=1 timeout 300 python tools/duo_check.py time 2>&1 | grep -v "Warning\|amdgpu.ids"
done
} > gpurun_out/duo_ab.txt 2>&1
tail -70 gpurun_out/duo_ab.txt
