# Q-Former GEMM shapes under each tile configuration (dispatcher retuning): tools/qf_tiles.sh  (through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SH="8192,8192,1024 14912,768,768,f32,res 7456,768,768,f32,res 7456,768,768 4096,768,768,f32,res 4096,768,768 4096,2304,768 4096,3072,768,bf16,gelu 4096,768,3072,f32,res 7456,3072,768,bf16,gelu 7456,768,3072,f32,res 14912,2304,768 14912,3072,768,bf16,gelu 14912,768,3072,f32,res 128,6144,1408,bf16,gelu 128,1408,1408,f32,res 128,4224,1408"
for tile in 0 1 2 4; do
  echo "== SPRC_GEMM_TILE=$tile"
  SPRC_GEMM_TILE=$tile timeout 300 python tools/gemm_shapes.py $SH 2>&1 | grep -v amdgpu.ids | tail -n +2
done
