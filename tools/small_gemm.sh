cd $GRAFT_REPO_ROOT
SH="4096,2304,768 4096,768,768,f32,res 4096,3072,768,gelu 4096,768,3072,f32,res 14912,768,768,f32,res 7456,768,768,f32,res 128,4224,1408 128,1408,1408,f32,res 4096,256,768,f32"
for t in 0 1 2; do echo "TILE=$t"; SPRC_GEMM_TILE=$t python tools/gemm_shapes.py $SH 2>/dev/null; done
