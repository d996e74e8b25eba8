# A/B of GEMM library variants on the ViT-g / Q-Former shapes: tools/ab_gemm.sh tagA tagB ...   ("main" = libsprc_hip.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SH="32768,6144,1408,bf16,gelu 32768,4224,1408 32768,1408,1408,f32,res 32768,1408,6144,f32,res 14336,2304,768 14336,3072,768,bf16,gelu 8192,8192,8192"
for rep in 1 2; do
for tag in "$@"; do
  if [ $tag = main ]; then unset SPRC_LIB_PATH; else export SPRC_LIB_PATH=$R/sprc_amd/libsprc_hip_$tag.so; fi
  echo "== $tag (rep $rep)"
  timeout 300 python tools/gemm_shapes.py $SH 2>&1 | tail -8
done
done
