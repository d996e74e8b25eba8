# A/B of the streaming attention kernels on the ViT shapes: SPRC_ATTN_DMA=0 (register staging) / 2 / 3 (DMA ring of 2 / 3 tiles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3
for r in 1 2; do
  for dh in 88 64; do
  echo -n "DMA=0 NW=3 T=257 "; SPRC_ATTN_DMA=0 python tools/attn_one.py 128 16 257 $dh 30 2>/dev/null
  echo -n "DMA=2 NW=3 T=257 "; SPRC_ATTN_DMA=2 python tools/attn_one.py 128 16 257 $dh 30 2>/dev/null
  echo -n "DMA=2 NW=3 T=256 "; SPRC_ATTN_DMA=2 python tools/attn_one.py 128 16 256 $dh 30 2>/dev/null
  echo -n "DMA=2 NW=3 T=260 "; SPRC_ATTN_DMA=2 python tools/attn_one.py 128 16 260 $dh 30 2>/dev/null
  echo -n "DMA=2 NW=3 T=261 "; SPRC_ATTN_DMA=2 python tools/attn_one.py 128 16 261 $dh 30 2>/dev/null
  done
done
