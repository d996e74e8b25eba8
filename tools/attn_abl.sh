cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -1
for d in 0 0; do echo -n "  dh88 "; python tools/attn_one.py 128 16 257 88 20 2>/dev/null; done
echo -n "  dh64 "; python tools/attn_one.py 128 16 257 64 20 2>/dev/null
