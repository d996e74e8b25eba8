# timing ablations of the streaming attention kernel (WRONG results by construction): SPRC_ATTN_DEBUG bits
#   1 no global loads in the key loop | 2 no LDS commit | 4 no tile math | 8 no barrier | 16 no output store | 32 no Q load
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 8 16 7 0; do echo -n "debug=$d  "; SPRC_ATTN_DEBUG=$d python tools/attn_one.py 128 16 257 88 30 2>/dev/null; done
