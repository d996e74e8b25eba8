#!/bin/bash
# quick baseline: headline bench line + per-kernel breakdown of the single-stream step  ->  gpurun_out/<tag>/
tag=${1:-base}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall 2>/dev/null | tail -1 > $O/bench.json
head -c 700 $O/bench.json; echo
bash tools/trace_top.sh --pipeline 0 --qf-streams 1 --no-recall > $O/trace_top.txt 2>&1
rm -rf $R/gpurun_out/trace_top/kt
head -45 $O/trace_top.txt
