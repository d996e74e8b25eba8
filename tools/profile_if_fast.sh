#!/bin/bash
# Box lottery: run the round's profile script only on a box whose headline step is at or under the threshold (ms); always records the probe.
thr=${1:-87.5}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
ms=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "probe: $ms ms per step (threshold $thr)" | tee -a $O/box_probe.txt
if python -c "import sys; sys.exit(0 if float('$ms') <= float('$thr') else 1)"; then bash tools/profile_r06.sh; else echo "slow box: profile skipped"; fi
