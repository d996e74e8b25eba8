#!/bin/bash
# A/B of --qf-group (Q-Former stage once per G steps) on one box: tools/r05_ab_group.sh "1 4 1 4"
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_group; mkdir -p $O; cd $R
for g in ${1:-1 4}; do
  echo "--qf-group $g" | tee -a $O/ab.txt
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall --qf-group $g 2>$O/err_$g.txt | tail -1 > $O/line_$g.json
  python -c "import json,sys; d=json.load(open('$O/line_$g.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['roofline']['launches'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})" | tee -a $O/ab.txt || tail -5 $O/err_$g.txt
done
