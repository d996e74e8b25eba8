#!/bin/bash
# A/B of --vit-overlap (two ViT batches in flight on two streams) on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_overlap; mkdir -p $O; cd $R
for v in ${1:-1 2 1 2}; do
  echo "--vit-overlap $v" | tee -a $O/ab.txt
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall --vit-overlap $v 2>$O/err_$v.txt | tail -1 > $O/line_$v.json
  python -c "import json,sys; d=json.load(open('$O/line_$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['roofline']['launches'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})" | tee -a $O/ab.txt || tail -5 $O/err_$v.txt
done
