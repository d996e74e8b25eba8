R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
: > $O/args_sweep.txt
for round in 1 2; do for v in "" "--vit-streams 2" "--qf-streams 1" "--qf-group 2" "--qf-group 4"; do
python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall --no-extra --no-power $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-18s' % '$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'])" | tee -a $O/args_sweep.txt
done; done
