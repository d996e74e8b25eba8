#!/bin/bash
# bench.py's default (GPU_MAX_HW_QUEUES=2, set by the script) against the runtime's default (4) forced from the environment, one box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for round in 1 2 3; do for v in "" "GPU_MAX_HW_QUEUES=4"; do
env $v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s' % '$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], d['config'].get('hw_queues'))" | tee -a $O/hwq_ab.txt
done; done
