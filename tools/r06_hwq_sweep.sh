#!/bin/bash
# GPU_MAX_HW_QUEUES (ROCm default 4: HIP streams beyond that share hardware queues) on the pipelined bench step, one box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for round in 1 2 3; do for v in "" "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=3"; do
env $v python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s' % '$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['kernels']['gemm_bf16']['ms_per_step'], round(d['ms_per_step']/d['kernels']['gemm_bf16']['ms_per_step'],4))" | tee -a $O/hwq_sweep.txt
done; done
echo "--- next box" >> $O/hwq_sweep.txt
