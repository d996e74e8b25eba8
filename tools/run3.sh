cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s 2>&1 | grep -v "Warning\|warn" | tail -25 > gpurun_out/t_train.txt
for cfg in "fp16 fp32" "fp16 fp16"; do
  timeout 600 python tests/bench_train_step.py 32 5 0 $cfg 2>&1 | grep "train step" 
done > gpurun_out/train_step_r05.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o out --output-format csv -- python $GRAFT_REPO_ROOT/tests/bench_train_step.py 32 3 0 fp16 fp16 > $GRAFT_REPO_ROOT/gpurun_out/train_kt.log 2>&1)
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/train_kernel_stats_fp16.csv; rm -rf gpurun_out/prof_train
cat gpurun_out/t_train.txt gpurun_out/train_step_r05.txt; head -16 gpurun_out/train_kernel_stats_fp16.csv | cut -c1-150
