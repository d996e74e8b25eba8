cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/traffic_ab.sh
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o out --output-format csv -- python $GRAFT_REPO_ROOT/tests/bench_train_step.py 32 3 0 fp16 > $GRAFT_REPO_ROOT/gpurun_out/train_kt.log 2>&1)
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/train_kernel_stats.csv; head -40 gpurun_out/train_kernel_stats.csv | cut -c1-170
tail -3 gpurun_out/train_kt.log
rm -rf gpurun_out/prof_train
