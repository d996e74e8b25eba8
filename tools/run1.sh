cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "flat_1e3 or c2_size_recall" -s 2>&1 | grep -v Warning | tail -15 > gpurun_out/t_h16.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "c5_full" -s 2>&1 | tail -8 > gpurun_out/t_c5.txt
timeout 600 python tools/c5_rank_full.py > gpurun_out/c5_tool.txt 2>&1
SPRC_GEMM_DUO=1 timeout 300 python tools/duo_check.py check 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/duo_check.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_a.json 2> gpurun_out/bench_r05_a.err
cat gpurun_out/t_h16.txt gpurun_out/t_c5.txt gpurun_out/c5_tool.txt gpurun_out/duo_check.txt; tail -c 3000 gpurun_out/bench_r05_a.json; tail -5 gpurun_out/bench_r05_a.err
