#!/bin/bash
# Round-2 measurement artifacts, all from ONE box and ONE build (run through gpurun; copy gpurun_out/r02/* to profiles/):
#   bench_n1.json             un-profiled `python bench.py --steps 20 --warmup 5` line (headline: ViT-g bf16, N = 1)
#   bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`
#   traffic.json              FETCH_SIZE / WRITE_SIZE counter passes over the same command (GEMM bytes per step) + kernel_source_sha
#   pmc.json                  SQ / GRBM / TCC counter passes over the same command, summarised per kernel class: MFMA busy
#                             fraction (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE), effective clock, LDS bank conflicts, wait
#                             fractions, HBM-side GB/s of the GEMM / attention / LayerNorm kernels
#   bench_vitL_bf16.json, bench_vitL_fp8.json   config C5's backbone: bf16 vs fp8 (e4m3fn) lines
#   bench_n2_one_gpu.json     `bench.py --gpus 2` on this 1-GPU box (two ranks sharing the GPU, gloo): plumbing check only
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
bash tools/profile_bench.sh r02/pb > $O/profile_bench.log 2>&1
cp $O/pb/kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/pb/traffic.json $O/traffic.json
cp $O/traffic.json profiles/r02_traffic.json        # bench.py reads roofline.traffic from here (same box, same kernel sources)
python bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log > $O/bench_n1.json
bash tools/pmc_kernel.sh r02/pmc "gemm_anti=gemm_anti_kernel,gemm_128=gemm_kernel,attention=attn_,layernorm=layernorm_kernel" -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc.log 2>&1
cp $O/pmc/summary.json $O/pmc.json
for dt in bf16 fp8; do python bench.py --backbone pretrain_vitL --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_vitL_$dt.json; done
python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_n2_one_gpu.json
# drop the bulky raw counter CSVs from what gets merged back
rm -rf $O/pb/kt $O/pb/pmc_* $O/pmc/p? $O/pmc/kt
head -c 400 $O/bench_n1.json; echo; tail -40 $O/pmc.log
