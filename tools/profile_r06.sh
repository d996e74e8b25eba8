#!/bin/bash
# Round-6 measurement artifacts, all from ONE box and ONE build (run through gpurun; copy gpurun_out/r06/* to profiles/r06_*):
#   bench_n1.json             un-profiled `python bench.py --steps 20 --warmup 5` line (headline: ViT-g fp16, split-precision Q-Former)
#   bench_n1_bf16.json, bench_n1_fp16_single.json   the same step in bf16 and in fp16 without the split-precision Q-Former (same box)
#   bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of `bench.py --steps 3 --warmup 1 --no-cpu-baseline` (one stream)
#   traffic.json              FETCH_SIZE / WRITE_SIZE counter passes over the same command (GEMM bytes per step) + kernel_source_sha
#   pmc.json                  SQ / GRBM / TCC counter passes summarised per kernel class
#   bench_vitL_{bf16,fp16,fp8}.json   config C5's backbone on the C2-shaped step;  bench_c5_slice_fp8.json  one GPU's share of C5, 200 timed steps
#   bench_n2_one_gpu.json     `bench.py --gpus 2` on this 1-GPU box (two ranks sharing the GPU, gloo): plumbing check only
#   trace_top.txt             per (kernel, grid) breakdown of the single-stream step
#   blas_ref_fp16.txt         hipBLASLt (torch.matmul) on the model's GEMM shapes next to sprc_gemm: what the vendor library reaches here
#   gemm_split_bench.txt      split-precision product: plain / three fp16 segments (ABI 3) / fp16 + e4m3 segments (ABI 4)
#   train_step.txt            one training step (train mode: dropout on), fp16 frozen trunk, batch 32, fp16 products (the reference's autocast arithmetic)
#   qf_group_ab.txt           bench.py --qf-group 1 / 4 on this box;  qf_shapes.txt  the Q-Former's launches timed alone
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
bash tools/profile_bench.sh r06/pb --pipeline 0 --qf-streams 1 --no-recall --no-extra --no-power > $O/profile_bench.log 2>&1      # one stream: a kernel's duration is its own
cp $O/pb/kernel_stats.csv $O/bench_kernel_stats.csv; cp $O/pb/traffic.json $O/traffic.json
cp $O/traffic.json profiles/r06_traffic.json        # bench.py reads roofline.traffic from here (same box, same kernel sources)
python bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log > $O/bench_n1.json
python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 > $O/bench_n1_bf16.json
SPRC_X3_OFF=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 > $O/bench_n1_fp16_single.json
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extra --no-power 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])"; done > $O/bench_repeat.txt
bash tools/pmc_kernel.sh r06/pmc "gemm_anti=gemm_anti_kernel,gemm_128=gemm_kernel,attention=attn_,layernorm=layernorm_kernel" -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-extra --no-power --qf-streams 1 > $O/pmc.log 2>&1
cp $O/pmc/summary.json $O/pmc.json
for dt in bf16 fp16 fp8; do python bench.py --backbone pretrain_vitL --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-power 2>/dev/null | tail -1 > $O/bench_vitL_$dt.json; done
python bench.py --workload c5-slice --backbone pretrain_vitL --dtype fp8 --steps 200 --warmup 5 2>/dev/null | tail -1 > $O/bench_c5_slice_fp8.json
python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-recall 2>/dev/null | grep '^{' | tail -1 > $O/bench_n2_one_gpu.json
bash tools/trace_top.sh --pipeline 0 --qf-streams 1 --no-recall --no-extra --no-power > $O/trace_top.txt 2>&1
python tools/blas_ref.py fp16 2>&1 | grep -v amdgpu > $O/blas_ref_fp16.txt
python tools/gemm_split_bench.py 14912,768,3072 14912,768,768 7456,768,768 4096,3072,768 4096,768,3072 32896,9216,1408 2>&1 | grep -v amdgpu > $O/gemm_split_bench.txt
for pr in fp32 fp16; do python tests/bench_train_step.py 32 5 2 fp16 $pr 2>&1 | grep "train step, HIP"; done > $O/train_step.txt
for g in 1 4; do echo "--qf-group $g"; python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-recall --no-extra --no-power --qf-group $g 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'])"; done > $O/qf_group_ab.txt
python tools/qf_shapes.py 2>&1 | grep -v amdgpu.ids > $O/qf_shapes.txt
rm -rf $O/pb/kt $O/pb/pmc_* $O/pmc/p? $O/pmc/kt $R/gpurun_out/trace_top/kt
head -c 600 $O/bench_n1.json; echo; cat $O/bench_repeat.txt; tail -20 $O/pmc.log
SPRC_TRACE_GALLERY=1 TMPDIR=/tmp timeout 900 python tools/c2_e2e.py 2>&1 | grep -v "Warning\|^/opt\|it/s\|warn" | grep "trace\|c2_e2e" > $O/c2_e2e.txt
cat $O/c2_e2e.txt
