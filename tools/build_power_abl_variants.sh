#!/bin/bash
# Ablation builds for tools/gemm_power_abl.py: apply tools/gemm_power_abl.patch (SPRC_ANTI_ABL hooks in gemm_impl.hpp), build the fp16 GEMM translation unit
# four times (-DSPRC_ANTI_ABL=1|2|3|4) against the CURRENT objects of everything else -> sprc_amd/libsprc_hip_abl<n>.so, and REVERT the header (the product
# sources, their hash and the product library stay untouched).  WRONG results in these libraries: timing / power only.
set -e
R=$(cd $(dirname $0)/.. && pwd); B=$R/sprc_amd/csrc/build
cd $R && git apply tools/gemm_power_abl.patch
trap "cd $R && git checkout sprc_amd/csrc/gemm_impl.hpp" EXIT
for n in 1 2 3 4; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DSPRC_ANTI_ABL=$n -I$R/include \
        -c $R/sprc_amd/csrc/gemm_f16.hip -o $B/gemm_f16_abl$n.o
    objs=$(ls $B/*.hip.o | grep -v "/gemm_f16.hip.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sprc_amd/libsprc_hip_abl$n.so $objs $B/gemm_f16_abl$n.o; echo built abl$n ) &
done
wait
