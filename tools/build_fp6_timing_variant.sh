#!/bin/bash
# Timing-only variant of the split-precision GEMM translation unit: the correction segments' MX MFMAs issued in the fp6 (e2m3) format on the
# first 24 of each 32 operand bytes (half the passes; WRONG numbers) -> sprc_amd/libsprc_hip_fp6t.so (SPRC_LIB_PATH).  tools/r06_ab_fp6.sh runs the A/B.
set -e
R=$(cd $(dirname $0)/.. && pwd); B=$R/sprc_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize -DSPRC_MX_FP6_TIMING=1 -I$R/include \
    -c $R/sprc_amd/csrc/gemm_f16e.hip -o $B/gemm_f16e_fp6t.o
objs=$(ls $B/*.hip.o | grep -v "/gemm_f16e.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sprc_amd/libsprc_hip_fp6t.so $objs $B/gemm_f16e_fp6t.o
echo $R/sprc_amd/libsprc_hip_fp6t.so
