#!/bin/bash
# A/B library of the duo GEMM translation unit: tools/build_duo_variant.sh <tag> -DNAME=VALUE ...  -> sprc_amd/libsprc_hip_<tag>.so (SPRC_LIB_PATH)
set -e
tag=$1; shift
R=$(cd $(dirname $0)/.. && pwd); B=$R/sprc_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-int-to-pointer-cast -fno-slp-vectorize -DSPRC_DUO_FAST "$@" -I$R/include \
    -c $R/sprc_amd/csrc/gemm_duo.hip -o $B/gemm_duo_$tag.o
objs=$(ls $B/*.hip.o | grep -v "/gemm_duo.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sprc_amd/libsprc_hip_$tag.so $objs $B/gemm_duo_$tag.o
echo $R/sprc_amd/libsprc_hip_$tag.so
