#!/bin/bash
# A/B library: rebuild the bf16 GEMM translation unit with extra -D flags and link it with the current objects of everything
# else into sprc_amd/libsprc_hip_<tag>.so (load it with SPRC_LIB_PATH=...).  Usage: tools/build_variant.sh <tag> -DNAME=VALUE ...
set -e
tag=$1; shift
R=$(cd $(dirname $0)/.. && pwd); B=$R/sprc_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize "$@" -I$R/include \
    -c $R/sprc_amd/csrc/gemm.hip -o $B/gemm_$tag.o
objs=$(ls $B/*.hip.o | grep -v "/gemm.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sprc_amd/libsprc_hip_$tag.so $objs $B/gemm_$tag.o
echo $R/sprc_amd/libsprc_hip_$tag.so
