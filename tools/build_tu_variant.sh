#!/bin/bash
# A/B variant library: the named translation units rebuilt with extra -D flags, everything else from the current objects
# -> sprc_amd/libsprc_hip_<tag>.so (SPRC_LIB_PATH).  Usage: tools/build_tu_variant.sh <tag> "<tu1.hip tu2.hip ...>" -DNAME=VALUE ...
set -e
tag=$1; tus=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); B=$R/sprc_amd/csrc/build
objs=$(ls $B/*.hip.o); extra=""
for tu in $tus; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize "$@" -I$R/include \
      -c $R/sprc_amd/csrc/$tu -o $B/${tu%.hip}_$tag.o &
  objs=$(echo "$objs" | grep -v "/$tu.o"); extra="$extra $B/${tu%.hip}_$tag.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sprc_amd/libsprc_hip_$tag.so $objs $extra
echo $R/sprc_amd/libsprc_hip_$tag.so
