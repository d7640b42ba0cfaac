#!/bin/bash
# Development aid: build libeuler_hip variants with extra -D flags for k_pcg.hip into tools/micro/lib_ablate/
#   usage: variant.sh NAME -DFLAG=1 ...   ->  tools/micro/lib_ablate/libeuler_hip_NAME.so   (use with EULER_HIP_LIB=...)
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/micro/lib_ablate
mkdir -p $OUT
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Iinclude -Ieuler_amd/csrc "$@" \
   -c euler_amd/csrc/k_pcg.hip -o $OUT/k_pcg_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libeuler_hip_$NAME.so $(ls euler_amd/csrc/obj/*.o | grep -v k_pcg.o) $OUT/k_pcg_$NAME.o
rm $OUT/k_pcg_$NAME.o
