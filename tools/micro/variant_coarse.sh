#!/bin/bash
# Development aid: build libeuler_hip variants with extra -D flags for k_coarse.hip into tools/micro/lib_ablate/   (usage: variant_coarse.sh NAME -DFLAG=1 ...)
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/micro/lib_ablate
mkdir -p $OUT
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Iinclude -Ieuler_amd/csrc "$@" \
   -c euler_amd/csrc/k_coarse.hip -o $OUT/k_coarse_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libeuler_hip_$NAME.so $(ls euler_amd/csrc/obj/*.o | grep -v k_coarse.o) $OUT/k_coarse_$NAME.o
rm $OUT/k_coarse_$NAME.o
