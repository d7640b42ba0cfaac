#!/bin/bash
# Development aid: build a libeuler_hip variant with extra -D flags for ONE source file into tools/micro/lib_ablate/
#   usage: variant_file.sh FILE(.hip, without extension) NAME -DFLAG=1 ...   ->  tools/micro/lib_ablate/libeuler_hip_NAME.so   (use with EULER_HIP_LIB=...)
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/micro/lib_ablate
mkdir -p $OUT
FILE=$1; NAME=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Iinclude -Ieuler_amd/csrc "$@" \
   -c euler_amd/csrc/$FILE.hip -o $OUT/${FILE}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libeuler_hip_$NAME.so $(ls euler_amd/csrc/obj/*.o | grep -v "/$FILE.o") $OUT/${FILE}_$NAME.o
rm $OUT/${FILE}_$NAME.o
