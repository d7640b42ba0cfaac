#!/bin/bash
# Development aid: build a libeuler_hip variant with extra -D flags for ONE source file into tools/micro/lib_ablate/ (use with EULER_HIP_LIB=...).
#   usage: variant_file.sh FILE(.hip, without extension) NAME [-DFLAG=1 ...]
# The timing experiments of rounds 3-5 - builds with a piece switched off, WRONG results - are not in the product's sources: tools/micro/ablations/timing_experiments.patch
# puts their #ifdef blocks (SA_ABL_NO_WINDOW, SA_ABL_NO_EDGE, SW_ABL_FORCE_LONE, SW_ABL_NO_RING, RS_POLL_SLEEP, EU_EXP_BIN_NOATOMIC, MG_ABL_NO_TRANS / _TAIL / _UP) back into
# a COPY of csrc/ that this script compiles from whenever a -D flag names one of them.
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/micro/lib_ablate
mkdir -p $OUT
FILE=$1; NAME=$2; shift; shift
SRC=euler_amd/csrc
if echo "$*" | grep -qE "_ABL_|EU_EXP_|RS_POLL_SLEEP"; then
  SRC=$(mktemp -d)/csrc
  mkdir -p $SRC; cp euler_amd/csrc/*.hip euler_amd/csrc/*.h $SRC/
  (cd $SRC/.. && patch -s -p2 < "$OLDPWD/tools/micro/ablations/timing_experiments.patch")
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Iinclude -I$SRC "$@" -c $SRC/$FILE.hip -o $OUT/${FILE}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libeuler_hip_$NAME.so $(ls euler_amd/csrc/obj/*.o | grep -v "/$FILE.o") $OUT/${FILE}_$NAME.o
rm $OUT/${FILE}_$NAME.o
