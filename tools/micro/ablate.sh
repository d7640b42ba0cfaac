#!/bin/bash
# Development aid: build libeuler_hip variants with parts of the sweep step knocked out
# (k_pcg.hip SW_ABLATE bits: 1 no result store, 2 no operand loads, 4 no DPP shift, 8 no LDS carry
# write, 16 no polls) into gpurun_out-independent tools/micro/lib_ablate/, then time them with
# tools/sweep_timeline.py on the GPU box:  for v in ...; EULER_HIP_LIB=... python tools/sweep_timeline.py
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/micro/lib_ablate
mkdir -p $OUT
for v in ${@:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Iinclude -Ieuler_amd/csrc \
     -DSW_EXPERIMENT=$v -c euler_amd/csrc/k_pcg.hip -o $OUT/k_pcg_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libeuler_hip_$v.so euler_amd/csrc/obj/driver.o euler_amd/csrc/obj/k_grid.o \
     euler_amd/csrc/obj/k_markers.o $OUT/k_pcg_$v.o euler_amd/csrc/obj/euler_host.o
  rm $OUT/k_pcg_$v.o
done
