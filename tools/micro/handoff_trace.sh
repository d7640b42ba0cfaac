#!/bin/bash
# Development aid: build libeuler_hip with SW_TRACE_HANDOFF (time stamps of one band-to-band hand-off in
# the sweep timeline), to be used as  EULER_HIP_LIB=tools/micro/lib_ablate/libeuler_hip_trace.so python tools/handoff_trace.py
set -eu
cd "$(dirname "$0")/../.."
OUT=tools/micro/lib_ablate
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Iinclude -Ieuler_amd/csrc \
   -DSW_TRACE_HANDOFF=1 -c euler_amd/csrc/k_pcg.hip -o $OUT/k_pcg_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libeuler_hip_trace.so euler_amd/csrc/obj/driver.o euler_amd/csrc/obj/k_grid.o \
   euler_amd/csrc/obj/k_markers.o $OUT/k_pcg_trace.o euler_amd/csrc/obj/euler_host.o
rm $OUT/k_pcg_trace.o
