#!/bin/bash
# timing of the V-cycle's three launches with pieces switched off (WRONG results): rocprofv3 kernel trace of one frame per variant
export TMPDIR=/tmp
ROOT="$(cd "$(dirname "$0")/../../.." && pwd)"
cd /tmp
for v in ${2:-base notail notrans noup}; do
  rm -rf /tmp/abl_$v
  EULER_HIP_LIB=$ROOT/tools/micro/lib_ablate/libeuler_hip_mg_$v.so timeout 150 rocprofv3 --kernel-trace -d /tmp/abl_$v -o t -- python3 $ROOT/tools/r05/mg_probe.py ${1:-8192} 1 40 > /tmp/abl_$v.log 2>&1 < /dev/null
  echo "== $v: $(grep frame /tmp/abl_$v.log | tail -1)"
  timeout 60 python3 $ROOT/tools/r05/kstats.py /tmp/abl_$v k_mg_ k_search_apply k_precond_tile < /dev/null
done
