#!/bin/bash
# k_precond_tile<16, true, 2> with / without the stores of the tiles' partial sums (WRONG results without): what the 160 scattered doubles per tile cost
export TMPDIR=/tmp
ROOT="$(cd "$(dirname "$0")/../../.." && pwd)"
cd /tmp
for v in mg_base pt_nostore; do
  rm -rf /tmp/abl_$v
  EULER_HIP_LIB=$ROOT/tools/micro/lib_ablate/libeuler_hip_$v.so timeout 150 rocprofv3 --kernel-trace -d /tmp/abl_$v -o t -- python3 $ROOT/tools/r05/mg_probe.py ${1:-8192} 1 40 > /tmp/abl_$v.log 2>&1 < /dev/null
  echo "== $v: $(grep frame /tmp/abl_$v.log | tail -1)"
  timeout 60 python3 $ROOT/tools/r05/kstats.py /tmp/abl_$v k_mg_down1 k_search_apply "k_precond_tile<16, true" < /dev/null
done
