#!/bin/bash
export EULER_HIP_LIB=$PWD/tools/micro/lib_ablate/libeuler_hip_salds.so
for l in 0 45000 60000 100000; do
  echo "== dynamic LDS $l bytes per block of k_search_apply"
  EULER_EXP_SA_LDS=$l timeout 300 python bench.py --steps 5 --warmup 2 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<'P'
import json
d=json.load(open('bench_full.json'))
k=d['kernels']
print({n:(r['avg_us']) for n,r in k.items() if n in ('apply_a','precond_tile')}, d['pcg_iteration']['us_per_iteration'], d['value'])
P
done
