#!/bin/bash
# timing-only: k_search_apply with a whole run's loads in flight (WRONG results at band boundaries) against the production kernel without its edge rows
for v in "" noedge preload noedge preload ""; do
  if [ -n "$v" ]; then export EULER_HIP_LIB=$PWD/tools/micro/lib_ablate/libeuler_hip_$v.so; else unset EULER_HIP_LIB; fi
  echo "== variant '${v:-production}'"
  timeout 300 python bench.py --steps 4 --warmup 1 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<'P'
import json
d=json.load(open('bench_full.json'))
k=d['kernels']
print({n:(r['avg_us'], r['launches']) for n,r in k.items() if n in ('apply_a','precond_tile')}, d['pcg_iteration']['us_per_iteration'], d['value'])
P
done
