#!/bin/bash
# timing-only ablations of k_search_apply's re-reads (WRONG results in the variants)
for v in "" nowindow noedge neither ""; do
  if [ -n "$v" ]; then export EULER_HIP_LIB=$PWD/tools/micro/lib_ablate/libeuler_hip_$v.so; else unset EULER_HIP_LIB; fi
  echo "== variant '${v:-production}'"
  timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<'P'
import json
d=json.load(open('bench_full.json'))
k=d['kernels']
print({n:(r['avg_us'], r['launches']) for n,r in k.items()}, d['pcg_iteration']['us_per_iteration'], d['value'])
P
done
