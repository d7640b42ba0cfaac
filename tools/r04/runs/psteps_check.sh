#!/bin/bash
mkdir -p gpurun_out


for v in 4 8 4 8; do
  echo "== EULER_P_STEPS=$v"
  EULER_P_STEPS=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<'P'
import json
d=json.load(open('bench_full.json'))
k=d['kernels']
print({n:(r['avg_us'], r['launches']) for n,r in k.items()}, d['pcg_iteration']['us_per_iteration'], d['pcg_iteration']['bytes_per_cell_iteration'], d['value'], d['ms_per_step'])
P
done
