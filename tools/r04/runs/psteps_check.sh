#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tile_precond.py tests/test_gpu_resident.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/psteps_tests.txt 2>&1
tail -5 gpurun_out/psteps_tests.txt
for v in 2 4 2 4; do
  echo "== EULER_P_STEPS=$v"
  EULER_P_STEPS=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<'P'
import json
d=json.load(open('bench_full.json'))
k=d['kernels']
print({n:(r['avg_us'], r['launches']) for n,r in k.items()}, d['pcg_iteration']['us_per_iteration'], d['pcg_iteration']['bytes_per_cell_iteration'], d['value'], d['ms_per_step'])
P
done
