#!/bin/bash
# A s' recomputed in k_precond_tile: parity tests of everything that runs the tile-local modes, then old vs new timing on the same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_tile_precond.py tests/test_gpu_resident.py tests/test_slab_rows.py tests/test_slab.py -m gpu -q -x > gpurun_out/recompute_tests.txt 2>&1
tail -5 gpurun_out/recompute_tests.txt
for v in 1 0 1 0; do
  echo "== EULER_TILE_STORE_AS=$v"
  EULER_TILE_STORE_AS=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-secondary --no-pmc --no-cpu-baseline 2>/dev/null > /tmp/line.json
  python - <<'P'
import json
d=json.load(open('bench_full.json'))
k=d['kernels']
print({n:(r['avg_us'], r['launches']) for n,r in k.items()}, d['pcg_iteration']['us_per_iteration'], d['value'])
P
done
