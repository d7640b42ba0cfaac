#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -k "multilevel or coarse or two_level or closed_box or cut_off" > gpurun_out/r04_run4_tests.txt 2>&1
tail -3 gpurun_out/r04_run4_tests.txt
for N in 8192 1024; do
timeout 300 python bench.py --size $N --precond ic0_tile_mg --tol 1e-6 --max-iterations 20000 --no-secondary --no-pmc --no-strong --no-cpu-baseline --steps 3 --warmup 1 --profile-all 2>/dev/null > gpurun_out/r04_mg_$N.json
python - <<P
import json
d=json.load(open('bench_full.json'))
print($N, d['value'], d['pcg_iteration'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items() if k in ('apply_a','precond_tile','coarse_cycle')})
P
done
