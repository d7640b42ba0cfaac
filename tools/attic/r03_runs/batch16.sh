#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
python bench.py --precond ic0 --size 16384 --workload half_tank --no-pmc --no-secondary --no-cpu-baseline --steps 1 > $O/bench_16384_half_tank_ic0.json 2>/dev/null
python bench.py --precond ic0 --size 16384 --workload dam_break --no-pmc --no-secondary --no-cpu-baseline --steps 2 > $O/bench_16384_dam_break_ic0.json 2>/dev/null
bash tools/profile_run.sh 8192 1 half_tank ic0 > $O/prof_8192_ic0.txt 2>&1; cp gpurun_out/prof_8192_half_tank_ic0/summary.md $O/prof_8192_ic0_summary.md
for f in bench_16384_half_tank_ic0 bench_16384_dam_break_ic0; do python - $f <<'P'
import json,sys
d=json.loads([l for l in open('gpurun_out/r03/%s.json'%sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1], '%.4g' % d['value'], d['substeps'], d['pcg_iterations'], d['pcg_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})
P
done
grep -E "k_sweep|k_search|k_precond|k_update" $O/prof_8192_ic0_summary.md | head -12
