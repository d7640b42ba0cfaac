#!/bin/bash
# kernel trace + PMC traffic of the roofline mode at 8192^2 (per instantiation: k_search_apply PMODE 1 / 2), and the interior path on / off
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
bash tools/profile_run.sh 8192 2 half_tank ic0_tile > $O/prof_8192_tile.txt 2>&1
cp gpurun_out/prof_8192_half_tank_ic0_tile/summary.md $O/prof_8192_tile_summary.md
python bench.py --no-pmc --no-secondary --no-cpu-baseline --steps 2 > $O/b_interior_on.json 2>/dev/null
EULER_NO_INTERIOR=1 python bench.py --no-pmc --no-secondary --no-cpu-baseline --steps 2 > $O/b_interior_off.json 2>/dev/null
python - <<'P'
import json
for n in ('on','off'):
    d=json.load(open('gpurun_out/r03/b_interior_%s.json'%n))
    print(n, d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})
P
cat $O/prof_8192_tile_summary.md | head -70
