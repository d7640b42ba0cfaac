#!/bin/bash
for R in 0 1; do
EULER_RESIDENT=$R timeout 600 python bench.py --size 4096 --workload waterfall --no-secondary --no-pmc --no-strong --no-cpu-baseline --steps 20 --warmup 2 --max-preroll 200 2>/dev/null > gpurun_out/r04_waterfall_4096_res$R.json
python - <<P
import json
d=json.load(open('bench_full.json'))
print("EULER_RESIDENT=$R", d['value'], d['ms_per_step'], d['substeps'], d['pcg_iterations'], d['fluid_cells'], d['config']['preroll_frames'], d['pcg_iteration'])
P
done
