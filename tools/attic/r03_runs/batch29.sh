#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
# 16384^2 in the multilevel mode, solves to the reference's tolerance: half tank and dam break (configs[3]'s grid) on one GPU
for wl in half_tank dam_break; do
timeout 1200 python bench.py --size 16384 --workload $wl --steps 1 --warmup 1 --tol 1e-6 --max-iterations 20000 --no-pmc --no-strong --no-cpu-baseline --no-secondary --precond ic0_tile_mg > $O/bench_16384_${wl}_mg.json 2> $O/bench_16384_${wl}_mg.err; tail -2 $O/bench_16384_${wl}_mg.err
python - <<P
import json
d=json.load(open("gpurun_out/r03/bench_16384_${wl}_mg.json"))
print("$wl", d["value"], d["ms_per_step"], d["config"].get("substeps"), d["config"].get("pcg_iterations"), json.dumps(d.get("pcg_iteration")), d.get("last_residual"))
P
done
