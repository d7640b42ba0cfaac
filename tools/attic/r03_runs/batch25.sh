#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time timeout 1500 python bench.py --no-pmc --no-strong --no-cpu-baseline > $O/bench_mg.json 2> $O/bench_mg.err ) 2>&1 | tail -4
tail -2 $O/bench_mg.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_mg.json"))
e=d["equal_residual"]
print(json.dumps(e["pressure_error_vs_converged"], indent=1))
for k in ("two_level","multilevel"):
    b=e[k]; print(k, b["budget_for_equal_residual"], b["at_that_budget"], b["at_100_iterations"], b.get("frames_at_that_budget",{}).get("value"), b.get("converged_frames"))
    print("   scan", b["scan"])
print(json.dumps(d["secondary"]["time_to_solution"], indent=1))
P
timeout 600 python bench.py --no-pmc --no-strong --no-cpu-baseline --no-secondary --precond ic0_tile_mg > $O/bench_mg_main.json 2> $O/bench_mg_main.err; tail -2 $O/bench_mg_main.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_mg_main.json"))
print(d["value"], d["ms_per_step"], json.dumps(d.get("pcg_iteration")))
for k,v in d.get("kernels",{}).items(): print(k, v)
P
