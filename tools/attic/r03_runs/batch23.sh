#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -2 $O/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"))
e=d["equal_residual"]; two=e["two_level"]
print("equal", e["tile_budget_for_equal_residual"], e.get("frames_at_that_budget",{}).get("value"), "two", two["budget_for_equal_residual"], two.get("frames_at_that_budget",{}).get("value"), two["at_100_iterations"])
print(json.dumps(d["secondary"]["time_to_solution"]))
print(d["secondary"]["exact_ic0"]["value"])
P
timeout 900 python bench.py --size 16384 --steps 1 --warmup 0 --no-pmc --no-strong --no-cpu-baseline --no-secondary --precond ic0_tile2 > $O/bench_16384_two.json 2> $O/bench_16384_two.err; tail -2 $O/bench_16384_two.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_16384_two.json"))
print("16384 two-level", d["value"], d["ms_per_step"], json.dumps(d.get("pcg_iteration")))
P
