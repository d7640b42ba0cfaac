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
e=d["equal_residual"]
for k in ("two_level","multilevel"):
    b=e[k]; print(k, b["budget_for_equal_residual"], b.get("frames_at_that_budget",{}).get("value"), b.get("converged_frames"))
print(e["pressure_error_vs_converged"])
print(json.dumps(d["secondary"]["time_to_solution"]))
P
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee $O/full_gpu_suite.txt | tail -12
