#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) 2>&1 | tee $O/full_gpu_suite.txt | tail -12
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"))
e=d["equal_residual"]
for k in ("two_level","multilevel"):
    b=e[k]; print(k, b["budget_for_equal_residual"], b.get("frames_at_that_budget",{}).get("value"), b.get("converged_frames"))
print(d["strong_16384_dam_break"]["value"], d["strong_16384_dam_break"]["converged_frames_multilevel"])
P
