#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -k "two_level" 2>&1 | tail -5
( time timeout 1500 python bench.py --no-pmc --no-strong --no-cpu-baseline > $O/bench_err.json 2> $O/bench_err.err ) 2>&1 | tail -4
tail -2 $O/bench_err.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_err.json"))
e=d["equal_residual"]
print(json.dumps(e["pressure_error_vs_converged"], indent=1))
print(e["tile_budget_for_equal_residual"], e["two_level"]["budget_for_equal_residual"], e["two_level"].get("frames_at_that_budget",{}).get("value"))
P
