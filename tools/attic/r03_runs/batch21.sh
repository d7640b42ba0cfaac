#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 900 python bench.py --no-pmc --no-strong --no-cpu-baseline > $O/bench_two.json 2> $O/bench_two.err; tail -3 $O/bench_two.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_two.json"))
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("equal_residual"), indent=1)[:6000])
print(json.dumps(d["secondary"].get("time_to_solution"), indent=1))
P
timeout 600 python bench.py --no-pmc --no-strong --no-cpu-baseline --no-secondary --precond ic0_tile2 > $O/bench_two_main.json 2> $O/bench_two_main.err; tail -3 $O/bench_two_main.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_two_main.json"))
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"]), json.dumps(d.get("pcg_iteration")))
for k,v in d.get("kernels",{}).items(): print(k, v)
P
