#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tile_precond.py -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tail -9
for g in 0 1; do
  if [ $g = 1 ]; then export EULER_BUILD_GATHER=1; fi
  timeout 600 python bench.py --no-pmc --no-strong --no-cpu-baseline --no-secondary > $O/bench_bs_$g.json 2> $O/bench_bs_$g.err
  python - <<P
import json
d=json.load(open("gpurun_out/r03/bench_bs_$g.json"))
k=d["kernels"]
print("gather" if $g else "lds", d["value"], d["ms_per_step"], {n:k[n]["avg_us"] for n in k if "build" in n or "assembl" in n}, list(k.keys())[:30])
P
done
