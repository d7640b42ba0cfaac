#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
python - <<'P' 2>&1 | tee $O/parity_cases.txt
import sys, json
sys.path.insert(0, '.')
import bench, euler_amd as ea
from euler_amd import scenarios
libs = bench.build_native_oracle()
for e in bench.parity_vs_reference(ea, scenarios, libs["strict"], 0, ea.DOT_TREE, 0):
    print(json.dumps(e))
P
