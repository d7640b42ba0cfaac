#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -k "multilevel" 2>&1 | tail -5
bash tools/profile_run.sh 8192 2 half_tank ic0_tile_mg --no-strong > /tmp/prof_mg.log 2>&1
sed -n 7,24p gpurun_out/prof_8192_half_tank_ic0_tile_mg/summary.md
grep -n "value\|us_per_iteration" gpurun_out/prof_8192_half_tank_ic0_tile_mg/summary.md | head -5
