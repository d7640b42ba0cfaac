#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -k "two_level" 2>&1 | tail -15 > $O/t_two_level.txt; cat $O/t_two_level.txt
bash tools/profile_run.sh 8192 2 half_tank ic0_tile2 --no-strong > $O/prof_two.log 2>&1
sed -n 1,24p gpurun_out/prof_8192_half_tank_ic0_tile2/summary.md
tail -12 gpurun_out/prof_8192_half_tank_ic0_tile2/summary.md
