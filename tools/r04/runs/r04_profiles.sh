#!/bin/bash
# round 4: the rocprofv3 summaries committed under profiles/ (kernel trace + the two PMC passes per workload, tools/profile_run.sh)
cd "$(dirname "$0")/../../.."
bash tools/profile_run.sh 8192 3 half_tank ic0_tile > gpurun_out/r04_prof_a.log 2>&1
bash tools/profile_run.sh 1024 6 dam_break ic0_tile > gpurun_out/r04_prof_b.log 2>&1
bash tools/profile_run.sh 8192 2 half_tank ic0_tile_mg "--tol 1e-6 --max-iterations 20000" > gpurun_out/r04_prof_c.log 2>&1
for d in gpurun_out/prof_8192_half_tank_ic0_tile gpurun_out/prof_1024_dam_break_ic0_tile gpurun_out/prof_8192_half_tank_ic0_tile_mg; do
  echo "== $d"; head -30 $d/summary.md
  rm -rf $d/trace $d/pmc_fetch $d/pmc_write
done
