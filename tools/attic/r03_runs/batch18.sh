#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
bash tools/profile_run.sh 1024 4 dam_break ic0 > $O/prof_1024_ic0.txt 2>&1; cp gpurun_out/prof_1024_dam_break_ic0/summary.md $O/prof_1024_dam_break_ic0_summary.md
bash tools/profile_run.sh 4096 4 waterfall ic0_tile > $O/prof_4096_tile.txt 2>&1; cp gpurun_out/prof_4096_waterfall_ic0_tile/summary.md $O/prof_4096_waterfall_ic0_tile_summary.md
bash tools/profile_run.sh 16384 1 dam_break ic0_tile > $O/prof_16384_dam_tile.txt 2>&1; cp gpurun_out/prof_16384_dam_break_ic0_tile/summary.md $O/prof_16384_dam_break_ic0_tile_summary.md
head -20 $O/prof_1024_dam_break_ic0_summary.md; grep -E "k_search|k_precond" $O/prof_16384_dam_break_ic0_tile_summary.md | head
