#!/bin/bash
# round 3 records: the default bench line (live PMC passes included), kernel traces + PMC tables of both modes at 8192^2 and of the
# roofline mode at 16384^2, the one-rank RCCL communicator line
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
bash tools/profile_run.sh 8192 2 half_tank ic0_tile > $O/prof_8192_tile.txt 2>&1; cp gpurun_out/prof_8192_half_tank_ic0_tile/summary.md $O/prof_8192_tile_summary.md
bash tools/profile_run.sh 8192 1 half_tank ic0 > $O/prof_8192_ic0.txt 2>&1; cp gpurun_out/prof_8192_half_tank_ic0/summary.md $O/prof_8192_ic0_summary.md
bash tools/profile_run.sh 16384 1 half_tank ic0_tile > $O/prof_16384_tile.txt 2>&1; cp gpurun_out/prof_16384_half_tank_ic0_tile/summary.md $O/prof_16384_tile_summary.md
python bench.py --force-slab --no-pmc --no-secondary --no-cpu-baseline --steps 4 > $O/bench_forceslab_rccl_1rank.json 2> /dev/null
python bench.py --size 16384 --workload dam_break --no-pmc --no-secondary --no-cpu-baseline --steps 2 --precond ic0 > $O/bench_16384_dam_break_ic0.json 2>/dev/null
python bench.py --size 4096 --workload waterfall --no-pmc --no-secondary --no-cpu-baseline --steps 4 > $O/bench_4096_waterfall_tile.json 2>/dev/null
head -c 600 $O/bench_default.json; echo; head -30 $O/prof_8192_tile_summary.md
