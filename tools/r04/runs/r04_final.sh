#!/bin/bash
# round 4, last GPU call of a batch: the full -m gpu suite with durations and the driver's bench command at HEAD
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r04_suite_final.txt 2>&1
tail -4 gpurun_out/r04_suite_final.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.log ) 2> gpurun_out/r04_bench_default.time
wc -c gpurun_out/r04_bench_default.json; tail -2 gpurun_out/r04_bench_default.log; cat gpurun_out/r04_bench_default.time
cp bench_full.json gpurun_out/r04_bench_full.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.txt 2>&1; tail -2 gpurun_out/r04_smoke.txt
# the rocprofv3 summaries of the same box (kernel trace + the two PMC passes): the headline, configs[1] resident, the multilevel mode
export TMPDIR=/tmp
bash tools/profile_run.sh 8192 3 half_tank ic0_tile > gpurun_out/r04_prof_a.log 2>&1
bash tools/profile_run.sh 1024 6 dam_break ic0_tile > gpurun_out/r04_prof_b.log 2>&1
bash tools/profile_run.sh 8192 2 half_tank ic0_tile_mg "--tol 1e-6 --max-iterations 20000" > gpurun_out/r04_prof_c.log 2>&1
for d in gpurun_out/prof_8192_half_tank_ic0_tile gpurun_out/prof_1024_dam_break_ic0_tile gpurun_out/prof_8192_half_tank_ic0_tile_mg; do rm -rf $d/trace $d/pmc_fetch $d/pmc_write; done
head -14 gpurun_out/prof_8192_half_tank_ic0_tile/summary.md | tail -6
