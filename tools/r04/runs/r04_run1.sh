#!/bin/bash
# round 4, GPU call 1: the tests this round touched (with durations) and the default bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --durations=25 -k "lean_assembly or 1024_half_tank or diffusion or dye_vs_oracle or ragged or free_running or set_precond_validates or part_file_one_rank or multilevel_mode_on_row_slabs or bench_multi_rank or snapshot or closed_box or cut_off" > gpurun_out/r04_run1_tests.txt 2>&1
tail -3 gpurun_out/r04_run1_tests.txt
( time timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.log ) 2> gpurun_out/r04_bench_default.time
wc -c gpurun_out/r04_bench_default.json; tail -3 gpurun_out/r04_bench_default.log; cat gpurun_out/r04_bench_default.time
cp bench_full.json gpurun_out/r04_bench_full.json 2>/dev/null
