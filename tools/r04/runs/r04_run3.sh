#!/bin/bash
mkdir -p gpurun_out
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.log ) 2> gpurun_out/r04_bench_default.time
wc -c gpurun_out/r04_bench_default.json; tail -2 gpurun_out/r04_bench_default.log; cat gpurun_out/r04_bench_default.time
cp bench_full.json gpurun_out/r04_bench_full.json 2>/dev/null
timeout 900 python tools/r04/configs1_500.py 500 > gpurun_out/r04_configs1_500.txt 2>&1
tail -8 gpurun_out/r04_configs1_500.txt
