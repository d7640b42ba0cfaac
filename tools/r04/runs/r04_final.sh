#!/bin/bash
# round 4, last GPU call of a batch: the full -m gpu suite with durations and the driver's bench command at HEAD
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r04_suite_final.txt 2>&1
tail -4 gpurun_out/r04_suite_final.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.log ) 2> gpurun_out/r04_bench_default.time
wc -c gpurun_out/r04_bench_default.json; tail -2 gpurun_out/r04_bench_default.log; cat gpurun_out/r04_bench_default.time
cp bench_full.json gpurun_out/r04_bench_full.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_smoke.txt 2>&1; tail -2 gpurun_out/r04_smoke.txt
