#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_gputest_head.txt
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 400 gpurun_out/r02_bench_default.json
