#!/bin/bash
# end-of-round check on the GPU box: smoke, the whole GPU suite, the default bench line
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.txt 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r02_gputest_head.txt
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -2 gpurun_out/r02_smoke.txt; cat gpurun_out/r02_gputest_head.txt; tail -c 300 gpurun_out/r02_bench_default.json
