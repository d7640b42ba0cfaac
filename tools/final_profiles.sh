#!/bin/bash
# end-of-round refresh of profiles/: every rocprofv3 summary and the single-GPU bench lines of the large BASELINE configs
cd "$(dirname "$0")/.."
bash tools/profile_run.sh 8192 2 half_tank ic0_tile > /dev/null 2>&1
bash tools/profile_run.sh 8192 2 half_tank ic0 > /dev/null 2>&1
bash tools/profile_run.sh 1024 10 dam_break ic0 > /dev/null 2>&1
bash tools/profile_run.sh 4096 4 waterfall ic0_tile > /dev/null 2>&1
bash tools/profile_run.sh 4096 4 waterfall ic0 > /dev/null 2>&1
bash tools/profile_run.sh 16384 1 half_tank ic0_tile > /dev/null 2>&1
for p in ic0_tile ic0; do
  python bench.py --size 16384 --workload dam_break --precond $p --steps 2 --no-secondary --no-pmc > gpurun_out/r02_bench_16384_dam_break_$p.json 2> gpurun_out/r02_bench_16384_dam_break_$p.err
  python bench.py --size 4096 --workload waterfall --precond $p --steps 10 --no-secondary --no-pmc > gpurun_out/r02_bench_4096_waterfall_$p.json 2> gpurun_out/r02_bench_4096_waterfall_$p.err
done
ls gpurun_out/prof_*/summary.md
