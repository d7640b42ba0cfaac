#!/bin/bash
# round 3, first contact: the new interior-chunk / two-step-p kernels against the oracle, the copy probe variants, the bench line
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x 2>&1 | tail -15 > $O/tile_tests.txt; cat $O/tile_tests.txt
for mb in 256 1024 4096; do ./tools/micro/copy_bench $mb; done > $O/copy_bench.txt 2>&1
timeout 600 python bench.py --no-pmc > $O/bench_nopmc.json 2> $O/bench_nopmc.err; tail -c 1500 $O/bench_nopmc.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/bench_nopmc.json'))
print(d['value'], d['ms_per_step'], d['pcg_iteration'])
for k,v in d['kernels'].items(): print(k, v.get('avg_us'), v.get('GBps_active'))
s=d['secondary']
for k in s: print(k, {kk:vv for kk,vv in s[k].items() if kk in ('value','pcg_iteration','error','roofline_mode_us_per_iteration','parity_in_run','speedup_tile_over_exact')})
P
