#!/bin/bash
# the default bench line with round 3's new blocks; the configs[3] tests; the reverse-order experiment; the long runs
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_default.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['pcg_iteration'])
print('roofline', {k:d['roofline'][k] for k in ('kernel','achieved','frac','frac_traffic','traffic_over_algorithmic','avg_launch_us','algorithmic_bytes_per_cell','measured_copy_GBps')})
print('equal_residual', json.dumps(d['equal_residual'])[:1500])
print('strong', json.dumps(d['strong_16384_dam_break'])[:1200])
for e in (d['secondary'].get('parity_vs_reference_ic0') or []): print('parity', e)
print('cpu', json.dumps(d['cpu_baseline'])[:600])
P
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16384" 2>&1 | tail -8 > $O/t16384.txt; cat $O/t16384.txt
timeout 900 python -m pytest tests/test_slab_rows.py -m gpu -q -x -k "configs3" 2>&1 | tail -8 > $O/tslab2048.txt; cat $O/tslab2048.txt
for r in 0 1; do EULER_TILE_REVERSE=$r python bench.py --no-pmc --no-secondary --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('reverse $r', d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; done
python tools/r03/long_runs.py $O/long_runs.md > $O/long_runs.log 2>&1; tail -3 $O/long_runs.log
