#!/bin/bash
# end-of-round check on the GPU box: smoke, the whole GPU suite, the default bench line
cd "$(dirname "$0")/../.."
O=gpurun_out/r03
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python -m pytest tests -m gpu -q ) 2>&1 | tail -12 > $O/gputest.txt; cat $O/gputest.txt
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_default.json') if l.startswith('{')][-1])
print('value %.4g ms %.1f'%(d['value'], d['ms_per_step']), d['pcg_iteration'])
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','frac_traffic','traffic_over_algorithmic','avg_launch_us','measured_copy_GBps')})
e=d['equal_residual']; print('equal', e.get('tile_budget_for_equal_residual'), e.get('frames_at_that_budget',{}).get('value'), e['reference_ic0_100_iterations'], e['tile_100_iterations'])
st=d['strong_16384_dam_break']; print('strong', st['value'], st['pcg_iteration']['us_per_iteration'], st['roofline']['frac'])
s=d['secondary']
print('exact', s['exact_ic0']['value'], s['exact_ic0']['pcg_iteration'])
print('16384', s['projection_16384']['value'], s['projection_16384']['pcg_iteration']['us_per_iteration'], s['projection_16384']['pcg_iteration']['frac_active'])
c=s['configs1_1024_dam_break']; print('c1', c['value'], c['pcg_iteration']['us_per_iteration'], c['roofline_mode_value'], c['parity_in_run'])
print(s['time_to_solution']['speedup_tile_over_exact'])
P
