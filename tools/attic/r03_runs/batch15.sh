#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_tile_precond.py tests/test_slab.py tests/test_compat_shim.py -m gpu -q -x ) 2>&1 | tail -12 > $O/gputest_parity.txt; cat $O/gputest_parity.txt
run() { python bench.py --no-pmc --no-secondary --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4g' % d['value'], d['pcg_iteration']['us_per_iteration'], d['pcg_iteration']['frac_active'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
echo "8192 half tank ic0: $(run --precond ic0 --steps 2)"
echo "16384 dam break ic0: $(run --precond ic0 --size 16384 --workload dam_break --steps 2)"
echo "1024 dam break ic0: $(run --precond ic0 --size 1024 --workload dam_break --steps 4)"
