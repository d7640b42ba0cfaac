#!/bin/bash
cd "$(dirname "$0")/../../.."
run() { python bench.py --no-pmc --no-secondary --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4g' % d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | head -12
echo "cold, steps 4: $(run --steps 4)"
echo "again steps 4: $(run --steps 4)"
echo "again steps 8: $(run --steps 8)"
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | head -12
sleep 45
echo "after 45 s idle, steps 2: $(run --steps 2)"
echo "force-slab steps 4: $(run --steps 4 --force-slab)"
