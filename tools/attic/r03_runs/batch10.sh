#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
run() { python bench.py --no-pmc --no-secondary --no-cpu-baseline --steps 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.4g' % d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
echo "default:      $(run)"
echo "force-slab:   $(run --force-slab)"
echo "force-slab torch comm: $(run --force-slab --comm torch)"
done
for st in 0 4096 65536 73728 1048576 1118208 2162688; do echo "stagger $st: $(EULER_ARRAY_STAGGER=$st run)"; done
