#!/bin/bash
# the multi-rank paths after the ghost-row rework: slab tests (gloo ranks sharing the GPU, RCCL with one rank), tile tests again
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests/test_slab_rows.py tests/test_slab.py tests/test_gpu_tile_precond.py -m gpu -q -x 2>&1 | tail -25 > $O/slab_tests.txt; cat $O/slab_tests.txt
python bench.py --force-slab --no-pmc --no-secondary --no-cpu-baseline --steps 2 > $O/b_forceslab.json 2> $O/b_forceslab.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/b_forceslab.json'))
print('force-slab (1 rank RCCL)', d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})
P
tail -3 $O/b_forceslab.err
