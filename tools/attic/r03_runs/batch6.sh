#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
echo "== k_search_apply ablations (WRONG results: timing only), roofline mode 8192^2"
for v in "" nowin noedge nowinedge; do
  if [ -n "$v" ]; then export EULER_HIP_LIB=$PWD/tools/micro/lib_ablate/libeuler_hip_$v.so; else unset EULER_HIP_LIB; fi
  python bench.py --no-pmc --no-secondary --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('variant ${v:-default}', '%.4g' % d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done 2>&1 | tee $O/sa_ablate.txt
echo "== sweep helper variants, parity mode 8192^2"
for v in "" ann8 ann32 fet8 fet32 ann16fet16 noring; do
  if [ -n "$v" ]; then export EULER_HIP_LIB=$PWD/tools/micro/lib_ablate/libeuler_hip_$v.so; else unset EULER_HIP_LIB; fi
  python bench.py --precond ic0 --no-pmc --no-secondary --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('variant ${v:-default}', '%.4g' % d['value'], d['pcg_iteration']['us_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done 2>&1 | tee $O/sweep_variants.txt
unset EULER_HIP_LIB
python - <<'P' 2>&1 | tee $O/parity_cases.txt
import sys, json
sys.path.insert(0, '.')
import bench, euler_amd as ea
from euler_amd import scenarios
libs = bench.build_native_oracle()
for e in bench.parity_vs_reference(ea, scenarios, libs["strict"], 0, ea.DOT_TREE, 0):
    print(json.dumps(e))
P
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16384" 2>&1 | tail -8 > $O/t16384.txt; cat $O/t16384.txt
