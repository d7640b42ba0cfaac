#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
python tools/r03/debug_parity.py > $O/debug_parity.txt 2>&1; cat $O/debug_parity.txt
timeout 900 python -m pytest tests/test_slab_rows.py -m gpu -q -x -k "snapshot" 2>&1 | tail -15 > $O/tsnap.txt; cat $O/tsnap.txt
timeout 600 python -m pytest tests/test_snapshot.py tests/test_gpu_tile_precond.py -m gpu -q -x 2>&1 | tail -5
echo "== sweep ablations"
for v in "" lone sleep4 sleep16 noring; do
  if [ -n "$v" ]; then export EULER_HIP_LIB=$PWD/tools/micro/lib_ablate/libeuler_hip_$v.so; else unset EULER_HIP_LIB; fi
  echo "-- variant ${v:-default}"
  python tools/sweep_timeline.py 1024x1024 8192x8192 8192x64 2>&1 | grep -v "^$"
done > $O/sweep_ablate.txt 2>&1
unset EULER_HIP_LIB
cat $O/sweep_ablate.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16384" 2>&1 | tail -8 > $O/t16384.txt; cat $O/t16384.txt
