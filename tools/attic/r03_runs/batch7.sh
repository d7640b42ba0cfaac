#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "16384" 2>&1 | tail -8 > $O/t16384.txt; cat $O/t16384.txt
timeout 900 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -s -k "per_baseline" 2>&1 | tail -25 > $O/tparity.txt; cat $O/tparity.txt
