#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -s -k "two_level" 2>&1 | tail -30 > $O/t_two_level.txt; cat $O/t_two_level.txt
