#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -s -k "multilevel or two_level" 2>&1 | tail -25 > $O/t_mg.txt; cat $O/t_mg.txt
