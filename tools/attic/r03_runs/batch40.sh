#!/bin/bash
cd "$(dirname "$0")/../../.."
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tile_precond.py tests/test_slab_rows.py -m gpu -q -x 2>&1 | tail -6 ) 2>&1 | tail -9
bash tools/r03/runs/batch39.sh
