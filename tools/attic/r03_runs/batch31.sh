#!/bin/bash
cd "$(dirname "$0")/../../.."
timeout 1500 python -m pytest tests/test_slab_rows.py -m gpu -q -x -k "multilevel" 2>&1 | tail -30
