#!/bin/bash
cd "$(dirname "$0")/../../.."
( time python -m pytest tests/test_slab_rows.py tests/test_slab.py tests/test_dist_gloo.py -m gpu -q -x ) 2>&1 | tail -8
