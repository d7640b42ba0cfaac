#!/bin/bash
# per-kernel times of the multilevel mode's iteration at N^2 (half tank, fixed iterations): rocprofv3 kernel trace of tools/r05/mg_probe.py
export TMPDIR=/tmp
ROOT="$(cd "$(dirname "$0")/../../.." && pwd)"
cd /tmp
rm -rf /tmp/mgp; timeout 200 rocprofv3 --kernel-trace -d /tmp/mgp -o t -- python3 $ROOT/tools/r05/mg_probe.py ${1:-8192} ${2:-2} ${3:-44} > /tmp/mgp.log 2>&1 < /dev/null
grep frame /tmp/mgp.log | tail -2
timeout 60 python3 $ROOT/tools/r05/kstats.py /tmp/mgp k_mg_ k_search_apply k_precond_tile < /dev/null | head -${4:-12}
