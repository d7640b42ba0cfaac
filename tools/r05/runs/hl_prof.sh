#!/bin/bash
export TMPDIR=/tmp
ROOT=$1; shift
cd /tmp
cat > /tmp/hl_probe.py <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import euler_amd as ea
sim = ea.Simulation(8192, 8192, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=100, tol=0.0).load_half_tank()
for f in range(2):
    sim.step()
print("ok", sim.stats().total_pcg_iterations)
PY
rm -rf /tmp/hlp; timeout 200 rocprofv3 --kernel-trace -d /tmp/hlp -o t -- python3 /tmp/hl_probe.py > /tmp/hlp.log 2>&1 < /dev/null
tail -1 /tmp/hlp.log
timeout 60 python3 $ROOT/tools/r05/kstats.py /tmp/hlp ${1:-k_search_apply k_precond_tile} < /dev/null | head -6
