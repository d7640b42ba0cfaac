#!/bin/bash
# per-kernel times of the multilevel mode on a small grid (1024^2 dam break in its solving phase)
export TMPDIR=/tmp
ROOT="$(cd "$(dirname "$0")/../../.." && pwd)"
cd /tmp
cat > /tmp/small_probe.py <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import euler_amd as ea
from euler_amd import scenarios
s = ea.Simulation(1024, 1024, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, max_iterations=20000, pcg_poll_interval=16).load_text(scenarios.dam_break(), upscale=True)
for f in range(${1:-60}):
    s.step()
st = s.stats(); print("iterations", st.total_pcg_iterations, "substeps", st.total_substeps)
PY
rm -rf /tmp/sp; timeout 200 rocprofv3 --kernel-trace -d /tmp/sp -o t -- python3 /tmp/small_probe.py > /tmp/sp.log 2>&1 < /dev/null
tail -1 /tmp/sp.log
timeout 60 python3 $ROOT/tools/r05/kstats.py /tmp/sp < /dev/null | head -${2:-24}
