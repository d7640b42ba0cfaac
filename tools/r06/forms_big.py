"""Round 6: the default forms (fused stages, the tile map with its 64-row workgroups on large grids) against rounds 1-5's full passes at a BASELINE size, two handles side by
side - the bit-exact tests at 8192^2 run ONE substep from a fresh load, where the tile map is not yet valid; this walks frames.

    python tools/r06/forms_big.py [N] [workload] [frames]     -> a line per frame: which fields differ (none should)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workload = sys.argv[2] if len(sys.argv) > 2 else "half_tank"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sims = []
for old in (False, True):
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=20)
    if workload == "half_tank":
        s.load_half_tank()
    else:
        s.load_text(getattr(scenarios, workload)(), upscale=True)
    if old:
        for k in (ea.OPT_MARKERS_TWO_PASS, ea.OPT_BUILD_TWO_PASS, ea.OPT_VELOCITY_TWO_PASS, ea.OPT_NO_TILE_MAP):
            s.set_option(k, 1)
    sims.append(s)
a, b = sims
bad = 0
for f in range(frames):
    a.step(); b.step()
    out = []
    for name, fld in (("u", ea.F_U), ("v", ea.F_V), ("utmp", ea.F_UTMP), ("vtmp", ea.F_VTMP), ("count", ea.F_COUNT), ("prev_count", ea.F_PREV_COUNT), ("p", ea.F_PRESSURE), ("markers", ea.F_MARKERS)):
        x, y = a.get(fld), b.get(fld)
        same = x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))
        bad += 0 if same else 1
        out.append("%s:%s" % (name, "=" if same else "DIFF"))
        del x, y
    sa, sb = a.stats(), b.stats()
    print("%d %s frame %d: %s | substeps %d %d, iterations %d %d, markers %d %d" % (N, workload, f, " ".join(out), sa.last_substeps, sb.last_substeps, sa.last_pcg_iterations, sb.last_pcg_iterations,
                                                                                  sa.n_markers, sb.n_markers), flush=True)
print("differences:", bad)
sys.exit(1 if bad else 0)
