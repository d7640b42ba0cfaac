"""Round 6 soak: rounds 1-5's stage kernels (EULER_OPT_MARKERS_TWO_PASS / _BUILD_TWO_PASS / _VELOCITY_TWO_PASS / _NO_TILE_MAP) against round 6's fused ones, two handles of one process stepping
side by side for hundreds of frames - the same bits in u, v, the count grids, the marker array and the pressure at every checkpoint.  (The test suite does this for 30-40 frames;
the lean zero_bounds and the velocity update's skipped zero stores rest on an invariant - every sample without the fluid property, every wall's sample is zero at the end of
a substep - that a long run with sources, sinks and splashes exercises far more.)

    python tools/r06/forms_soak.py [frames]     -> one JSON line per scenario"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CASES = [("waterfall", 640, 512, ea.PRECOND_IC0_TILE, 100), ("dam_break", 1024, 1024, ea.PRECOND_IC0_TILE, 100), ("waterfall", 768, 640, ea.PRECOND_IC0_TILE_MG, 4000),
         ("dam_break", 512, 768, ea.PRECOND_IC0, 100)]


def digest(sim):
    h = hashlib.sha1()
    for f in (ea.F_U, ea.F_V, ea.F_UTMP, ea.F_VTMP, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_MARKERS, ea.F_PRESSURE):
        h.update(np.ascontiguousarray(sim.get(f)).tobytes())
    return h.hexdigest()[:16]


for scn, X, Y, precond, maxit in CASES:
    kw = dict(precond=precond, dot_mode=ea.DOT_TREE, max_iterations=maxit)
    a = ea.Simulation(X, Y, **kw).load_text(getattr(scenarios, scn)(), upscale=True)
    b = ea.Simulation(X, Y, **kw).load_text(getattr(scenarios, scn)(), upscale=True)
    for key in (ea.OPT_MARKERS_TWO_PASS, ea.OPT_BUILD_TWO_PASS, ea.OPT_VELOCITY_TWO_PASS, ea.OPT_NO_TILE_MAP):
        b.set_option(key, 1)
    t0 = time.perf_counter()
    checks, first_diff = 0, None
    for f in range(FRAMES):
        a.step(); b.step()
        if (f + 1) % 20 == 0 or f == FRAMES - 1:
            checks += 1
            if digest(a) != digest(b) and first_diff is None:
                first_diff = f
    sa, sb = a.stats(), b.stats()
    print(json.dumps({"scenario": scn, "grid": [X, Y], "precond": precond, "frames": FRAMES, "checkpoints": checks, "first_difference_at_frame": first_diff,
                      "substeps": [sa.total_substeps, sb.total_substeps], "pcg_iterations": [sa.total_pcg_iterations, sb.total_pcg_iterations],
                      "markers": [sa.n_markers, sb.n_markers], "dt_events": sa.marker_dt_events, "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    a.close(); b.close()
