#!/usr/bin/env python3
"""The long runs BASELINE.json names, at full size on one MI355X (VERDICT r2 next #1c): configs[1] = 1024^2 dam break, 500 steps;
configs[4] = 4096^2 waterfall (sources and sinks active), 2000 steps.  Each in the parity mode (the reference's IC(0)), in the
roofline mode (tile-local IC(0), budget 100) and in the roofline mode with a larger iteration budget (the equal-residual budget
bench.py measures: what the tile-local mode needs to reach the residual the reference's IC(0) reaches in 100).  Recorded: wall time,
substeps, PCG iterations, markers, fluid cells, max |u|, the frame at which the source latch fell (main.c:281,290), NaN checks.
usage: long_runs.py [markdown out]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import euler_amd as ea
from euler_amd import scenarios

BUDGET = int(os.environ.get("EULER_EQUAL_RESIDUAL_BUDGET", "132"))
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout


def emit(line):
    print(line, file=out, flush=True)
    if out is not sys.stdout:
        print(line, flush=True)


emit("| config | mode | steps | substeps | PCG iterations | markers (start -> end) | fluid cells | max abs u / v | source latch at frame | finite | wall |")
emit("|---|---|---|---|---|---|---|---|---|---|---|")
for label, size, text, steps in (("configs[1] 1024^2 dam break", 1024, scenarios.dam_break(), 500),
                                 ("configs[4] 4096^2 waterfall", 4096, scenarios.waterfall(), 2000)):
    modes = ((ea.PRECOND_IC0, 100, "parity (reference IC(0), 100)"), (ea.PRECOND_IC0_TILE, 100, "roofline (tile-local, 100)"),
             (ea.PRECOND_IC0_TILE, BUDGET, "roofline (tile-local, %d)" % BUDGET),
             (ea.PRECOND_IC0_TILE_MG, 100, "multilevel (100)"), (ea.PRECOND_IC0_TILE_MG, 20000, "multilevel, every solve to 1e-6"))
    if os.environ.get("EULER_LONG_RUN_MODES"):      # e.g. "multilevel": only the modes whose name starts like that
        modes = [m for m in modes if m[2].startswith(os.environ["EULER_LONG_RUN_MODES"])]
    for pc, budget, name in modes:
        sim = ea.Simulation(size, size, dot_mode=ea.DOT_TREE, precond=pc, max_iterations=budget, pcg_poll_interval=8 if budget <= 200 else 32).load_text(text, upscale=True)
        n0 = sim.stats().n_markers
        latch = None
        t0 = time.time()
        for f in range(steps):
            sim.step()
            if latch is None and f % 10 == 9 and sim.stats().source_exhausted:
                latch = "<= %d" % (f + 1)
        wall = time.time() - t0
        st = sim.stats()
        u, v, mk = sim.get(ea.F_U), sim.get(ea.F_V), sim.get(ea.F_MARKERS)
        finite = bool(np.isfinite(u).all() and np.isfinite(v).all() and np.isfinite(mk).all())
        emit("| %s | %s | %d | %d | %d | %d -> %d | %d | %.1f / %.1f | %s | %s | %.1f s |" % (
            label, name, st.frames, st.total_substeps, st.total_pcg_iterations, n0, st.n_markers, st.fluid_cells, np.abs(u).max(), np.abs(v).max(),
            latch or ("never" if not st.source_exhausted else "?"), finite, wall))
        assert finite
        sim.close()
