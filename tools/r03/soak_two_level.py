#!/usr/bin/env python3
"""Development aid: what the preconditioner modes do to a long run.  2048^2 waterfall 120 frames and 1024^2 dam break 300 frames in the
parity mode (reference IC(0)), the tile-local mode and the two-level mode at the reference's cap of 100 iterations, and with the cap lifted
(solves run to the reference's tolerance 1e-6): the converged runs are the yardstick the capped ones are read against."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import euler_amd as ea
from euler_amd import scenarios

print("| grid | scenario | mode | iteration cap | frames | substeps | PCG iterations | solves that hit the cap | markers | fluid cells | max abs u | wall |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for size, wl, frames in ((2048, "waterfall", 120), (1024, "dam", 300)):
    rows = ((ea.PRECOND_IC0, "parity", 100), (ea.PRECOND_IC0_TILE, "roofline", 100), (ea.PRECOND_IC0_TILE2, "two-level", 100),
            (ea.PRECOND_IC0_TILE2, "two-level", 20000), (ea.PRECOND_IC0, "parity", 20000), (ea.PRECOND_IC0_TILE_MG, "multilevel", 100), (ea.PRECOND_IC0_TILE_MG, "multilevel", 20000))
    if len(sys.argv) > 1:
        rows = [r for r in rows if r[1] in sys.argv[1:]]
    for pc, name, cap in rows:
        sim = ea.Simulation(size, size, dot_mode=ea.DOT_TREE, precond=pc, tile_records=16, max_iterations=cap, pcg_poll_interval=8 if cap == 100 else 32)
        sim.load_text(scenarios.dam_break() if wl == "dam" else scenarios.waterfall(), upscale=True)
        t0 = time.time()
        capped = 0
        for f in range(frames):
            sim.step()
            st = sim.stats()
            capped += int(st.last_residual > 1e-6)
        u = sim.get(ea.F_U)
        assert np.isfinite(u).all() and np.isfinite(sim.get(ea.F_V)).all()
        print("| %d^2 | %s | %s | %d | %d | %d | %d | %d frames | %d | %d | %.1f | %.1f s |" % (size, wl, name, cap, st.frames, st.total_substeps, st.total_pcg_iterations,
              capped, st.n_markers, st.fluid_cells, np.abs(u).max(), time.time() - t0), flush=True)
        sim.close()
