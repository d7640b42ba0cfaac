#!/usr/bin/env python3
"""Soak run (development aid): hundreds of frames at 1024^2 / 2048^2, checking that no in-kernel wait ever times out
and every field stays finite."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import euler_amd as ea
from euler_amd import scenarios
for size, wl, frames in ((1024, "dam", 300), (2048, "waterfall", 120)):
    sim = ea.Simulation(size, size, dot_mode=ea.DOT_TREE)
    sim.load_text(scenarios.dam_break() if wl == "dam" else scenarios.waterfall(), upscale=True)
    t0 = time.time()
    for f in range(frames):
        sim.step()
    st = sim.stats()
    u = sim.get(ea.F_U)
    print(size, wl, "frames", st.frames, "substeps", st.total_substeps, "iters", st.total_pcg_iterations, "markers", st.n_markers,
          "fluid", st.fluid_cells, "finite", bool(np.isfinite(u).all()), "max|u| %.2f" % np.abs(u).max(), "%.1f s" % (time.time() - t0))
    sim.close()
