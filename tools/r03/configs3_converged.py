#!/usr/bin/env python3
"""Development aid: BASELINE configs[3]'s scenario (16384^2 dam break) on ONE GPU in the multilevel mode with every solve run to the reference's
tolerance - frames until the time budget is used up, a line every 20 frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, max_iterations=20000, pcg_poll_interval=32).load_text(scenarios.dam_break(), upscale=True)
print("| frames | wall s | substeps | PCG iterations | iterations per solve (last 20 frames) | worst residual (last 20) | fluid cells | markers |")
print("|---|---|---|---|---|---|---|---|")
t0 = time.time(); f = 0; last_it = 0; last_sub = 0; worst = 0.0
while time.time() - t0 < budget_s:
    sim.step(); f += 1
    st = sim.stats()
    worst = max(worst, st.last_residual)
    if f % 20 == 0:
        print("| %d | %.1f | %d | %d | %.0f | %.2g | %d | %d |" % (f, time.time() - t0, st.total_substeps, st.total_pcg_iterations,
              (st.total_pcg_iterations - last_it) / max(1, st.total_substeps - last_sub), worst, st.fluid_cells, st.n_markers), flush=True)
        last_it, last_sub, worst = st.total_pcg_iterations, st.total_substeps, 0.0
