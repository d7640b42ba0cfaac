#!/usr/bin/env python3
"""BASELINE configs[1] (1024 x 1024 dam break, 500 steps) with EVERY solve run to the reference's tolerance, multilevel mode: wall time, iterations per solve, and
where the time of a solve goes.  usage: configs1_converged.py [steps]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import euler_amd as ea
from euler_amd import scenarios
N, STEPS = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 500
s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, max_iterations=20000, pcg_poll_interval=16).load_text(scenarios.dam_break(), upscale=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
solves = 0
for f in range(STEPS):
    s.step()
    st = s.stats()
    solves += st.last_substeps if st.last_pcg_iterations else 0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = s.stats()
print(json.dumps(dict(mode="multilevel, f64, every solve to 1e-6", seconds=round(dt, 2), cells_steps_per_s=round(N * N * STEPS / dt), substeps=int(st.total_substeps),
                      pcg_iterations=int(st.total_pcg_iterations), solves_that_iterated=solves, iterations_per_solve=round(st.total_pcg_iterations / max(solves, 1), 1),
                      last_residual=st.last_residual, markers=int(st.n_markers), fluid_cells=int((s.get(ea.F_COUNT) > 0).sum()))))
