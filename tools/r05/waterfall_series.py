#!/usr/bin/env python3
"""configs[4]'s scenario (waterfall: sources and sinks active) with every solve to 1e-6: PCG iterations per solve over the run, multilevel against tile-local.
usage: waterfall_series.py [N] [steps] [window] [modes ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import euler_amd as ea
from euler_amd import scenarios
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
WIN = int(sys.argv[3]) if len(sys.argv) > 3 else 100
modes = sys.argv[4:] or ["mg"]
for mode in modes:
    pc = {"mg": ea.PRECOND_IC0_TILE_MG, "tile": ea.PRECOND_IC0_TILE, "two": ea.PRECOND_IC0_TILE2}[mode]
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=pc, max_iterations=20000, pcg_poll_interval=16, resident=ea.RESIDENT_OFF).load_text(scenarios.waterfall(), upscale=True)
    t0 = time.perf_counter()
    it0 = sub0 = 0
    worst = 0
    for f in range(STEPS):
        s.step()
        st = s.stats()
        worst = max(worst, st.last_pcg_iterations / max(st.last_substeps, 1))
        if (f + 1) % WIN == 0:
            torch.cuda.synchronize()
            print(json.dumps(dict(mode=mode, step=f + 1, its_per_solve=round((st.total_pcg_iterations - it0) / max(st.total_substeps - sub0, 1), 1), worst_frame=round(worst, 1),
                                  substeps=int(st.total_substeps - sub0), fluid=int(st.fluid_cells), markers=int(st.n_markers), residual=st.last_residual,
                                  seconds=round(time.perf_counter() - t0, 1))), flush=True)
            it0, sub0, worst = st.total_pcg_iterations, st.total_substeps, 0
    s.close()
