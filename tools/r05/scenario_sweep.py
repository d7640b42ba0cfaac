#!/usr/bin/env python3
"""The reference's five scenarios (as stored with the golden fixtures) + the two generated ones, upscaled to N^2, every solve to 1e-6 in the multilevel mode: PCG iterations per
solve over the run (windows), so that a scene whose geometry defeats the coarse correction shows up (round 5: spray drops did, until the Jacobi steps were damped per node).
usage: scenario_sweep.py [N] [steps] [window]"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import euler_amd as ea
from euler_amd import scenarios
from golden_util import load as gload, scenario_text
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
WIN = int(sys.argv[3]) if len(sys.argv) > 3 else 100
texts = {n: scenario_text(gload(n + "_frames.npz")) for n in ("basic", "block", "filter", "waterfall", "weird-edges")}
texts["dam_break (generated)"] = scenarios.dam_break()
texts["waterfall (generated)"] = scenarios.waterfall()
for name, text in texts.items():
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, max_iterations=20000, pcg_poll_interval=16).load_text(text, upscale=True)
    t0 = time.perf_counter()
    it0 = sub0 = 0
    wins = []
    for f in range(STEPS):
        s.step()
        if (f + 1) % WIN == 0:
            st = s.stats()
            wins.append(round((st.total_pcg_iterations - it0) / max(st.total_substeps - sub0, 1), 1))
            it0, sub0 = st.total_pcg_iterations, st.total_substeps
    st = s.stats()
    print(json.dumps(dict(scenario=name, N=N, steps=STEPS, its_per_solve_by_window=wins, fluid_cells=int(st.fluid_cells), markers=int(st.n_markers), residual=st.last_residual,
                          seconds=round(time.perf_counter() - t0, 1))), flush=True)
    s.close()
