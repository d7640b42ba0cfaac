import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import euler_amd as ea
from golden_util import SCENARIOS, load, scenario_text

# (a) a closed box completely full of water: every fluid cell is cut off from the air - A is singular (pure Neumann)
W, H = 40, 30
rows = ["X" * W] + ["X" + "0" * (W - 2) + "X" for _ in range(H - 2)] + ["X" * W]
full = "\n".join(rows) + "\n"
# (b) a box full of water with one enclosed pocket next to an open pool
rows2 = [r for r in rows]
for y in range(8, 16):
    rows2[y] = rows2[y][:10] + "X" + " " * 8 + "X" + rows2[y][20:]
rows2[7] = rows2[7][:10] + "X" * 10 + rows2[7][20:]
rows2[16] = rows2[16][:10] + "X" * 10 + rows2[16][20:]
pocket = "\n".join(rows2) + "\n"
cases = [("closed full box", full), ("box with an air pocket", pocket)] + [(s, scenario_text(load(s + "_frames.npz"))) for s in SCENARIOS]
for name, text in cases:
    for pc, pn in ((ea.PRECOND_IC0, "ic0"), (ea.PRECOND_IC0_TILE, "tile"), (ea.PRECOND_IC0_TILE2, "two-level"), (ea.PRECOND_IC0_TILE_MG, "multilevel")):
        sim = ea.Simulation(320, 256, dot_mode=ea.DOT_TREE, precond=pc, max_iterations=3000).load_text(text, upscale=True)
        worst = 0.0; its = 0
        for f in range(40):
            sim.step()
            st = sim.stats()
            worst = max(worst, st.last_residual)
        u = sim.get(ea.F_U); p = sim.get(ea.F_PRESSURE)
        print("%-24s %-10s finite %s worst residual %.3g iterations %d max|u| %.3g fluid %d" % (name, pn, bool(np.isfinite(u).all() and np.isfinite(p).all()), worst, st.total_pcg_iterations, float(np.abs(u[np.isfinite(u)]).max()) if np.isfinite(u).any() else -1, st.fluid_cells), flush=True)
        sim.close()
