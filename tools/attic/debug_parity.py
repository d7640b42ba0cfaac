import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import euler_amd as ea
from euler_amd import scenarios
import bench
from oracle_lib import build_oracle

so = build_oracle()
sim = ea.Simulation(1024, 1024, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0)
sim.load_text(scenarios.dam_break(), upscale=True)
pre = bench.preroll_into_solves(sim, 400)
print("preroll", pre)
o = bench.oracle_from_sim(sim, ea, so)        # reference IC(0)
ot = bench.oracle_from_sim(sim, ea, so, 16)   # oracle's tile mode
o.step(); ot.step()
sim.set_precond(ea.PRECOND_IC0_TILE, 0)
sim.step()
p, pr, pt = sim.get(ea.F_PRESSURE), o.p, ot.p
for name, q in (("oracle exact", pr), ("oracle tile", pt)):
    d = np.abs(p - q)
    iy, ix = np.unravel_index(np.argmax(d), d.shape)
    print(name, "max|dp|", d.max(), "at", (ix, iy), "gpu", p[iy, ix], "other", q[iy, ix], "pmax", np.abs(q).max(), "cells with |dp|>1e-3 pmax:", int((d > 1e-3 * np.abs(q).max()).sum()))
    print("  count there", sim.get(ea.F_COUNT)[iy-1:iy+2, ix-1:ix+2], "oracle count", o.count[iy-1:iy+2, ix-1:ix+2], "solid", sim.get(ea.F_SOLID)[iy-1:iy+2, ix-1:ix+2])
    print("  p gpu\n", p[iy-2:iy+3, ix-2:ix+3], "\n  p other\n", q[iy-2:iy+3, ix-2:ix+3])
print("du exact", np.abs(sim.get(ea.F_U) - o.u).max(), "du tile", np.abs(sim.get(ea.F_U) - ot.u).max())
print("iters gpu", sim.stats().last_pcg_iterations, "exact", o.c.last_pcg_iterations, "tile", ot.c.last_pcg_iterations)
print("residual gpu", sim.stats().last_residual, "exact", o.c.last_residual, "tile", ot.c.last_residual)
# percentile picture of the relative pressure difference on fluid cells
fl = o.count > 0
rel = np.abs(p - pr)[fl] / np.abs(pr).max()
print("relative dp percentiles (50, 90, 99, 99.9, max):", np.percentile(rel, [50, 90, 99, 99.9]), rel.max())
