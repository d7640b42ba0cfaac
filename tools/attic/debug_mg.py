import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import euler_amd as ea
from oracle_lib import Oracle

for X, Y in ((260, 300), (1100, 200), (130, 1030), (512, 512), (2048, 2048)):
    for cap in (1, 2, 5, 4000):
        o = Oracle(X, Y, fast=X > 1024); sim = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, tile_records=16, max_iterations=cap)
        o.load_half_tank(); sim.load_half_tank()
        o.c.tile_records = 16; o.c.coarse_m = o.lib.eo_coarse_m(X, Y); o.c.coarse_mg = 1; o.c.max_iterations = cap
        its = []
        for k in range(2):
            dt = sim.timestep(0.1); dto = o.timestep(0.1)
            assert dt == dto, (dt, dto)
            sim.substep(dt); o.substep(dt)
            st = sim.stats()
            p, pr = sim.get(ea.F_PRESSURE), o.p
            its.append((st.last_pcg_iterations, o.c.last_pcg_iterations, "%.3g" % (np.abs(p - pr).max() / max(np.abs(pr).max(), 1e-300)), "%.3g/%.3g" % (st.last_residual, o.c.last_residual)))
        print(X, Y, "coarse_m", o.c.coarse_m, "cap", cap, its, flush=True)
        sim.close(); o.close()
