"""Development aid: CFL substeps per frame of the half tank from rest (tol 0, 100 iterations per solve), both modes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for pc in (ea.PRECOND_IC0_TILE, ea.PRECOND_IC0):
    sim = ea.Simulation(N, N, precond=pc, tol=0.0)
    sim.load_half_tank()
    out = []
    for f in range(frames):
        sim.step()
        st = sim.stats()
        out.append(st.last_substeps)
    print("precond", pc, "substeps per frame:", out, flush=True)
    sim.close()
