#!/usr/bin/env python3
"""One frame of an N^2 half tank in the multilevel mode, a FIXED number of iterations per solve (tol 0: timing builds with pieces switched off never converge); run under
rocprofv3 --kernel-trace --stats to see the cycle's launches.  usage: mg_probe.py [N] [frames] [iterations per solve]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
its = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, max_iterations=its, tol=0.0).load_half_tank()
for f in range(frames):
    t0 = time.perf_counter()
    sim.step()
    st = sim.stats()
    print("frame %d: %d substeps, %d iterations, residual %.3g, %.1f ms" % (f, st.last_substeps, st.last_pcg_iterations, st.last_residual, 1e3 * (time.perf_counter() - t0)), flush=True)
