#!/usr/bin/env python3
"""Per-frame cost profile of a workload on the GPU: substeps, PCG iterations, wall ms per frame.
Development aid used to choose bench.py's preroll criterion."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import euler_amd as ea
from euler_amd import scenarios

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--workload", default="dam_break")
ap.add_argument("--budget", type=float, default=240.0)
a = ap.parse_args()
sim = ea.Simulation(a.size, a.size, dot_mode=ea.DOT_TREE)
if a.workload == "dam_break":
    sim.load_text(scenarios.dam_break(), upscale=True)
elif a.workload == "waterfall":
    sim.load_text(scenarios.waterfall(), upscale=True)
else:
    sim.load_half_tank()
t00 = time.perf_counter()
print("frame substeps pcg_iters ms residual markers dt_events")
for f in range(a.frames):
    t0 = time.perf_counter()
    sim.step()
    st = sim.stats()
    ms = 1e3 * (time.perf_counter() - t0)
    print(f, st.last_substeps, st.last_pcg_iterations, "%.2f" % ms, "%.3g" % st.last_residual, st.n_markers, st.marker_dt_events, flush=True)
    if time.perf_counter() - t00 > a.budget:
        break
