"""Round 6, VERDICT r5 next 5 (configs[1] "fp32, converging, one persistent launch"): what a converged solve costs at 1024^2 in each mode, measured - the data behind DESIGN
section 0 item 5.  The 1024^2 dam break, tol 1e-6 (the reference's), iteration cap lifted; FRAMES frames (the block lands after ~90: the solves that matter are behind that).

    python tools/r06/configs1_modes.py [frames]     -> one JSON line per mode"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 140
for name, precond in (("ic0_tile_mg", ea.PRECOND_IC0_TILE_MG), ("ic0_tile2", ea.PRECOND_IC0_TILE2), ("ic0_tile (resident, f64)", ea.PRECOND_IC0_TILE)):
    sim = ea.Simulation(1024, 1024, precond=precond, dot_mode=ea.DOT_TREE, max_iterations=20000).load_text(scenarios.dam_break(), upscale=True)
    rows = []
    for f in range(FRAMES):
        t0 = time.perf_counter()
        sim.step()
        st = sim.stats()
        rows.append((time.perf_counter() - t0, st.last_substeps, st.last_pcg_iterations))
    tail = rows[-30:]      # the last 30 frames: water on the floor, every solve a real one
    out = {"mode": name, "frames": FRAMES, "seconds": round(sum(r[0] for r in rows), 2), "substeps": sum(r[1] for r in rows), "pcg_iterations": sum(r[2] for r in rows),
           "last_30_frames": {"ms_per_frame": round(1e3 * sum(r[0] for r in tail) / len(tail), 2), "substeps_per_frame": round(sum(r[1] for r in tail) / len(tail), 2),
                              "iterations_per_substep": round(sum(r[2] for r in tail) / max(1, sum(r[1] for r in tail)), 1),
                              "us_per_iteration_incl_stages": round(1e6 * sum(r[0] for r in tail) / max(1, sum(r[2] for r in tail)), 1)},
           "resident_info": list(sim.resident_info())}
    print(json.dumps(out), flush=True)
    sim.close()
