#!/usr/bin/env python3
"""BASELINE configs[4] as named: the 4096 x 4096 waterfall (sources and sinks active), 2000 steps, one MI355X - wall time per mode."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import euler_amd as ea
from euler_amd import scenarios

N, STEPS = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
which = sys.argv[2:] or ["resident", "multilevel"]
modes = {"resident": ("tile-local, resident where the active chunks fit (f64), cap 100", dict(precond=ea.PRECOND_IC0_TILE)),
         "multikernel": ("tile-local, multi-kernel, f64, cap 100", dict(precond=ea.PRECOND_IC0_TILE, resident=ea.RESIDENT_OFF)),
         "multilevel": ("multilevel, f64, every solve to 1e-6", dict(precond=ea.PRECOND_IC0_TILE_MG, max_iterations=20000, pcg_poll_interval=32)),
         "parity": ("parity: reference IC(0), f64, cap 100", dict(precond=ea.PRECOND_IC0))}
rows = []
for key in which:
    name, kw = modes[key]
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, **kw).load_text(scenarios.waterfall(), upscale=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solves = 0
    for f in range(STEPS):
        s.step()
        solves += s.stats().last_substeps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = s.stats()
    fl = int((s.get(ea.F_COUNT) > 0).sum())
    rows.append(dict(mode=name, seconds=round(dt, 2), cells_steps_per_s=round(N * N * STEPS / dt), substeps=int(st.total_substeps), pcg_iterations=int(st.total_pcg_iterations),
                     markers=int(st.n_markers), fluid_cells=fl, source_exhausted=int(st.source_exhausted), resident=list(s.resident_info()), solves=solves))
    print(json.dumps(rows[-1]), flush=True)
    s.close()
print("| mode | seconds for %d steps | cells*steps/s | substeps | PCG iterations | markers at the end | fluid cells | solves run resident |" % STEPS)
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %.2f | %.3g | %d | %d | %d | %d | %d of %d |" % (r["mode"], r["seconds"], r["cells_steps_per_s"], r["substeps"], r["pcg_iterations"], r["markers"], r["fluid_cells"], r["resident"][1], r["solves"]))
