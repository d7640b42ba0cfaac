#!/usr/bin/env python3
"""BASELINE configs[1] as named: the 1024 x 1024 dam break, 500 steps (frames), on one MI355X - wall time per mode.
parity = the reference's IC(0), bit-identical iterates (cap 100); tile = tile-local IC(0), resident solver (f64 / f32), cap 100; converged = multilevel mode, every solve to 1e-6."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import euler_amd as ea
from euler_amd import scenarios

N, STEPS = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 500
modes = (("parity: reference IC(0), f64, cap 100", dict(precond=ea.PRECOND_IC0)),
         ("tile-local, multi-kernel, f64, cap 100", dict(precond=ea.PRECOND_IC0_TILE, resident=ea.RESIDENT_OFF)),
         ("tile-local, resident, f64, cap 100", dict(precond=ea.PRECOND_IC0_TILE)),
         ("tile-local, resident, f32 (configs[1] 'fp32'), cap 100", dict(precond=ea.PRECOND_IC0_TILE, pcg_precision=ea.PCG_F32)),
         ("multilevel, f64, every solve to 1e-6", dict(precond=ea.PRECOND_IC0_TILE_MG, max_iterations=20000, pcg_poll_interval=32)))
rows = []
for name, kw in modes:
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, **kw).load_text(scenarios.dam_break(), upscale=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        s.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = s.stats()
    fl = int((s.get(ea.F_COUNT) > 0).sum())
    rows.append(dict(mode=name, seconds=round(dt, 2), cells_steps_per_s=round(N * N * STEPS / dt), substeps=int(st.total_substeps), pcg_iterations=int(st.total_pcg_iterations),
                     markers=int(st.n_markers), fluid_cells=fl, max_abs_u=float(np.abs(s.get(ea.F_U)).max()), resident=list(s.resident_info())))
    print(json.dumps(rows[-1]), flush=True)
    s.close()
print("| mode | seconds for %d steps | cells*steps/s | substeps | PCG iterations | markers at the end | fluid cells |" % STEPS)
print("|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %.2f | %.3g | %d | %d | %d | %d |" % (r["mode"], r["seconds"], r["cells_steps_per_s"], r["substeps"], r["pcg_iterations"], r["markers"], r["fluid_cells"]))
