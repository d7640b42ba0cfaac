#!/usr/bin/env python3
"""round 4 soak: thousands of frames through the resident solver (one persistent launch per solve): no time-out, no fall-back, finite fields, the marker count of a closed scene constant"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import euler_amd as ea
from euler_amd import scenarios

rows = []
for name, N, text, frames, kw in (("1024^2 dam break, resident f64", 1024, scenarios.dam_break(), 3000, {}),
                                  ("1024^2 dam break, resident f32", 1024, scenarios.dam_break(), 3000, dict(pcg_precision=ea.PCG_F32)),
                                  ("768^2 waterfall (sources active), resident f64", 768, scenarios.waterfall(), 2000, {}),
                                  ("2048^2 dam break, resident f64 (1300+ active chunks)", 2048, scenarios.dam_break(), 600, {})):
    s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, **kw).load_text(text, upscale=True)
    n0 = s.stats().n_markers
    torch.cuda.synchronize(); t0 = time.perf_counter()
    solves = 0
    for f in range(frames):
        s.step()
        st = s.stats()
        solves += st.last_substeps if st.last_pcg_iterations else 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = s.stats()
    u, v, p = s.get(ea.F_U), s.get(ea.F_V), s.get(ea.F_PRESSURE)
    info = s.resident_info()
    rows.append(dict(run=name, frames=frames, seconds=round(dt, 1), substeps=int(st.total_substeps), pcg_iterations=int(st.total_pcg_iterations), markers=[int(n0), int(st.n_markers)],
                     finite=bool(np.isfinite(u).all() and np.isfinite(v).all() and np.isfinite(p).all()), max_abs_u=float(np.abs(u).max()), resident_solves=info[1], fallbacks=info[2], solves=solves))
    print(json.dumps(rows[-1]), flush=True)
    s.close()
print("| run | frames | seconds | substeps | PCG iterations | markers start -> end | finite | solves run resident / with a right-hand side | fall-backs |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %d | %.1f | %d | %d | %d -> %d | %s | %d / %d | %d |" % (r["run"], r["frames"], r["seconds"], r["substeps"], r["pcg_iterations"], r["markers"][0], r["markers"][1], r["finite"], r["resident_solves"], r["solves"], r["fallbacks"]))
