#!/usr/bin/env python3
"""Band-pipeline timeline of the IC(0) sweeps (euler_sweep_timeline): per band entry / exit time,
so that per-step cost, hand-off lag and stalls can be read off directly.  Development aid.

usage: sweep_timeline.py [XxY[:full] ...]     (":full" = fluid everywhere, default half tank)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import euler_amd as ea

shapes = sys.argv[1:] or ["1024x1024", "8192x128", "8192x1024", "8192x8192"]
for spec in shapes:
    full = spec.endswith(":full")
    X, Y = (int(t) for t in spec.split(":")[0].split("x"))
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE)
    sim.load_half_tank()
    if full:
        cnt = sim.get(ea.F_COUNT)
        cnt[2:Y - 2, 2:X - 2] = 4
        sim.set(ea.F_COUNT, cnt)
    sim.pcg_op(ea.OP_BUILD_SYSTEM, 0.1)
    rng = np.random.default_rng(1)
    sim.set(ea.F_PCG_R, rng.standard_normal((Y, X)) * (sim.get(ea.F_COUNT) > 0))
    for op in (ea.OP_PRECON_FACTOR, ea.OP_FORWARD_SOLVE, ea.OP_BACKWARD_SOLVE):
        sim.pcg_op(op)
    for op, name in ((ea.OP_FORWARD_SOLVE, "forward"), (ea.OP_BACKWARD_SOLVE, "backward")):
        for _ in range(3):
            sim.pcg_op(op)
        tl = [r for r in sim.sweep_timeline(raw=True) if r[3] > 0]
        if not tl:
            continue
        t0 = min(r[0] for r in tl)
        end = max(r[2] for r in tl)
        steps = tl[0][3] * 8
        dur = [r[2] - r[1] for r in tl]
        lag = [tl[i + 1][2] - tl[i][2] for i in range(len(tl) - 1)]
        print("%s %s%s: bands %d  blocks/band %d  total %.1f us | band run: first %.1f us (%.1f ns/step) median %.1f us | "
              "exit-to-exit lag: median %.2f us max %.2f us | stalled blocks: median %d max %d | entry spread %.1f us" % (
                  spec, name, "", len(tl), tl[0][3], end - t0, dur[0], 1e3 * dur[0] / steps, float(np.median(dur)),
                  float(np.median(lag)) if lag else 0.0, max(lag) if lag else 0.0,
                  int(np.median([r[4] for r in tl])), max(r[4] for r in tl), max(r[0] for r in tl) - t0))
        if os.environ.get("TIMELINE_DUMP"):
            for i, r in enumerate(tl):
                print("   band %3d entry %9.2f first %9.2f exit %9.2f blocks %d stalls %d  (raw words 4..7: %s)" % ((i,) + r[:5] + (r[5:],)))
    sim.close()
