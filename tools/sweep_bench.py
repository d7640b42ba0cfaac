#!/usr/bin/env python3
"""Micro-benchmark of the IC(0) sweeps: per-launch time vs grid shape (1 band vs many, narrow vs
wide) to separate per-step cost from band-pipeline fill.  Development aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import euler_amd as ea

shapes = [(1024, 64), (4096, 64), (1024, 128), (1024, 256), (1024, 1024), (4096, 1024), (4096, 4096), (8192, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
print("X Y bands  factor_us forward_us backward_us  | steps=X+63  fwd ns/step(1 band eq)  fwd algo GB/s")
for X, Y in shapes:
    for mode, name in ((ea.SWEEP_BAND, "skew"),):
        if name == "direct" and X * Y > 4096 * 1024:
            continue
        sim = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE, sweep_mode=mode).load_half_tank()
        sim.pcg_op(ea.OP_BUILD_SYSTEM, 0.1)
        rng = np.random.default_rng(1)
        sim.set(ea.F_PCG_R, rng.standard_normal((Y, X)) * (sim.get(ea.F_COUNT) > 0))
        for op in (ea.OP_PRECON_FACTOR, ea.OP_FORWARD_SOLVE, ea.OP_BACKWARD_SOLVE):
            sim.pcg_op(op)
        sim.profile_enable(["precon_factor", "forward_solve", "backward_solve"])
        sim.profile_reset()
        reps = 5
        for _ in range(reps):
            for op in (ea.OP_PRECON_FACTOR, ea.OP_FORWARD_SOLVE, ea.OP_BACKWARD_SOLVE):
                sim.pcg_op(op)
        pr = sim.profile()
        us = {k: 1e3 * v[0] / v[1] for k, v in pr.items()}
        nb = (Y + 63) // 64
        f = us["forward_solve"]
        print("%5d %5d %4d %-6s %9.1f %9.1f %9.1f | %6d  %7.1f  %8.1f" % (
            X, Y, nb, name, us["precon_factor"], f, us["backward_solve"], X + 63, 1e3 * f / (X + 63), 25.0 * X * Y / (f * 1e-6) / 1e9))
        sim.close()
