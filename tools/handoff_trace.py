#!/usr/bin/env python3
"""Where does the band-to-band hand-off time go?  Needs the development build of the library
(tools/micro/variant_file.sh k_pcg trace -DSW_TRACE_HANDOFF=1; EULER_HIP_LIB=tools/micro/lib_ablate/libeuler_hip_trace.so).  For column
block 40 of every band pair it prints: producer compute wave finished the block that completes those
columns -> its announce wave issued the granule store -> the consumer's fetch wave deposited the block in
LDS -> the consumer's compute wave picked it up."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import euler_amd as ea

X, Y = (int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "1024x1024").split("x"))
sim = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE)
sim.load_half_tank()
cnt = sim.get(ea.F_COUNT)
cnt[2:Y - 2, 2:X - 2] = 4
sim.set(ea.F_COUNT, cnt)
sim.pcg_op(ea.OP_BUILD_SYSTEM, 0.1)
sim.set(ea.F_PCG_R, np.random.default_rng(1).standard_normal((Y, X)) * (sim.get(ea.F_COUNT) > 0))
for op in (ea.OP_PRECON_FACTOR, ea.OP_FORWARD_SOLVE, ea.OP_BACKWARD_SOLVE):
    sim.pcg_op(op)
for op, name in ((ea.OP_FORWARD_SOLVE, "forward"), (ea.OP_BACKWARD_SOLVE, "backward")):
    for _ in range(3):
        sim.pcg_op(op)
    tl = sim.sweep_timeline(raw=True)
    print(name, "(us): compute done -> announced | announced -> deposited in the next band | deposited -> picked up | total")
    a, b, c = [], [], []
    for p, q in zip(tl, tl[1:]):
        if p[5] and p[6] and q[7] and q[8]:      # raw 100 MHz ticks -> us
            a.append((p[6] - p[5]) / 100.0); b.append((q[7] - p[6]) / 100.0); c.append((q[8] - q[7]) / 100.0)
    for k in range(len(a)):
        print("  band %2d -> %2d: %6.2f | %6.2f | %6.2f | %6.2f" % (k, k + 1, a[k], b[k], c[k], a[k] + b[k] + c[k]))
    if a:
        print("  median        : %6.2f | %6.2f | %6.2f" % (np.median(a), np.median(b), np.median(c)))
sim.close()
