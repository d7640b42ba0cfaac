#!/usr/bin/env python3
"""marker stages at 8192^2 (half tank at rest, first frames): HIP-event time per launch of every non-PCG class"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
s = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, tol=0.0).load_half_tank()
s.step()
names = [n for n in ea.profile_class_names() if n not in ("apply_a", "precond_tile")]
s.profile_reset(); s.profile_enable(names)
for _ in range(2):
    s.step()
for k, v in sorted(s.profile().items(), key=lambda kv: -kv[1][0]):
    print("%-18s %8.3f ms total %6d launches %9.1f us each" % (k, v[0], v[1], 1e3 * v[0] / v[1]))
