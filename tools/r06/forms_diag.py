"""Development aid (round 6): two handles side by side - the default forms against rounds 1-5's (EULER_OPT_MARKERS_TWO_PASS, _BUILD_TWO_PASS, _VELOCITY_TWO_PASS) - through the
legs of tests/test_gpu_parity.py::test_full_size_16384_dam_break_properties_both_modes at a size given on the command line; prints where they differ."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import euler_amd as ea
from euler_amd import scenarios

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE).load_text(scenarios.dam_break(), upscale=True)
b = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE).load_text(scenarios.dam_break(), upscale=True)
for k in (ea.OPT_MARKERS_TWO_PASS, ea.OPT_BUILD_TWO_PASS, ea.OPT_VELOCITY_TWO_PASS, ea.OPT_NO_TILE_MAP):
    b.set_option(k, 1)


def cmp(tag):
    out = []
    for name, f in (("u", ea.F_U), ("v", ea.F_V), ("p", ea.F_PRESSURE), ("count", ea.F_COUNT), ("markers", ea.F_MARKERS)):
        x, y = a.get(f), b.get(f)
        same = x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))
        out.append("%s:%s" % (name, "=" if same else "DIFF(%d)" % (int((x != y).sum()) if x.shape == y.shape else -1)))
    sa, sb = a.stats(), b.stats()
    p = a.get(ea.F_PRESSURE)
    print(tag, " ".join(out), "iters", sa.last_pcg_iterations, sb.last_pcg_iterations, "substeps", sa.last_substeps, sb.last_substeps, "p>0", int((p > 0).sum()), flush=True)


for f in range(60):
    a.step(); b.step()
    if f % 5 == 0 or a.stats().last_pcg_iterations >= 100:
        cmp("frame %d" % f)
    if a.stats().last_pcg_iterations >= 100:
        break
for precond in (ea.PRECOND_IC0_TILE, ea.PRECOND_IC0, ea.PRECOND_IC0_TILE_MG):
    for s in (a, b):
        s.set_precond(precond)
        if precond == ea.PRECOND_IC0_TILE_MG:
            s.set_solver(max_iterations=4000)
    a.step(); b.step()
    cmp("leg %d" % precond)
