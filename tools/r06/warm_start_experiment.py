"""Round 6, CPU experiment (the oracle's building blocks driven from numpy; nothing of the product): would starting a solve from the previous substep's pressure (scaled by
dt_prev / dt) instead of zero (main.c:739) save iterations in the multilevel mode?  A tank at rest: 29 -> 0.  A 512^2 dam break over eight frames from the impact on: 2673 -> 2680.
The tolerance is absolute (1e-6) against pressures of 1e4 - 1e6: ten to twelve orders of magnitude either way.  Not built.

    AFTER=10 PMIN=2000 python tools/r06/warm_start_experiment.py 512 dam_break"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
from oracle_lib import Oracle
from euler_amd import scenarios
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
scn = sys.argv[2] if len(sys.argv) > 2 else "dam_break"
o = Oracle(N, N, fast=True)
if scn == "half_tank": o.load_half_tank()
else: o.load_text(getattr(scenarios, scn)(), upscale=True)
o.c.tile_records = 16; o.c.coarse_m = o.lib.eo_coarse_m(N, N); o.c.coarse_mg = 1; o.c.max_iterations = 4000
tol = float(o.c.tol)
def pcg(p0):
    fl = o.count > 0
    p = np.where(fl, p0, 0.0).astype(np.float64)
    ap = np.zeros_like(p); z = np.zeros_like(p)
    o.lib.eo_apply_a(o.ptr, o.f64p(p), o.f64p(ap))
    r = np.where(fl, o.b - ap, 0.0)
    if o.lib.eo_inf_norm(o.ptr, o.f64p(r)) <= tol: return 0
    o.lib.eo_apply_preconditioner(o.ptr, o.f64p(r), o.f64p(z))
    s = z.copy(); sigma = o.lib.eo_dot(o.ptr, o.f64p(z), o.f64p(r))
    for it in range(1, 4001):
        o.lib.eo_apply_a(o.ptr, o.f64p(s), o.f64p(ap))
        alpha = sigma / o.lib.eo_dot(o.ptr, o.f64p(ap), o.f64p(s))
        p += alpha * s; r -= alpha * ap
        if o.lib.eo_inf_norm(o.ptr, o.f64p(r)) <= tol: return it
        o.lib.eo_apply_preconditioner(o.ptr, o.f64p(r), o.f64p(z))
        sn = o.lib.eo_dot(o.ptr, o.f64p(z), o.f64p(r)); s = z + (sn / sigma) * s; sigma = sn
    return 4000
stats = []
p_prev = None; dt_prev = None; frames = 0; after = 0
while after < int(os.environ.get("AFTER", 6)) and frames < 400:
    ft = 0.1; sub = 0
    cold_f = warm_f = ref_f = 0
    while ft > 0 and sub < 8:
        dt = o.timestep(ft); ft -= dt; sub += 1
        o.substep(dt)
        its = int(o.c.last_pcg_iterations)
        pmax = float(np.abs(o.p).max())
        if pmax > float(os.environ.get("PMIN", 10)) and p_prev is not None:
            cold = pcg(np.zeros_like(o.p))
            warm = pcg(p_prev * (dt_prev / dt))
            warm_ns = pcg(p_prev)
            stats.append((frames, sub, its, cold, warm, warm_ns))
            cold_f += cold; warm_f += warm
        p_prev = o.p.copy(); dt_prev = dt
    frames += 1
    if stats and stats[-1][0] == frames - 1:
        after += 1
        print("frame", frames, "substeps", sub, "oracle/cold/warm iterations:", sum(s[2] for s in stats if s[0]==frames-1), cold_f, warm_f, "unscaled warm", sum(s[5] for s in stats if s[0]==frames-1), flush=True)
print("total cold", sum(s[3] for s in stats), "warm", sum(s[4] for s in stats), "unscaled", sum(s[5] for s in stats))
