#!/usr/bin/env python3
"""CPU study (oracle only, test infrastructure): PCG iterations to tolerance of the reference's IC(0) against the
tile-local IC(0) extension for several tile widths, on the same right-hand side.
usage: precond_study.py [size] [workload: half_tank|dam_break] [frames of preroll]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from euler_amd import scenarios  # noqa: E402


def solve(o, dt, units, max_it, tol):
    c = o.c
    c.tile_records = units
    c.max_iterations = max_it
    c.tol = tol
    o.precon[...] = 0
    u, v = o.u.copy(), o.v.copy()
    t0 = time.perf_counter()
    it = o.lib.eo_project(o.ptr, oracle_lib.C.c_float(dt), o.f32p(o.utmp), o.f32p(o.vtmp), o.f32p(u), o.f32p(v))
    return it, c.last_residual, o.p.copy(), time.perf_counter() - t0


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    wl = sys.argv[2] if len(sys.argv) > 2 else "half_tank"
    pre = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    o = oracle_lib.Oracle(N, N, fast=True)
    if wl == "half_tank":
        o.load_half_tank()
    else:
        o.load_text(scenarios.dam_break(), upscale=True)
    for _ in range(pre):
        o.step()
    # one substep up to the projection: state with utmp / vtmp ready
    dt = o.timestep(0.1)
    L = o.lib
    L.eo_advect_markers(o.ptr, oracle_lib.C.c_float(dt)); L.eo_refresh_marker_counts(o.ptr); L.eo_update_fluid_sources(o.ptr)
    L.eo_extrapolate(o.ptr, o.f32p(o.u), 1); L.eo_extrapolate(o.ptr, o.f32p(o.v), 2)
    L.eo_zero_bounds(o.ptr, o.f32p(o.u), 1); L.eo_zero_bounds(o.ptr, o.f32p(o.v), 2)
    L.eo_advect_u(o.ptr, o.f32p(o.u), o.f32p(o.v), oracle_lib.C.c_float(dt), o.f32p(o.utmp))
    L.eo_advect_v(o.ptr, o.f32p(o.u), o.f32p(o.v), oracle_lib.C.c_float(dt), o.f32p(o.vtmp))
    L.eo_apply_body_forces(o.ptr, o.f32p(o.vtmp), oracle_lib.C.c_float(dt))
    L.eo_zero_bounds(o.ptr, o.f32p(o.utmp), 1); L.eo_zero_bounds(o.ptr, o.f32p(o.vtmp), 2)
    print("N=%d %s preroll=%d dt=%g fluid=%d" % (N, wl, pre, dt, int((o.count > 0).sum())))
    ref = None
    for units in (0, 8, 16, 32, 96, 576):
        it, res, p, sec = solve(o, dt, units, 5000, 1e-6)
        it100, res100, p100, _ = solve(o, dt, units, 100, 1e-6)
        if ref is None:
            ref, ref100 = p, p100
        print("tile_records=%3d: %4d iterations to 1e-6 (res %.2e, %.1fs); |dp|/max|p| vs exact: %.2e ; after 100 its: res %.3e, |dp100|/max|p| %.2e"
              % (units, it, res, sec, np.abs(p - ref).max() / max(np.abs(ref).max(), 1e-300), res100,
                 np.abs(p100 - ref).max() / max(np.abs(ref).max(), 1e-300)))


if __name__ == "__main__":
    main()
