#!/usr/bin/env python3
"""First-contact diagnostics on a GPU box: run the HIP path stage by stage against the oracle and
print where (stage, field, index) the first difference appears.  Development aid, not a test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C

import numpy as np

import euler_amd as ea
from golden_util import bits_equal, load, scenario_text
from oracle_lib import Oracle

FIELDS = ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_COUNT, "count"),
          (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon"), (ea.F_MARKERS, "markers"), (ea.F_PRESSURE, "p"))


def diff(sim, o, skip=()):
    out = []
    for f, n in FIELDS:
        if n in skip:
            continue
        got = sim.get(f)
        want = o.markers if n == "markers" else getattr(o, n)
        if not bits_equal(got, want):
            if got.shape != want.shape:
                out.append("%s shape %s vs %s" % (n, got.shape, want.shape))
                continue
            bad = np.argwhere(got != want)
            i = tuple(bad[0]) if len(bad) else None
            out.append("%s: %d differ, first %s got %r want %r" % (n, len(bad), i, got[i] if i else None, want[i] if i else None))
    return out


def oracle_stage(o, st, dt):
    L, p, f = o.lib, o.ptr, C.c_float(dt)
    if st == ea.STAGE_ADVECT_MARKERS: L.eo_advect_markers(p, f)
    elif st == ea.STAGE_REFRESH_COUNTS: L.eo_refresh_marker_counts(p)
    elif st == ea.STAGE_SOURCES: L.eo_update_fluid_sources(p)
    elif st == ea.STAGE_EXTRAPOLATE:
        L.eo_extrapolate(p, o.f32p(o.u), 1); L.eo_extrapolate(p, o.f32p(o.v), 2)
        L.eo_zero_bounds(p, o.f32p(o.u), 1); L.eo_zero_bounds(p, o.f32p(o.v), 2)
    elif st == ea.STAGE_ADVECT_VELOCITY:
        L.eo_advect_u(p, o.f32p(o.u), o.f32p(o.v), f, o.f32p(o.utmp)); L.eo_advect_v(p, o.f32p(o.u), o.f32p(o.v), f, o.f32p(o.vtmp))
        L.eo_apply_body_forces(p, o.f32p(o.vtmp), f); L.eo_zero_bounds(p, o.f32p(o.utmp), 1); L.eo_zero_bounds(p, o.f32p(o.vtmp), 2)
    elif st == ea.STAGE_PROJECT:
        L.eo_project(p, f, o.f32p(o.utmp), o.f32p(o.vtmp), o.f32p(o.u), o.f32p(o.v))


def run(Xs, Ys, scn, frames, **kw):
    text = scenario_text(load(scn + "_frames.npz"))
    up = (Xs, Ys) != (100, 40)
    o = Oracle(Xs, Ys).load_text(text, upscale=up)
    sim = ea.Simulation(Xs, Ys, **kw).load_text(text, upscale=up)
    d = diff(sim, o, skip=("p",))
    print("[%s %dx%d %s] init:" % (scn, Xs, Ys, kw), d or "ok")
    names = ["advect_markers", "refresh", "sources", "extrapolate", "advect_velocity", "project"]
    nbad = 0
    for fr in range(frames):
        ft = np.float32(0.1)
        for sub in range(8):
            if not ft > 0:
                break
            dt = sim.timestep(float(ft))
            dto = o.timestep(float(ft))
            if dt != dto:
                print("  frame %d sub %d dt %r vs %r" % (fr, sub, dt, dto)); nbad += 1
            ft = np.float32(ft - np.float32(dto))
            for st in range(6):
                sim.stage(st, dto)
                oracle_stage(o, st, dto)
                skip = ("p",) if st != ea.STAGE_PROJECT else ()
                if st == ea.STAGE_ADVECT_VELOCITY: skip = skip + ()
                d = diff(sim, o, skip)
                if d:
                    nbad += 1
                    print("  frame %d sub %d stage %s:" % (fr, sub, names[st]))
                    for x in d: print("     ", x)
                    # teacher-force: overwrite GPU state with the oracle's to keep going
                    for f, n in FIELDS:
                        if n == "markers": sim.set_markers(o.markers)
                        elif n != "p": sim.set(f, getattr(o, n))
                    if nbad > 6:
                        return nbad
        st_ = sim.stats()
    print("  done: mismatching stages:", nbad, "pcg iters (last)", sim.stats().last_pcg_iterations, "oracle", o.c.last_pcg_iterations)
    return nbad


if __name__ == "__main__":
    t0 = time.time()
    sim = ea.Simulation(100, 40)
    print(sim.device_name())
    bad = 0
    bad += run(100, 40, "block", 3, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=ea.SWEEP_SIMPLE)
    bad += run(100, 40, "block", 3, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=ea.SWEEP_BAND)
    bad += run(100, 40, "waterfall", 3, dot_mode=ea.DOT_SEQUENTIAL)
    bad += run(100, 40, "filter", 30, dot_mode=ea.DOT_SEQUENTIAL)
    bad += run(200, 150, "block", 2, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=ea.SWEEP_BAND)
    bad += run(200, 150, "block", 2, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=ea.SWEEP_SIMPLE)
    print("total mismatching stages", bad, "elapsed %.1fs" % (time.time() - t0))
