"""bench_blocks.cpu - the CPU beside the GPU: the oracle (test infrastructure, only ever the thing compared WITH) timed on the host's cores - the `cpu_baseline` leg of the line.

Split out of bench.py in round 5 (the contract line and the driver stay there); nothing here is imported by the product."""
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ CPU baseline
def build_native_oracle():
    """cpu_baseline leg only: compile the oracle for THIS host (reference flags -O3 -ffast-math
    -march=native, CMakeLists.txt:11,18, and strict IEEE) into a temp dir."""
    src = os.path.join(ROOT, "oracle", "euler_oracle.c")
    out = {}
    d = tempfile.mkdtemp(prefix="euler_oracle_")
    for name, flags in (("strict", ["-O3", "-ffp-contract=off"]), ("reference_flags", ["-O3", "-ffast-math", "-march=native"])):
        so = os.path.join(d, "liboracle_%s.so" % name)
        subprocess.check_call(["gcc", "-std=gnu99", "-fPIC", "-shared"] + flags + ["-o", so, src, "-lm"])
        out[name] = so
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_roofline_run(libs, tol, N=2048):
    """The oracle ('port': the from-scratch restatement proven bit-identical to the compiled reference at 100x40), single
    thread like the reference, on a BOUNDED sample of the headline workload: the half-filled tank at 2048^2 (1/16 of the
    8192^2 grid, same fluid fraction, same tol = 0 / 100 iterations per substep), one frame; plus configs[0]."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    res = {}
    for name, so in libs.items():
        o = oracle_lib.Oracle(N, N, lib_path=so).load_half_tank()
        o.c.tol = tol
        t0 = time.perf_counter()
        o.step()
        dt = time.perf_counter() - t0
        res[name] = dict(value=N * N / dt, seconds=round(dt, 3), steps=1, substeps=int(o.c.total_substeps),
                         pcg_iterations=int(o.c.total_pcg_iterations))
        o.close()
    try:      # BASELINE configs[0]: the reference's own grid and scenario (block layout, 100 x 40, 100 frames)
        from euler_amd import scenarios as _sc
        o = oracle_lib.Oracle(100, 40, lib_path=libs["reference_flags"]).load_text(_sc.dam_break())
        t0 = time.perf_counter()
        for _ in range(100):
            o.step()
        dt = time.perf_counter() - t0
        res["_native"] = dict(value=4000 * 100 / dt, seconds=round(dt, 3), steps=100, substeps=int(o.c.total_substeps),
                              pcg_iterations=int(o.c.total_pcg_iterations))
        o.close()
    except Exception as e:      # never let the extra figure break the bench line
        res["_native"] = {"error": str(e)}
    try:      # ... and the COMPILED REFERENCE itself (oracle/_ref: the unmodified main.c, built -O3 -ffp-contract=off in the build container) on the only grid it has
        if oracle_lib.have_ref():
            from golden_util import load as gload, scenario_text
            path = os.path.join(tempfile.mkdtemp(prefix="euler_ref_"), "block.txt")
            with open(path, "w") as f:
                f.write(scenario_text(gload("block_frames.npz")))
            r = oracle_lib.Reference().init(path)
            t0 = time.perf_counter()
            for _ in range(100):
                r.step()
            dt = time.perf_counter() - t0
            res["_reference"] = dict(value=4000 * 100 / dt, seconds=round(dt, 3), steps=100, kind="reference",
                                     what="oracle/_ref/libeuler_ref.so: the unmodified reference main.c (gcc -O3 -ffp-contract=off), scenarios/block.txt on its compile-time 100x40 grid, 100 sim_step calls, single thread")
    except Exception as e:
        res["_reference"] = {"error": str(e)}
    return res


def cpu_from_gpu_state(sim, ea, libs, budget_s=10.0, max_steps=2):
    """configs[1] block: the oracle from the SAME state the GPU timing starts from (single thread); also returns the strict
    build's state after its frames for the in-run parity note."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    res = {}
    snap = {n: sim.get(f) for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"),
                                        (ea.F_SINK, "sink"), (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"),
                                        (ea.F_PRECON, "precon"), (ea.F_MARKERS, "markers"))}
    st = sim.stats()
    for name, so in libs.items():
        o = oracle_lib.Oracle(sim.X, sim.Y, lib_path=so)
        for n in ("u", "v", "solid", "source", "sink", "count", "prev_count", "precon"):
            getattr(o, n)[...] = snap[n]
        o.set_markers(snap["markers"])
        o.c.rng_state = st.rng_state
        o.c.source_exhausted = st.source_exhausted
        t0 = time.perf_counter()
        nsteps = 0
        while True:
            o.step()
            nsteps += 1
            if time.perf_counter() - t0 > budget_s or nsteps >= max_steps:
                break
        dt = time.perf_counter() - t0
        res[name] = dict(value=sim.X * sim.Y * nsteps / dt, seconds=round(dt, 3), steps=nsteps,
                         substeps=int(o.c.total_substeps), pcg_iterations=int(o.c.total_pcg_iterations))
        if name == "strict":
            res["_oracle_after"] = (o.u.copy(), o.v.copy(), (o.count > 0).copy(), nsteps)
        o.close()
    return res


def oracle_from_sim(sim, ea, so, tile_records=0):
    """an oracle (test infrastructure, checker only) holding exactly the state of a GPU handle"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle(sim.X, sim.Y, lib_path=so)
    o.c.tile_records = tile_records
    for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"),
                 (ea.F_SINK, "sink"), (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon")):
        getattr(o, n)[...] = sim.get(f)
    o.set_markers(sim.get(ea.F_MARKERS))
    st = sim.stats()
    o.c.rng_state = st.rng_state
    o.c.source_exhausted = st.source_exhausted
    return o


def cpu_converged_baseline(libs, n=1024):
    """cpu_baseline at EQUAL TOLERANCE: the oracle with the reference's IC(0), single thread, tol 1e-6, cap lifted, one frame of the n x n half tank from rest"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle(n, n, lib_path=libs["reference_flags"]).load_half_tank()
    o.c.max_iterations = 20000
    t0 = time.perf_counter()
    o.step()
    dt = time.perf_counter() - t0
    out = {"value": round(n * n / dt, 1), "unit": "cells*steps/s", "cores": 1, "kind": "port", "seconds": round(dt, 2),
           "substeps": int(o.c.total_substeps), "pcg_iterations": int(o.c.total_pcg_iterations), "last_residual": float(o.c.last_residual),
           "sample": "1 frame of the %dx%d half tank from rest, the reference's IC(0) run to tol 1e-6 (cap lifted), -O3 -ffast-math -march=native, single thread; "
                     "its iteration count grows with N (445 / 880 / 1726 at 512 / 1024 / 2048), so the rate at 8192 is ~8x lower" % (n, n)}
    o.close()
    return out
