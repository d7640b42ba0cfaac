#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the COMPILED, UNMODIFIED reference.

Runs only in the authoring container (needs /root/reference and oracle/_ref/libeuler_ref.so,
built by `make -C oracle ref` with -O3 -ffp-contract=off, no -ffast-math).  The fixtures are
data: inputs (scenario text is NOT stored - only the parsed cell grids) and expected outputs.

  <scn>_frames.npz   full state after selected frames + FNV-1a-64 hashes after every frame
  <scn>_substep.npz  one teacher-forced substep: state before, dt, and every array a stage of
                     sim_step (main.c:855-893) changed, stage by stage
  <scn>_render.npz   draw_rows() bytes (main.c:914-951) for selected frames / window sizes

The RNG state (function-static in the reference, main.c:204) is not readable from outside;
it is reconstructed by replaying the reference's draw count with the same generator.
"""
import ctypes as C
import json
import os
import platform
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Reference, fnv1a64, REF_X, REF_Y  # noqa: E402

REF_SCN = "/root/reference/scenarios"
PLAN = {
    # scenario: (frames to run, frames stored in full, (frame, substep) for the teacher-forced cut)
    "basic": (120, [0, 1, 10, 60, 119], (8, 0)),
    "block": (100, [0, 1, 10, 40, 99], (45, 1)),
    "filter": (100, [0, 1, 10, 28, 29, 99], (28, 1)),   # 28/1: a dt-shortening collision (main.c:501)
    "waterfall": (460, [0, 1, 10, 100, 449, 450, 451, 459], (30, 0)),  # exhaustion latches at 450
    "weird-edges": (100, [0, 1, 10, 50, 99], (20, 0)),
}
MASK64 = (1 << 64) - 1


def rng_next(state):
    state ^= state >> 12
    state ^= (state << 25) & MASK64
    state ^= state >> 27
    return state


def rng_advance(state, n):
    for _ in range(n):
        state = rng_next(state)
    return state


def snapshot(ref):
    return dict(u=ref.u.copy(), v=ref.v.copy(), count=ref.count.copy(), prev_count=ref.prev_count.copy(),
                precon=ref.precon.copy(), markers=ref.markers.copy(),
                exhausted=np.uint8(ref.source_exhausted))


def hashes(ref):
    return [fnv1a64(ref.u), fnv1a64(ref.v), fnv1a64(ref.count), fnv1a64(ref.markers)]


def run_stages(ref, dt):
    """The body of the reference's substep loop (main.c:855-893), one external symbol at a time."""
    L = ref.lib
    f = C.c_float(dt)
    yield "advect_markers", lambda: L.advect_markers(f)
    yield "refresh_marker_counts", lambda: L.refresh_marker_counts()
    yield "update_fluid_sources", lambda: L.update_fluid_sources()
    yield "extrapolate_u", lambda: L.extrapolate(ref.fp(ref.u), 1)
    yield "extrapolate_v", lambda: L.extrapolate(ref.fp(ref.v), 2)
    yield "zero_bounds_u", lambda: L.zero_bounds(ref.fp(ref.u), 1)
    yield "zero_bounds_v", lambda: L.zero_bounds(ref.fp(ref.v), 2)
    yield "advect_u", lambda: L.advect_u(ref.fp(ref.u), ref.fp(ref.v), f, ref.fp(ref.utmp))
    yield "advect_v", lambda: L.advect_v(ref.fp(ref.u), ref.fp(ref.v), f, ref.fp(ref.vtmp))
    yield "apply_body_forces", lambda: L.apply_body_forces(ref.fp(ref.vtmp), f)
    yield "zero_bounds_utmp", lambda: L.zero_bounds(ref.fp(ref.utmp), 1)
    yield "zero_bounds_vtmp", lambda: L.zero_bounds(ref.fp(ref.vtmp), 2)
    yield "project", lambda: L.project(f, ref.fp(ref.utmp), ref.fp(ref.vtmp), ref.fp(ref.u), ref.fp(ref.v))


STATE = ("u", "v", "utmp", "vtmp", "count", "prev_count", "precon")


def full_state(ref):
    d = {n: getattr(ref, n).copy() for n in STATE}
    d["markers"] = ref.markers.copy()
    return d


def manual_frame(ref, draws, record):
    """One frame of the reference driven stage by stage = the body of sim_step (main.c:849-894)
    without the pause gate / colour / g_frame_count.  Returns (per-substep records, rng draws)."""
    L = ref.lib
    recs = []
    ft = np.float32(0.1)
    step = 0
    while ft > 0 and step < 8:
        dt = np.float32(L.calculate_timestep(C.c_float(ft)))
        ft = np.float32(ft - dt)
        rec = {"dt": dt}
        if record:
            rec.update(before=full_state(ref), stages=[], rng_before=np.uint64(rng_advance(SEED, draws)),
                       exhausted_before=np.uint8(ref.source_exhausted))
            prev = rec["before"]
        n_mid = None
        for name, fn in run_stages(ref, float(dt)):
            fn()
            if name == "refresh_marker_counts":
                n_mid = ref.n_markers
            if name == "update_fluid_sources":
                draws += 2 * (ref.n_markers - n_mid)
            if record:
                cur = full_state(ref)
                changed = {k: v for k, v in cur.items()
                           if v.shape != prev[k].shape or not np.array_equal(v.view(np.uint8), prev[k].view(np.uint8))}
                rec["stages"].append((name, changed))
                prev = cur
        if record:
            rec["exhausted_after"] = np.uint8(ref.source_exhausted)
            rec["rng_after"] = np.uint64(rng_advance(SEED, draws))
        recs.append(rec)
        step += 1
    return recs, draws


SEED = 0x9bd185c449534b91  # main.c:204


def main():
    manifest = {
        "generator": "tests/golden/make_golden.py",
        "reference": "cgmb/euler main.c + misc/*.c, unmodified, via oracle/Makefile `ref`",
        "flags": "-std=gnu99 -O3 -ffp-contract=off -fno-fast-math -fPIC -Dmain=euler_main",
        "gcc": subprocess.check_output(["gcc", "--version"]).decode().splitlines()[0],
        "machine": platform.machine(),
        "grid": [REF_X, REF_Y],
        "hash": "FNV-1a-64 over raw little-endian bytes of g_u, g_v, g_marker_count, g_markers[0:n]",
        "scenarios": {},
    }
    for scn, (nframes, keep, (tf_frame, tf_sub)) in PLAN.items():
        path = os.path.join(REF_SCN, scn + ".txt")
        ref = Reference().init(path)
        chk = Reference().init(path)       # stepped with sim_step(); must agree with the manual drive
        out = {"solid": ref.solid.copy(), "source": ref.source.copy(), "sink": ref.sink.copy()}
        for k, v in snapshot(ref).items():
            out["init_" + k] = v
        # init draws: 2 per marker (main.c:260-261); interior fluid cells are never sink/solid so
        # none is deleted by the refresh at the end of sim_init
        draws = 2 * ref.n_markers
        out["init_rng"] = np.uint64(rng_advance(SEED, draws))
        hs, nm, nsub, dts = [], [], [], []
        tf = None
        renders = {}
        for f in range(nframes):
            recs, draws = manual_frame(ref, draws, record=(f == tf_frame))
            chk.step()
            assert hashes(ref) == hashes(chk), (scn, f)
            if f == tf_frame:
                tf = recs[tf_sub]
            hs.append(hashes(ref))
            nm.append(ref.n_markers)
            nsub.append(len(recs))
            dts.append([float(r["dt"]) for r in recs] + [0.0] * (8 - len(recs)))
            if f in keep:
                for k, v in snapshot(ref).items():
                    out["f%d_%s" % (f, k)] = v
                out["f%d_rng" % f] = np.uint64(rng_advance(SEED, draws))
                for (wx, wy) in ((98, 38), (80, 24), (40, 10), (200, 100)):
                    renders["f%d_w%dx%d" % (f, wx, wy)] = np.frombuffer(ref.render(wx, wy), dtype=np.uint8)
        out["hashes"] = np.array(hs, dtype=np.uint64)
        out["n_markers"] = np.array(nm, dtype=np.int64)
        out["n_substeps"] = np.array(nsub, dtype=np.int64)
        out["dts"] = np.array(dts, dtype=np.float32)
        out["frames_full"] = np.array(keep, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, scn + "_frames.npz"), **out)
        np.savez_compressed(os.path.join(HERE, scn + "_render.npz"), **renders)
        sub = {"dt": tf["dt"], "frame": np.int64(tf_frame), "substep": np.int64(tf_sub),
               "exhausted_before": tf["exhausted_before"], "exhausted_after": tf["exhausted_after"],
               "rng_before": tf["rng_before"], "rng_after": tf["rng_after"],
               "solid": ref.solid.copy(), "source": ref.source.copy(), "sink": ref.sink.copy(),
               "stage_names": np.array([n for n, _ in tf["stages"]])}
        for k, v in tf["before"].items():
            sub["before_" + k] = v
        for i, (name, changed) in enumerate(tf["stages"]):
            for k, v in changed.items():
                sub["s%02d_%s" % (i, k)] = v
        np.savez_compressed(os.path.join(HERE, scn + "_substep.npz"), **sub)
        manifest["scenarios"][scn] = {
            "frames": nframes, "full_frames": keep, "teacher_forced": [tf_frame, tf_sub],
            "substeps": int(sum(nsub)),
            "final_hashes_u_v_count_markers": ["%016x" % h for h in hs[-1]], "final_markers": nm[-1],
        }
        print(scn, "frames", nframes, "substeps", sum(nsub), "markers", nm[-1], "hash_u %016x" % hs[-1][0])
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
