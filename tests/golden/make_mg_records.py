#!/usr/bin/env python3
"""Regenerate tests/golden/mg_records.npz: TOLERANCE records of the multilevel mode (EULER_PRECOND_IC0_TILE_MG) at BASELINE-sized grids.

The multilevel preconditioner replaces main.c:577-627; it is not the reference's arithmetic, so nothing about it can be bit-exact - what the GPU tests can pin is
the oracle's RESTATEMENT of it (eo_sim.coarse_mg: mg_build / mg_vcycle in oracle/euler_oracle.c) run to the reference's tolerance: the iteration counts, the residual
reached, max |p| and the pressure on a strided sample grid.  The small-grid GPU tests step the live oracle beside the GPU (tests/test_gpu_tile_precond.py); at 4096^2 and
2048^2 that is minutes of one core, so it is run ONCE here, in the build container:

    python tests/golden/make_mg_records.py [name ...]

Records (tests/test_gpu_tile_precond.py::test_multilevel_mode_at_baseline_sizes_against_recorded_oracle):
  half_tank_4096_mg   the 4096^2 half tank from rest, one substep to 1e-6
  half_tank_8192_mg   the same at 8192^2: BASELINE configs[2]'s grid
  dam_break_2048_mg   the 2048^2 dam break (configs[1] / [3]'s scenario), free-running from frame 0; the block falls freely for most of a hundred frames (the solves that happen
                      on the way work on rounding noise, max p ~ 1e-4), the record holds the first FRAMES_AFTER frames from the impact on (max p > IMPACT_P)
  dam_break_1024_f32  NOT the multilevel mode: configs[1] as named ("fp32") - the oracle's float restatement of the capped tile-local solve (eo_sim.pcg_f32) at 1024^2, the first
                      three frames whose solves run into the cap (tests/test_gpu_resident.py::test_f32_variant_at_configs1_size_against_the_recorded_oracle_restatement)
The file also carries the SHA-1 of oracle/euler_oracle.c it was generated from (tests/test_trajectories.py compares it with the source on every CPU run)."""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle_lib import Oracle  # noqa: E402

OUT = os.path.join(HERE, "mg_records.npz")
ORACLE_SRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "euler_oracle.c")
SAMPLE = 64            # the pressure on a SAMPLE x SAMPLE strided grid
FRAMES_AFTER = 6       # dam break: frames recorded from the impact on
IMPACT_P = 10.0        # ... i.e. from the first frame whose pressure is not rounding noise


def oracle_sha1():
    with open(ORACLE_SRC, "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def mg_oracle(X, Y, max_it=4000):
    o = Oracle(X, Y)
    o.c.tile_records = 16
    o.c.coarse_m = o.lib.eo_coarse_m(X, Y)
    o.c.coarse_mg = 1
    o.c.max_iterations = max_it
    return o


def sample(p):
    Y, X = p.shape
    ys = (np.arange(SAMPLE) * (Y - 1)) // (SAMPLE - 1)
    xs = (np.arange(SAMPLE) * (X - 1)) // (SAMPLE - 1)
    return np.ascontiguousarray(p[np.ix_(ys, xs)])


def rec_half_tank(out, N=4096):
    o = mg_oracle(N, N)
    o.load_half_tank()
    t0 = time.perf_counter()
    dt = o.timestep(0.1)
    o.substep(dt)
    p = o.p
    out["half_tank_%d_mg.scalars" % N] = np.array([o.c.last_pcg_iterations, o.c.last_residual, np.abs(p).max(), dt, o.n_markers], np.float64)
    out["half_tank_%d_mg.p" % N] = sample(p)
    print("half_tank_%d_mg: %d iterations, residual %.3e, max p %.6g, %.1f s" % (N, o.c.last_pcg_iterations, o.c.last_residual, np.abs(p).max(), time.perf_counter() - t0), flush=True)
    o.close()


def rec_dam_break(out, N=2048):
    from euler_amd import scenarios
    o = mg_oracle(N, N)
    o.load_text(scenarios.dam_break(), upscale=True)
    frames, got = 0, []
    while len(got) < FRAMES_AFTER and frames < 400:
        t0 = time.perf_counter()
        o.step()
        frames += 1
        p = o.p
        pmax = float(np.abs(p).max())
        print("  dam_break_%d_mg frame %d: %d substeps, %d iterations, residual %.3e, max p %.4g, %.1f s" % (N, frames, o.c.last_substeps, o.c.last_pcg_iterations, o.c.last_residual, pmax, time.perf_counter() - t0), flush=True)
        # the block falls freely for most of the run: whatever solves happen then work on rounding noise (max p ~ 1e-4) and pin nothing; the record starts when the
        # water has hit the floor and carries a real pressure
        if pmax > IMPACT_P:
            got.append((frames, o.c.last_substeps, o.c.last_pcg_iterations, o.c.last_residual, pmax, o.n_markers, int((o.count > 0).sum()), sample(p),
                        float(np.abs(o.u).max()), float(np.abs(o.v).max())))
    assert len(got) == FRAMES_AFTER
    out["dam_break_%d_mg.scalars" % N] = np.array([[g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[8], g[9]] for g in got], np.float64)
    out["dam_break_%d_mg.p" % N] = np.stack([g[7] for g in got])
    o.close()


def rec_f32_dam_break(out, N=1024, frames_after=3):
    """BASELINE configs[1] "fp32" at its own size: the oracle's FLOAT restatement of the tile-local solve (eo_sim.pcg_f32: every operation rounded to float, sums in double;
    the reference's cap of 100 iterations), free-running from frame 0; the record holds the first frames whose solves run into the cap (the block has landed)."""
    from euler_amd import scenarios
    o = Oracle(N, N)
    o.c.tile_records = 16
    o.c.pcg_f32 = 1
    o.load_text(scenarios.dam_break(), upscale=True)
    frames, got = 0, []
    while len(got) < frames_after and frames < 200:
        t0 = time.perf_counter()
        o.step()
        frames += 1
        print("  dam_break_%d_f32 frame %d: %d substeps, %d iterations, %.1f s" % (N, frames, o.c.last_substeps, o.c.last_pcg_iterations, time.perf_counter() - t0), flush=True)
        if got or o.c.last_pcg_iterations >= 100:
            p = o.p
            got.append((frames, o.c.last_substeps, o.c.last_pcg_iterations, float(np.abs(p).max()), o.n_markers, int((o.count > 0).sum()), float(np.abs(o.u).max()), float(np.abs(o.v).max()),
                        sample(p), sample(o.u), sample(o.v)))
    assert len(got) == frames_after
    out["dam_break_%d_f32.scalars" % N] = np.array([g[:8] for g in got], np.float64)
    out["dam_break_%d_f32.p" % N] = np.stack([g[8] for g in got])
    out["dam_break_%d_f32.u" % N] = np.stack([g[9] for g in got])
    out["dam_break_%d_f32.v" % N] = np.stack([g[10] for g in got])
    o.close()


RECS = {"half_tank_4096_mg": rec_half_tank, "half_tank_8192_mg": lambda out: rec_half_tank(out, 8192), "dam_break_2048_mg": rec_dam_break, "dam_break_1024_f32": rec_f32_dam_break}


def main():
    names = sys.argv[1:] or sorted(RECS)
    out = {}
    if os.path.exists(OUT):
        with np.load(OUT) as z:
            out = {k: z[k] for k in z.files}
    for n in names:
        RECS[n](out)
    out["oracle_sha1"] = np.frombuffer(oracle_sha1().encode(), np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
