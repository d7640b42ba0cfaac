#!/usr/bin/env python3
"""Soak run (development aid): hundreds of frames at 1024^2 / 2048^2 in both preconditioner modes, checking that no in-kernel
wait ever times out, every field stays finite, markers are conserved where nothing creates or deletes them, and - native grid,
all five scenarios, hundreds of frames - that the tile-local mode stays bit-identical to the oracle's restatement of it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import euler_amd as ea
from euler_amd import scenarios

print("| grid | scenario | mode | frames | substeps | PCG iterations | markers | fluid cells | max abs u | wall |")
print("|---|---|---|---|---|---|---|---|---|---|")
for size, wl, frames in ((1024, "dam", 300), (2048, "waterfall", 120)):
    for pc, name in ((ea.PRECOND_IC0, "parity"), (ea.PRECOND_IC0_TILE, "roofline")):
        sim = ea.Simulation(size, size, dot_mode=ea.DOT_TREE, precond=pc)
        sim.load_text(scenarios.dam_break() if wl == "dam" else scenarios.waterfall(), upscale=True)
        n0 = sim.stats().n_markers
        t0 = time.time()
        for f in range(frames):
            sim.step()
        st = sim.stats()
        u = sim.get(ea.F_U)
        assert np.isfinite(u).all() and np.isfinite(sim.get(ea.F_V)).all() and np.isfinite(sim.get(ea.F_MARKERS)).all()
        if wl == "dam":
            assert st.n_markers == n0
        print("| %d^2 | %s | %s | %d | %d | %d | %d | %d | %.1f | %.1f s |" % (size, wl, name, st.frames, st.total_substeps, st.total_pcg_iterations,
              st.n_markers, st.fluid_cells, np.abs(u).max(), time.time() - t0))
        sim.close()

# native grid, free-running, roofline mode against the oracle's tile mode, bit for bit
from golden_util import SCENARIOS, bits_equal, load, scenario_text
from oracle_lib import Oracle
print("\n| scenario (100x40, EULER_DOT_SEQUENTIAL, tile-local mode) | frames bit-identical to the oracle | PCG iterations |")
print("|---|---|---|")
for scn in SCENARIOS:
    text = scenario_text(load(scn + "_frames.npz"))
    frames = 500 if scn == "waterfall" else 300
    sim = ea.Simulation(100, 40, dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0_TILE, tile_records=16).load_text(text)
    o = Oracle(100, 40).load_text(text)
    o.c.tile_records = 16
    for f in range(frames):
        sim.step()
        o.step()
        if f % 25 == 24 or f == frames - 1:
            for fld, want in ((ea.F_U, o.u), (ea.F_V, o.v), (ea.F_COUNT, o.count), (ea.F_MARKERS, o.markers), (ea.F_PRESSURE, o.p)):
                assert bits_equal(sim.get(fld), want), (scn, f, fld)
    assert sim.stats().rng_state == o.c.rng_state
    print("| %s | %d | %d |" % (scn, frames, sim.stats().total_pcg_iterations))
