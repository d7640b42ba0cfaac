"""EULER_PRECOND_IC0_TILE on the GPU (k_sweep_skew<OP, false, true>) against the oracle's restatement of the same
blocks (oracle eo_tile_start): bit-exact in EULER_DOT_SEQUENTIAL - each block's recurrences have no reduction - and, against
the REFERENCE's IC(0), tolerance parity where PCG converges (the mode changes the iterates, not the solution)."""
import numpy as np
import pytest

import euler_amd as ea
from golden_util import load, scenario_text
from oracle_lib import Oracle
from test_gpu_parity import assert_bits, compare_all

pytestmark = pytest.mark.gpu

TILE_SHAPES = [((144, 100), 1), ((400, 130), 1), ((400, 130), 2), ((1024, 200), 1), ((1024, 200), 3), ((1025, 130), 6),
               ((65, 300), 1), ((2000, 70), 2), ((333, 127), 1000)]


@pytest.mark.parametrize("sweep", [ea.SWEEP_BAND, ea.SWEEP_SIMPLE])
@pytest.mark.parametrize("shape,units", TILE_SHAPES)
def test_tile_sweeps_random_masks_bit_exact(shape, units, sweep):
    """Factor (stale entries of non-fluid neighbours inside a block included), forward and backward solve of the tile-local
    mode on random fluid / solid patterns; widths of one and of many tiles, partial top band, both schedules."""
    X2, Y2 = shape
    rng = np.random.default_rng(X2 * 1000 + Y2 + units)
    count = np.zeros((Y2, X2), np.uint8)
    solid = np.zeros((Y2, X2), np.uint8)
    inner = (slice(1, Y2 - 1), slice(1, X2 - 1))
    count[inner] = (rng.random((Y2 - 2, X2 - 2)) < 0.7) * rng.integers(1, 5, (Y2 - 2, X2 - 2))
    solid[inner] = rng.random((Y2 - 2, X2 - 2)) < 0.1
    count[solid > 0] = 0
    if X2 > 600:
        count[:, 300:520] = 0          # whole tiles without fluid: skipped by every sweep
    sink = np.zeros((Y2, X2), np.uint8)
    sink[0, :] = sink[-1, :] = sink[:, 0] = sink[:, -1] = 1
    stale = rng.random((Y2, X2)) * (rng.random((Y2, X2)) < 0.5)
    fluid = count > 0
    r = np.where(fluid, rng.standard_normal((Y2, X2)), 0.0)
    zero_f = np.zeros((Y2, X2), np.float32)

    o = Oracle(X2, Y2)
    o.c.tile_units = units
    o.count[...] = count; o.solid[...] = solid; o.sink[...] = sink
    o.precon[...] = stale
    o.lib.eo_build_system(o.ptr, np.float32(0.1), o.f32p(o.utmp), o.f32p(o.vtmp))
    o.r[...] = r
    o.lib.eo_apply_preconditioner(o.ptr, o.f64p(o.r), o.f64p(o.z))

    sim = ea.Simulation(X2, Y2, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=sweep, precond=ea.PRECOND_IC0_TILE, tile_units=units)
    for f, a in ((ea.F_SOLID, solid), (ea.F_SOURCE, np.zeros_like(solid)), (ea.F_SINK, sink), (ea.F_COUNT, count),
                 (ea.F_PREV_COUNT, count), (ea.F_UTMP, zero_f), (ea.F_VTMP, zero_f), (ea.F_PRECON, stale)):
        sim.set(f, a)
    sim.set_markers(np.zeros((0, 2), np.float32))
    sim.pcg_op(ea.OP_BUILD_SYSTEM, 0.1)
    sim.set(ea.F_PCG_R, r)
    for rep in range(2):
        sim.pcg_op(ea.OP_PRECON_FACTOR)
        assert_bits(sim.get(ea.F_PRECON), o.precon, "precon %s rep %d" % (shape, rep), nan_class=True)
        sim.pcg_op(ea.OP_FORWARD_SOLVE)
        assert_bits(sim.get(ea.F_PCG_Q), o.q, "q %s" % (shape,), nan_class=True)
        sim.pcg_op(ea.OP_BACKWARD_SOLVE)
        assert_bits(sim.get(ea.F_PCG_Z), o.z, "z %s" % (shape,), nan_class=True)
        o.lib.eo_apply_preconditioner(o.ptr, o.f64p(o.r), o.f64p(o.z))


@pytest.mark.parametrize("size,scn,units,frames", [((320, 192), "weird-edges", 1, 6), ((512, 256), "block", 2, 4),
                                                   ((257, 129), "filter", 1, 8), ((130, 70), "block", 6, 12)])
def test_tile_mode_free_running_bit_exact_vs_oracle(size, scn, units, frames):
    text = scenario_text(load(scn + "_frames.npz"))
    o = Oracle(size[0], size[1]).load_text(text, upscale=True)
    o.c.tile_units = units
    sim = ea.Simulation(size[0], size[1], dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0_TILE, tile_units=units).load_text(text, upscale=True)
    for f in range(frames):
        o.step()
        sim.step()
        st = sim.stats()
        assert st.last_substeps == o.c.last_substeps and st.last_pcg_iterations == o.c.last_pcg_iterations, f
        compare_all(o, sim, "%s %s frame %d" % (scn, size, f))


def test_tile_mode_reaches_the_reference_pressure_where_pcg_converges():
    """Against the REFERENCE's preconditioner (oracle, exact IC(0)): 256x256 half tank, tolerance 1e-6, enough iterations
    for both: |dp| <= 1e-5 max|p|, identical cell grids, velocities within 1e-5; iteration counts printed (-s)."""
    o = Oracle(256, 256).load_half_tank()
    o.c.max_iterations = 3000
    sim = ea.Simulation(256, 256, dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0_TILE, tile_units=1, max_iterations=3000).load_half_tank()
    for f in range(2):
        o.step()
        sim.step()
    st = sim.stats()
    print("iterations to 1e-6: reference IC(0) %d, tile-local (64 x 96 blocks) %d" % (o.c.total_pcg_iterations, st.total_pcg_iterations))
    assert st.last_residual <= 1e-6 and o.c.last_residual <= 1e-6
    p, pr = sim.get(ea.F_PRESSURE), o.p
    assert np.abs(p - pr).max() <= 1e-5 * np.abs(pr).max()
    assert_bits(sim.get(ea.F_COUNT) > 0, o.count > 0, "fluid/air grid")
    assert np.abs(sim.get(ea.F_U) - o.u).max() < 1e-5 and np.abs(sim.get(ea.F_V) - o.v).max() < 1e-5
    assert st.total_pcg_iterations <= 1.5 * o.c.total_pcg_iterations


def test_tile_mode_tree_dot_1024_vs_oracle():
    """The production combination at BASELINE's 1024^2: tile-local IC(0) + EULER_DOT_TREE, 2 frames of the half tank
    (100 iterations each) against the oracle's tile mode: only the dot products round differently.
    Tolerance: |dp| <= 1e-9 max|p|, velocities within 1e-9, cell grid identical."""
    o = Oracle(1024, 1024).load_half_tank()
    o.c.tile_units = 6
    sim = ea.Simulation(1024, 1024, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, tile_units=6).load_half_tank()
    for f in range(2):
        o.step()
        sim.step()
    assert sim.stats().total_pcg_iterations == o.c.total_pcg_iterations == 200
    pr = o.p
    assert np.abs(sim.get(ea.F_PRESSURE) - pr).max() <= 1e-9 * np.abs(pr).max()
    assert_bits(sim.get(ea.F_COUNT), o.count, "count")
    assert np.abs(sim.get(ea.F_U) - o.u).max() < 1e-9 and np.abs(sim.get(ea.F_V) - o.v).max() < 1e-9


def test_switching_the_preconditioner_on_a_live_handle():
    """euler_set_precond: exact IC(0) -> tile-local -> exact on one handle equals fresh handles of each mode (g_precon is
    rewritten on every fluid cell by each factorisation; stale non-fluid entries are shared state)."""
    text = scenario_text(load("block_frames.npz"))
    a = ea.Simulation(300, 200, dot_mode=ea.DOT_SEQUENTIAL).load_text(text, upscale=True)
    o = Oracle(300, 200).load_text(text, upscale=True)
    for units in (0, 2, 0):
        a.set_precond(ea.PRECOND_IC0_TILE if units else ea.PRECOND_IC0, units)
        o.c.tile_units = units
        for _ in range(3):
            a.step()
            o.step()
        compare_all(o, a, "units %d" % units)
