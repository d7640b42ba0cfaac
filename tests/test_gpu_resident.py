"""The resident solver (csrc/k_resident.hip): the tile-local PCG of a small grid as ONE persistent launch whose vectors stay in registers, and its
float variant (euler_config.pcg_precision = EULER_PCG_F32: BASELINE configs[1]'s "fp32").

* f64: the multi-kernel tile mode's arithmetic expression for expression; only the dot products fold in another order (per workgroup instead of per
  block), so the two agree like any two tree shapes: the same iteration counts to a few, the same pressure to solver tolerance where the solves
  converge, identical cell grids - and both agree with the oracle's restatement of the tile-local mode within the bar of the tree mode
  (test_gpu_tile_precond.py::test_tree_dot_1024_dam_break_expensive_phase_vs_oracle).
* f32: NOT the reference's iterates (its PCG is double, main.c:577-578,716) - tolerance parity: against the oracle's f64 solve of the same systems and against
  the oracle's own float restatement (eo_sim.pcg_f32)."""
import numpy as np
import pytest

import euler_amd as ea
from euler_amd import scenarios
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


def pair(X, Y, text=None, **kw):
    a = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, **kw)
    b = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, resident=ea.RESIDENT_OFF, **{k: v for k, v in kw.items() if k != "pcg_precision"})
    for s in (a, b):
        if text is None:
            s.load_half_tank()
        else:
            s.load_text(text, upscale=True)
    return a, b


@pytest.mark.parametrize("X,Y,workload,frames", [(512, 512, "dam_break", 40), (300, 200, "waterfall", 30), (1024, 1024, "half_tank", 2), (130, 70, "half_tank", 3),
                                                 # ragged: widths that are no multiple of a chunk, heights that end inside a band, one band, one chunk column
                                                 (257, 129, "waterfall", 20), (333, 200, "dam_break", 30), (65, 300, "dam_break", 30), (1000, 900, "dam_break", 26), (40, 64, "half_tank", 3)])
def test_resident_f64_against_the_multi_kernel_tile_mode(X, Y, workload, frames):
    text = None if workload == "half_tank" else getattr(scenarios, workload)()
    a, b = pair(X, Y, text, max_iterations=2000)      # cap lifted: converged solves compare to tolerance
    assert a.resident_info()[0] and not b.resident_info()[0]
    solved = 0
    for f in range(frames):
        a.step(); b.step()
        sa, sb = a.stats(), b.stats()
        assert sa.last_substeps == sb.last_substeps, f
        assert abs(sa.last_pcg_iterations - sb.last_pcg_iterations) <= 0.02 * sb.last_pcg_iterations + 2 * sb.last_substeps, (f, sa.last_pcg_iterations, sb.last_pcg_iterations)
        assert sa.last_residual <= 1e-6 and sb.last_residual <= 1e-6
        pa, pb = a.get(ea.F_PRESSURE), b.get(ea.F_PRESSURE)
        assert np.abs(pa - pb).max() <= 1e-6 * max(np.abs(pb).max(), 1.0) + 1e-6, (f, np.abs(pa - pb).max(), np.abs(pb).max())
        assert np.array_equal(a.get(ea.F_COUNT), b.get(ea.F_COUNT)), f
        assert np.abs(a.get(ea.F_U) - b.get(ea.F_U)).max() < 1e-4 and np.abs(a.get(ea.F_V) - b.get(ea.F_V)).max() < 1e-4
        solved += sa.last_pcg_iterations > 0
    assert solved >= 2
    info = a.resident_info()
    assert info[1] >= solved and info[2] == 0      # every solve with a right-hand side ran resident, none fell back


def test_resident_capped_solves_take_the_reference_budget():
    """the reference's cap (main.c:735): exactly 100 iterations per substep with tol 0, counters and the last residual in step with the multi-kernel form"""
    a, b = pair(512, 512, None, tol=0.0)
    for f in range(3):
        a.step(); b.step()
        sa, sb = a.stats(), b.stats()
        assert sa.last_substeps == sb.last_substeps and sa.last_pcg_iterations == sb.last_pcg_iterations == 100 * sa.last_substeps
    # a tank at rest, 100 unconverged iterations: the dot products' rounding is amplified; the fields agree loosely, the cell grid exactly
    assert np.array_equal(a.get(ea.F_COUNT) > 0, b.get(ea.F_COUNT) > 0)
    # z, s and A s of a resident solve never leave the registers: the test surface says so instead of showing the arrays' leftovers; p, r, b are there
    for f in (ea.F_PCG_Z, ea.F_PCG_S, ea.F_PCG_Q):
        with pytest.raises(ea.EulerError) as e:
            a.get(f)
        assert e.value.code == -5 and "resident" in str(e.value)      # EULER_ESTATE
        assert b.get(f).shape == (512, 512)
    assert np.isfinite(a.get(ea.F_PCG_R)).all() and np.abs(a.get(ea.F_PRESSURE)).max() > 0


def test_resident_against_the_oracle_tile_mode():
    """one frame of a 384 x 320 dam break in the expensive phase, teacher-forced from the GPU's state: resident solver (tree sums) against the oracle's tile-local mode (sequential sums)"""
    sim = ea.Simulation(384, 320, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=3000).load_text(scenarios.dam_break(), upscale=True)
    for _ in range(60):
        sim.step()
        if sim.stats().last_pcg_iterations > 200:
            break
    o = Oracle(384, 320)
    o.c.tile_records = 16
    o.c.max_iterations = 3000
    for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"), (ea.F_SINK, "sink"),
                 (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon")):
        getattr(o, n)[...] = sim.get(f)
    o.set_markers(sim.get(ea.F_MARKERS))
    o.c.rng_state = sim.stats().rng_state
    sim.step(); o.step()
    st = sim.stats()
    assert st.last_substeps == o.c.last_substeps and abs(st.last_pcg_iterations - o.c.last_pcg_iterations) <= 0.02 * o.c.last_pcg_iterations + 8
    assert np.array_equal(sim.get(ea.F_COUNT), o.count)
    assert np.abs(sim.get(ea.F_U) - o.u).max() <= 1e-5 and np.abs(sim.get(ea.F_V) - o.v).max() <= 1e-5
    assert np.abs(sim.get(ea.F_PRESSURE) - o.p).max() <= 1e-6 * np.abs(o.p).max() + 1e-6
    assert np.array_equal(sim.get(ea.F_PRECON), o.precon)      # E^-1 is element-wise: bit for bit (g_precon persists, main.c:577)


def test_f32_pcg_variant_tolerance_parity():
    """BASELINE configs[1] "fp32": solver vectors in float.  One frame of the 1024^2 dam break in its solving phase under the reference's cap of 100: against the f64
    resident run from the same state - the cell grid is identical, velocities agree to 1e-3 of their magnitude; the stated tolerance of the variant."""
    N = 1024
    f64 = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE).load_text(scenarios.dam_break(), upscale=True)
    for _ in range(60):
        f64.step()
        if f64.stats().last_pcg_iterations >= 100:
            break
    f32 = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, pcg_precision=ea.PCG_F32)
    for f in (ea.F_SOLID, ea.F_SOURCE, ea.F_SINK, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_U, ea.F_V, ea.F_UTMP, ea.F_VTMP, ea.F_PRECON):
        f32.set(f, f64.get(f))
    f32.set_markers(f64.get(ea.F_MARKERS))
    f32.set_rng(f64.stats().rng_state, f64.stats().source_exhausted)
    for k in range(10):
        f64.step(); f32.step()
        assert f32.stats().last_substeps == f64.stats().last_substeps
        vmax = max(np.abs(f64.get(ea.F_U)).max(), np.abs(f64.get(ea.F_V)).max())
        du = max(np.abs(f32.get(ea.F_U) - f64.get(ea.F_U)).max(), np.abs(f32.get(ea.F_V) - f64.get(ea.F_V)).max())
        assert du <= 1e-3 * vmax + 1e-3, (k, du, vmax)
        assert np.array_equal(f32.get(ea.F_COUNT) > 0, f64.get(ea.F_COUNT) > 0), k
    assert f32.resident_info()[1] > 0 and f32.resident_info()[2] == 0


def test_f32_gpu_against_the_oracle_float_restatement():
    """the float variant on the GPU against the oracle's restatement of it (eo_sim.pcg_f32: every operation rounded to float, sums in double), one frame of a 384 x 320
    dam break in its solving phase under the reference's cap, teacher-forced from the GPU's state.  The two differ in the order of the dot products only; a float
    solve amplifies that more than a double one: pressures to 1e-3 of max |p|, velocities to 1e-3 of their magnitude, the same cell grid, E^-1 bit for bit."""
    sim = ea.Simulation(384, 320, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, pcg_precision=ea.PCG_F32).load_text(scenarios.dam_break(), upscale=True)
    for _ in range(60):
        sim.step()
        if sim.stats().last_pcg_iterations >= 100:
            break
    o = Oracle(384, 320)
    o.c.tile_records = 16
    o.c.pcg_f32 = 1
    for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"), (ea.F_SINK, "sink"),
                 (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon")):
        getattr(o, n)[...] = sim.get(f)
    o.set_markers(sim.get(ea.F_MARKERS))
    o.c.rng_state = sim.stats().rng_state
    sim.step(); o.step()
    st = sim.stats()
    assert st.last_substeps == o.c.last_substeps and st.last_pcg_iterations == o.c.last_pcg_iterations
    assert np.array_equal(sim.get(ea.F_COUNT), o.count)
    vmax = max(np.abs(o.u).max(), np.abs(o.v).max())
    assert max(np.abs(sim.get(ea.F_U) - o.u).max(), np.abs(sim.get(ea.F_V) - o.v).max()) <= 1e-3 * vmax + 1e-4
    assert np.abs(sim.get(ea.F_PRESSURE) - o.p).max() <= 1e-3 * np.abs(o.p).max() + 1e-5
    assert np.array_equal(sim.get(ea.F_PRECON), o.precon)
    assert np.array_equal(sim.get(ea.F_PRESSURE), sim.get(ea.F_PRESSURE).astype(np.float32).astype(np.float64))      # p holds float values


def test_f32_variant_at_configs1_size_against_the_recorded_oracle_restatement():
    """BASELINE configs[1] as named (1024^2 dam break, "fp32") against the ORACLE's float restatement at that size (eo_sim.pcg_f32), recorded in the build container
    (tests/golden/mg_records.npz `dam_break_1024_f32`, make_mg_records.py: 30 s of one core): both free-running from frame 0, compared on the first three frames whose solves run
    into the reference's cap (the block has landed: frames 23 - 25).  The two differ in the order of their dot products only, which a float solve amplifies: the same substep and
    marker counts, iteration counts within 5 %, max |p| and the pressure, u, v on a 64 x 64 sample grid within 2e-3 of their maxima (the variant's stated tolerance is 1e-3 per
    frame), the number of fluid cells within 1e-4."""
    import os
    from golden_util import GOLDEN
    from test_gpu_parity import mg_sample
    with np.load(os.path.join(GOLDEN, "mg_records.npz")) as z:
        sc, ps, us, vs = (z["dam_break_1024_f32." + k] for k in ("scalars", "p", "u", "v"))
    N = 1024
    sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, pcg_precision=ea.PCG_F32).load_text(scenarios.dam_break(), upscale=True)
    k, dev = 0, []
    for frame in range(1, int(sc[-1][0]) + 1):
        sim.step()
        if frame != int(sc[k][0]):
            continue
        fr, nsub, its, pmax, nmark, nfluid, umax, vmax = sc[k]
        st = sim.stats()
        p, u, v = sim.get(ea.F_PRESSURE), sim.get(ea.F_U), sim.get(ea.F_V)
        dev.append((frame, st.last_substeps - int(nsub), st.last_pcg_iterations, int(its), st.n_markers - int(nmark), int((sim.get(ea.F_COUNT) > 0).sum()) - int(nfluid),
                    float(np.abs(p).max() / pmax - 1.0), float(np.abs(mg_sample(p) - ps[k]).max() / pmax),
                    float(max(np.abs(mg_sample(u) - us[k]).max(), np.abs(mg_sample(v) - vs[k]).max()) / max(umax, vmax))))
        k += 1
    print(dev)
    assert k == len(sc) == 3
    for frame, dsub, it, it_o, dmark, dfluid, dpm, dp, duv in dev:
        assert dsub == 0 and dmark == 0, dev
        assert abs(it - it_o) <= 0.05 * it_o + 3 and abs(dfluid) <= 1e-4 * nfluid + 2, dev
        assert abs(dpm) <= 2e-3 and dp <= 2e-3 and duv <= 2e-3, dev
    assert sim.resident_info()[1] > 0 and sim.resident_info()[2] == 0      # (the one-launch solver ran them)
    sim.close()


def test_a_scene_that_outgrows_the_chip_moves_to_the_multi_kernel_path():
    """The resident launch is sized from the PREVIOUS solve's active chunks (no host round trip in front of it); a solve that does not fit after all says so on the
    device (error word 2), is redone with the multi-kernel path, and the following ones go there directly.  Driven here by a capacity of 8 workgroups (32 chunks,
    EULER_OPT_RESIDENT_CAP on this handle only) under a waterfall whose water keeps growing: the run equals the multi-kernel run to the tree mode's tolerance, nothing timed out."""
    from euler_amd import scenarios
    a = ea.Simulation(300, 200, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=2000).load_text(scenarios.waterfall(), upscale=True)
    a.set_option(ea.OPT_RESIDENT_CAP, 8)
    b = ea.Simulation(300, 200, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=2000, resident=ea.RESIDENT_OFF).load_text(scenarios.waterfall(), upscale=True)
    solves, worst = 0, 0.0
    for f in range(70):
        a.step(); b.step()
        solves += a.stats().last_substeps if a.stats().last_pcg_iterations else 0
        assert a.stats().last_substeps == b.stats().last_substeps
        assert np.array_equal(a.get(ea.F_COUNT), b.get(ea.F_COUNT)), f
        pa, pb = a.get(ea.F_PRESSURE), b.get(ea.F_PRESSURE)
        worst = max(worst, float(np.abs(pa - pb).max() / max(np.abs(pb).max(), 1.0)))
    info = a.resident_info()
    assert info[0] and info[2] == 0                 # still eligible, no time-out
    assert 0 < info[1] < solves, (info, solves)     # the early solves ran resident, the later ones did not fit
    assert worst <= 1e-6, worst
    a.close(); b.close()


@pytest.mark.parametrize("precision", [ea.PCG_F64, ea.PCG_F32])
def test_a_resident_launch_that_times_out_is_redone_by_the_multi_kernel_path(precision):
    """error word 1 - a wait inside the persistent launch ran out (the workgroups were not all resident: a shared or CU-masked device) - was never reached by a
    test.  EULER_OPT_RESIDENT_FORCE_TIMEOUT makes the next launch give up at once with that word.  The host then solves the SAME system with the multi-kernel path and
    keeps to it on that handle: the frame equals a RESIDENT_OFF run bit for bit from there on (double), the bookkeeping of the solve before (iteration counts, the
    look-ahead poll) is the previous solve's and not the aborted launch's, euler_resident_info counts one fallback.  A float handle goes on in DOUBLE (the substep's
    markers have moved when the solve starts: failing there would leave the handle between two stages) - its pressures then equal the double handle's."""
    from euler_amd import scenarios
    kw = dict(dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=300)
    a = ea.Simulation(384, 320, pcg_precision=precision, **kw).load_half_tank()      # (a tank at rest: every solve iterates from the first frame on)
    b = ea.Simulation(384, 320, resident=ea.RESIDENT_OFF, **kw).load_half_tank()
    for f in range(2):
        a.step(); b.step()
    assert a.resident_info()[1] > 0 and a.resident_info()[2] == 0
    tol = 1e-6 if precision == ea.PCG_F64 else 2e-3
    assert np.abs(a.get(ea.F_PRESSURE) - b.get(ea.F_PRESSURE)).max() <= tol * max(np.abs(b.get(ea.F_PRESSURE)).max(), 1.0)
    before = a.stats().total_pcg_iterations
    a.set_option(ea.OPT_RESIDENT_FORCE_TIMEOUT, 1)
    a.step(); b.step()
    info = a.resident_info()
    assert info[2] == 1 and not info[0], info                     # one fallback; the handle keeps to the multi-kernel path
    assert a.get_option(ea.OPT_RESIDENT_FORCE_TIMEOUT) == 0
    assert a.stats().total_pcg_iterations > before and a.stats().last_substeps == b.stats().last_substeps
    # from the redone solve on the handle runs what the RESIDENT_OFF handle runs (the substeps before ran resident - sums folded per workgroup, in float on the float
    # handle - so the states agree to that mode's tolerance, not to the bit)
    pa, pb = a.get(ea.F_PRESSURE), b.get(ea.F_PRESSURE)
    assert np.abs(pa - pb).max() <= tol * max(np.abs(pb).max(), 1.0)
    for f in range(3):
        a.step(); b.step()
        assert a.stats().last_pcg_iterations > 0 and a.stats().last_substeps == b.stats().last_substeps
        assert np.array_equal(a.get(ea.F_COUNT) > 0, b.get(ea.F_COUNT) > 0), f
    assert np.abs(a.get(ea.F_PRESSURE) - b.get(ea.F_PRESSURE)).max() <= 10 * tol * max(np.abs(b.get(ea.F_PRESSURE)).max(), 1.0)
    assert a.resident_info()[2] == 1
    a.close(); b.close()


def test_f32_needs_the_resident_solver():
    with pytest.raises(ea.EulerError):
        ea.Simulation(4096, 4096, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, pcg_precision=ea.PCG_F32)      # too many chunks
    with pytest.raises(ea.EulerError):
        ea.Simulation(512, 512, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0, pcg_precision=ea.PCG_F32)             # the reference's IC(0) is a double path
    s = ea.Simulation(512, 512, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, pcg_precision=ea.PCG_F32)
    with pytest.raises(ea.EulerError):
        s.set_precond(ea.PRECOND_IC0_TILE_MG, 16)
