"""GPU parity tests proper: the HIP path, called through the C ABI (include/euler.h), against
  (1) the golden fixtures generated from the compiled reference (100x40, bit-exact), and
  (2) the CPU oracle on the same seeded inputs at sizes the oracle finishes in seconds.

Bars: cell grids, marker arrays (order included), velocities and the persistent preconditioner
are BIT-EXACT with EULER_DOT_SEQUENTIAL; with EULER_DOT_TREE (the large-grid default) only the
dot products differ in rounding, and fields must agree to the tolerance stated in each test.
"""
import numpy as np
import pytest

import euler_amd as ea
from golden_util import SCENARIOS, X, Y, bits_equal, load, scenario_text
from oracle_lib import Oracle, fnv1a64

pytestmark = pytest.mark.gpu


def hashes(sim):
    return [fnv1a64(sim.get(ea.F_U)), fnv1a64(sim.get(ea.F_V)), fnv1a64(sim.get(ea.F_COUNT)), fnv1a64(sim.get(ea.F_MARKERS))]


def assert_bits(got, want, what, nan_class=False):
    """nan_class: NaN payload/sign bits are implementation-defined (0*inf on x86 vs gfx950), so where
    BOTH sides hold a NaN they count as equal; everything else stays bit-exact."""
    if bits_equal(got, want):
        return
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    if nan_class and got.shape == want.shape and got.dtype.kind == "f":
        both = np.isnan(got) & np.isnan(want)
        if both.any():
            got, want = got.copy(), want.copy()
            got[both] = 0
            want[both] = 0
            if bits_equal(got, want):
                return
    if got.shape != want.shape or got.dtype != want.dtype:
        raise AssertionError("%s: %s%s != %s%s" % (what, got.dtype, got.shape, want.dtype, want.shape))
    ut = {1: np.uint8, 4: np.uint32, 8: np.uint64}[got.dtype.itemsize]
    bad = np.argwhere(got.view(ut) != want.view(ut))
    first = tuple(bad[0])
    raise AssertionError("%s: %d of %d entries differ, first at %s: got %r want %r" %
                         (what, len(bad), got.size, first, got[first], want[first]))


def test_device_is_mi355x():
    sim = ea.Simulation(X, Y)
    assert "gfx950" in sim.device_name()


@pytest.mark.parametrize("scn", SCENARIOS)
def test_init_matches_reference(scn):
    g = load(scn + "_frames.npz")
    sim = ea.Simulation(X, Y).load_text(scenario_text(g))
    for f, n in ((ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"), (ea.F_SINK, "sink")):
        assert_bits(sim.get(f), g[n], n)
    assert_bits(sim.get(ea.F_COUNT), g["init_count"], "count")
    assert_bits(sim.get(ea.F_PREV_COUNT), np.zeros((Y, X), np.uint8), "prev_count")
    assert_bits(sim.get(ea.F_MARKERS), g["init_markers"], "markers")
    assert int(sim.stats().rng_state) == int(g["init_rng"])


@pytest.mark.parametrize("sweep", [ea.SWEEP_BAND, ea.SWEEP_SIMPLE])
@pytest.mark.parametrize("scn", SCENARIOS)
def test_free_running_bit_exact_vs_reference(scn, sweep):
    """The whole sim_step path, free-running from the scenario text, against the compiled
    reference's per-frame hashes and full states (u, v, counts, markers in order, g_precon)."""
    g = load(scn + "_frames.npz")
    nframes = len(g["hashes"])
    if sweep == ea.SWEEP_SIMPLE:
        nframes = min(nframes, 40)
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=sweep).load_text(scenario_text(g))
    keep = set(int(f) for f in g["frames_full"])
    for f in range(nframes):
        sim.step()
        st = sim.stats()
        assert st.last_substeps == int(g["n_substeps"][f]), (scn, f)
        assert st.n_markers == int(g["n_markers"][f]), (scn, f)
        if f in keep:
            for fld, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"),
                           (ea.F_PRECON, "precon"), (ea.F_MARKERS, "markers")):
                assert_bits(sim.get(fld), g["f%d_%s" % (f, n)], "%s frame %d %s" % (scn, f, n))
            assert st.source_exhausted == int(g["f%d_exhausted" % f])
            assert int(st.rng_state) == int(g["f%d_rng" % f])
        assert hashes(sim) == [int(h) for h in g["hashes"][f]], (scn, f)
    assert sim.stats().marker_multi_events == 0


def test_block_known_answer_counters():
    g = load("block_frames.npz")
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL).load_text(scenario_text(g))
    for _ in range(100):
        sim.step()
    st = sim.stats()
    assert (st.total_substeps, st.total_pcg_iterations, st.n_markers, st.fluid_cells) == (359, 15997, 4488, 1117)


def load_substep_state(sim, g):
    for f, n in ((ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"), (ea.F_SINK, "sink")):
        sim.set(f, g[n])
    for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_COUNT, "count"),
                 (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon")):
        sim.set(f, g["before_" + n])
    sim.set_markers(g["before_markers"])
    sim.set_rng(int(g["rng_before"]), int(g["exhausted_before"]))


# reference stage index (make_golden.run_stages) after which each fused GPU stage must agree
GPU_STAGES = [
    (ea.STAGE_ADVECT_MARKERS, 0), (ea.STAGE_REFRESH_COUNTS, 1), (ea.STAGE_SOURCES, 2),
    (ea.STAGE_EXTRAPOLATE, 6), (ea.STAGE_ADVECT_VELOCITY, 11), (ea.STAGE_PROJECT, 12),
]
FIELDS = ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_COUNT, "count"),
          (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon"), (ea.F_MARKERS, "markers"))


@pytest.mark.parametrize("scn", SCENARIOS)
def test_teacher_forced_stages_vs_reference(scn):
    g = load(scn + "_substep.npz")
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL)
    load_substep_state(sim, g)
    dt = float(g["dt"])
    nstages = len(g["stage_names"])
    expect = {n: g["before_" + n] for _, n in FIELDS}
    ref_i = 0
    for stage, upto in GPU_STAGES:
        while ref_i <= upto and ref_i < nstages:
            for _, n in FIELDS:
                key = "s%02d_%s" % (ref_i, n)
                if key in g:
                    expect[n] = g[key]
            ref_i += 1
        sim.stage(stage, dt)
        for fld, n in FIELDS:
            if stage == ea.STAGE_ADVECT_VELOCITY and n in ("u", "v"):
                continue
            assert_bits(sim.get(fld), expect[n], "%s stage %d %s" % (scn, stage, n))
    st = sim.stats()
    assert int(st.rng_state) == int(g["rng_after"]) and st.source_exhausted == int(g["exhausted_after"])


def test_filter_substep_exercises_dt_chain():
    """The filter fixture contains a collision that shortens dt for later markers (main.c:501)."""
    g = load("filter_substep.npz")
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL)
    load_substep_state(sim, g)
    sim.stage(ea.STAGE_ADVECT_MARKERS, float(g["dt"]))
    assert sim.stats().marker_dt_events >= 1
    assert_bits(sim.get(ea.F_MARKERS), g["s00_markers"], "markers")


# ----------------------------------------------------------------------------- vs the oracle, ragged / larger grids
def make_pair(X2, Y2, scn="block", oracle=True, **kw):
    text = scenario_text(load(scn + "_frames.npz"))
    o = Oracle(X2, Y2).load_text(text, upscale=True) if oracle else None
    sim = ea.Simulation(X2, Y2, **kw).load_text(text, upscale=True)
    return o, sim


def compare_all(o, sim, what):
    """every observable bit for bit; `o` is a live Oracle or the Recorded stand-in of a trajectory (tests/trajectories.py: digests, live replay on a mismatch)"""
    from trajectories import same
    same(sim.get(ea.F_COUNT), o, "count", what + " count")
    same(sim.get(ea.F_PREV_COUNT), o, "prev_count", what + " prev_count")
    same(sim.get(ea.F_MARKERS), o, "markers", what + " markers")
    same(sim.get(ea.F_U), o, "u", what + " u")
    same(sim.get(ea.F_V), o, "v", what + " v")
    same(sim.get(ea.F_PRECON), o, "precon", what + " precon")
    same(sim.get(ea.F_PRESSURE), o, "p", what + " p")


@pytest.mark.parametrize("size,scn,frames", [((130, 70), "block", 12), ((257, 129), "filter", 8),
                                             ((192, 200), "waterfall", 10), ((320, 192), "weird-edges", 6),
                                             ((112, 48), "filter", 40), ((144, 200), "block", 30)])
def test_ragged_grids_bit_exact_vs_oracle(size, scn, frames):
    """Sizes that are not multiples of the 64-row band / 64-lane tiles, several bands deep."""
    from trajectories import oracle_for
    _, sim = make_pair(size[0], size[1], scn, dot_mode=ea.DOT_SEQUENTIAL, oracle=False)
    o = oracle_for("ragged_%dx%d_%s" % (size[0], size[1], scn))
    compare_all(o, sim, "init")
    for f in range(frames):
        o.step()
        sim.step()
        st = sim.stats()
        assert st.last_substeps == o.c.last_substeps and st.last_pcg_iterations == o.c.last_pcg_iterations, f
        compare_all(o, sim, "%s %s frame %d" % (scn, size, f))


def test_512_dam_break_bit_exact_vs_oracle():
    o, sim = make_pair(512, 512, "block", dot_mode=ea.DOT_SEQUENTIAL)
    for f in range(2):
        o.step()
        sim.step()
        compare_all(o, sim, "512 frame %d" % f)


def test_tree_dot_mode_within_tolerance():
    """EULER_DOT_TREE changes only the rounding of the three dot products per iteration.
    Tolerance: velocities within 1e-4 absolute (|u| ~ 1..10), cell-type grid identical, over the
    first 10 frames of a 256x256 dam break (the system is chaotic beyond that horizon)."""
    o, sim = make_pair(256, 256, "block", dot_mode=ea.DOT_TREE)
    for f in range(10):
        o.step()
        sim.step()
    assert_bits(sim.get(ea.F_COUNT) > 0, o.count > 0, "fluid/air grid")
    assert np.abs(sim.get(ea.F_U) - o.u).max() < 1e-4
    assert np.abs(sim.get(ea.F_V) - o.v).max() < 1e-4


# ----------------------------------------------------------------------------- kernel-level PCG parity
def test_pcg_kernels_vs_oracle():
    rng = np.random.default_rng(7)
    o, sim = make_pair(200, 150, "block", dot_mode=ea.DOT_SEQUENTIAL)
    for _ in range(3):
        o.step()
        sim.step()
    dt = 0.01
    # advect both to the same pre-projection state
    sim.stage(ea.STAGE_ADVECT_VELOCITY, dt)
    o.lib.eo_advect_u(o.ptr, o.f32p(o.u), o.f32p(o.v), np.float32(dt), o.f32p(o.utmp))
    o.lib.eo_advect_v(o.ptr, o.f32p(o.u), o.f32p(o.v), np.float32(dt), o.f32p(o.vtmp))
    o.lib.eo_apply_body_forces(o.ptr, o.f32p(o.vtmp), np.float32(dt))
    o.lib.eo_zero_bounds(o.ptr, o.f32p(o.utmp), 1)
    o.lib.eo_zero_bounds(o.ptr, o.f32p(o.vtmp), 2)
    assert_bits(sim.get(ea.F_UTMP), o.utmp, "utmp")
    assert_bits(sim.get(ea.F_VTMP), o.vtmp, "vtmp")
    sim.pcg_op(ea.OP_BUILD_SYSTEM, dt)
    o.lib.eo_build_system(o.ptr, np.float32(dt), o.f32p(o.utmp), o.f32p(o.vtmp))
    assert_bits(sim.get(ea.F_PCG_B), o.b, "b")
    fluid = o.count > 0
    mask = sim.get(ea.F_CELLMASK)
    assert_bits((mask & 1) > 0, fluid, "mask fluid bit")
    assert_bits((mask >> 5)[fluid], o.a_diag[fluid].astype(np.uint8), "a_diag")
    # random vectors, zero outside the fluid as the solver keeps them
    r = np.where(fluid, rng.standard_normal(fluid.shape), 0.0)
    s = np.where(fluid, rng.standard_normal(fluid.shape), 0.0)
    o.r[...] = r
    o.s[...] = s
    sim.set(ea.F_PCG_R, r)
    sim.set(ea.F_PCG_S, s)
    # preconditioner = factor + forward + backward
    o.lib.eo_apply_preconditioner(o.ptr, o.f64p(o.r), o.f64p(o.z))
    sim.pcg_op(ea.OP_PRECON_FACTOR)
    assert_bits(sim.get(ea.F_PRECON), o.precon, "precon")
    sim.pcg_op(ea.OP_FORWARD_SOLVE)
    assert_bits(sim.get(ea.F_PCG_Q), o.q, "q")
    sim.pcg_op(ea.OP_BACKWARD_SOLVE)
    assert_bits(sim.get(ea.F_PCG_Z), o.z, "z")
    assert sim.pcg_op(ea.OP_DOT_ZR) == o.lib.eo_dot(o.ptr, o.f64p(o.z), o.f64p(o.r))
    assert sim.pcg_op(ea.OP_INF_NORM_R) == o.lib.eo_inf_norm(o.ptr, o.f64p(o.r))
    # apply_a
    o.lib.eo_apply_a(o.ptr, o.f64p(o.s), o.f64p(o.z))
    sim.pcg_op(ea.OP_APPLY_A)
    assert_bits(sim.get(ea.F_PCG_Z), o.z, "A s")
    assert sim.pcg_op(ea.OP_DOT_ZS) == o.lib.eo_dot(o.ptr, o.f64p(o.z), o.f64p(o.s))
    # A is symmetric: <x, A y> == <A x, y> up to rounding (SURVEY.md §4 property check)
    x = s
    y = r
    Ax = sim.get(ea.F_PCG_Z).copy()
    sim.set(ea.F_PCG_S, y)
    sim.pcg_op(ea.OP_APPLY_A)
    Ay = sim.get(ea.F_PCG_Z)
    assert abs((x * Ay).sum() - (Ax * y).sum()) < 1e-9 * np.abs(x * Ay).sum()


SWEEP_SHAPES = [(144, 100), (128, 70), (64, 300), (1024, 64), (208, 130), (96, 64), (100, 40), (333, 127),
                (257, 200), (65, 1100), (1025, 130), (80, 640)]   # X+63 multiple of 16; > 8 bands (two workgroups)


@pytest.mark.parametrize("sweep", [ea.SWEEP_BAND, ea.SWEEP_SIMPLE])
@pytest.mark.parametrize("shape", SWEEP_SHAPES)
def test_ic0_sweeps_random_masks_bit_exact(shape, sweep):
    """The three IC(0) sweeps (E^-1 factor incl. STALE entries of non-fluid neighbours, forward,
    backward) on random fluid/solid patterns and random vectors, every schedule, bit-exact.
    Shapes cover widths that are / are not multiples of the 32-column hand-off block, one band and
    many bands, and a partial top band."""
    X2, Y2 = shape
    rng = np.random.default_rng(X2 * 1000 + Y2)
    count = np.zeros((Y2, X2), np.uint8)
    solid = np.zeros((Y2, X2), np.uint8)
    inner = (slice(1, Y2 - 1), slice(1, X2 - 1))
    count[inner] = (rng.random((Y2 - 2, X2 - 2)) < 0.7) * rng.integers(1, 5, (Y2 - 2, X2 - 2))
    solid[inner] = rng.random((Y2 - 2, X2 - 2)) < 0.1
    count[solid > 0] = 0
    sink = np.zeros((Y2, X2), np.uint8)
    sink[0, :] = sink[-1, :] = sink[:, 0] = sink[:, -1] = 1
    stale = rng.random((Y2, X2)) * (rng.random((Y2, X2)) < 0.5)      # old precon values, some zero
    fluid = count > 0
    r = np.where(fluid, rng.standard_normal((Y2, X2)), 0.0)
    zero_f = np.zeros((Y2, X2), np.float32)

    o = Oracle(X2, Y2)
    o.count[...] = count; o.solid[...] = solid; o.sink[...] = sink
    o.precon[...] = stale
    o.utmp[...] = 0; o.vtmp[...] = 0
    o.lib.eo_build_system(o.ptr, np.float32(0.1), o.f32p(o.utmp), o.f32p(o.vtmp))
    o.r[...] = r
    o.lib.eo_apply_preconditioner(o.ptr, o.f64p(o.r), o.f64p(o.z))

    sim = ea.Simulation(X2, Y2, dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=sweep)
    for f, a in ((ea.F_SOLID, solid), (ea.F_SOURCE, np.zeros_like(solid)), (ea.F_SINK, sink), (ea.F_COUNT, count),
                 (ea.F_PREV_COUNT, count), (ea.F_UTMP, zero_f), (ea.F_VTMP, zero_f), (ea.F_PRECON, stale)):
        sim.set(f, a)
    sim.set_markers(np.zeros((0, 2), np.float32))
    sim.pcg_op(ea.OP_BUILD_SYSTEM, 0.1)
    sim.set(ea.F_PCG_R, r)
    for rep in range(2):       # twice: the second factorisation starts from the first one's output, like the reference
        sim.pcg_op(ea.OP_PRECON_FACTOR)
        # random masks can contain an isolated fluid cell (a = 0 -> precon = inf -> NaN, main.c:593-597)
        assert_bits(sim.get(ea.F_PRECON), o.precon, "precon %s rep %d" % (shape, rep), nan_class=True)
        sim.pcg_op(ea.OP_FORWARD_SOLVE)
        assert_bits(sim.get(ea.F_PCG_Q), o.q, "q %s" % (shape,), nan_class=True)
        sim.pcg_op(ea.OP_BACKWARD_SOLVE)
        assert_bits(sim.get(ea.F_PCG_Z), o.z, "z %s" % (shape,), nan_class=True)
        o.lib.eo_apply_preconditioner(o.ptr, o.f64p(o.r), o.f64p(o.z))


def test_band_and_simple_sweeps_agree_bitwise():
    sims = []
    for mode in (ea.SWEEP_BAND, ea.SWEEP_SIMPLE):
        _, sim = make_pair(300, 260, "block", dot_mode=ea.DOT_SEQUENTIAL, sweep_mode=mode)
        for _ in range(3):
            sim.step()
        sims.append(sim)
    for f in (ea.F_U, ea.F_V, ea.F_PRECON, ea.F_PRESSURE, ea.F_MARKERS):
        assert_bits(sims[0].get(f), sims[1].get(f), "field %d" % f)


def test_post_projection_divergence_small_when_converged():
    """Property check: whenever the last solve of a frame converged (residual <= tol),
    the projected field is divergence-free - checked on fluid cells whose own and four neighbours'
    pressures are positive, i.e. untouched by the p >= 0 clamp (main.c:773-779).
    div(u) = dt * residual <= 1e-7 in exact arithmetic; float velocities of magnitude ~10 add ~1e-5."""
    sim = ea.Simulation(X, Y).load_text(scenario_text(load("block_frames.npz")))
    checked = 0
    for _ in range(60):
        sim.step()
        st = sim.stats()
        if st.last_pcg_iterations == 0 or st.last_residual > 1e-6:
            continue
        u, v, cnt = sim.get(ea.F_U), sim.get(ea.F_V), sim.get(ea.F_COUNT)
        p = sim.get(ea.F_PRESSURE)
        div = np.zeros_like(u)
        div[1:, 1:] = u[1:, 1:] - u[1:, :-1] + v[1:, 1:] - v[:-1, 1:]
        ok = (cnt > 0) & (p > 0)
        interior = np.zeros_like(ok)
        interior[1:-1, 1:-1] = ok[1:-1, 1:-1] & ok[1:-1, 2:] & ok[1:-1, :-2] & ok[2:, 1:-1] & ok[:-2, 1:-1]
        if interior.any():
            assert np.abs(div[interior]).max() < 1e-4, np.abs(div[interior]).max()
            checked += 1
    assert checked > 10


def test_half_tank_hydrostatic():
    """Synthetic config 3 at rest: the solve must return (nearly) zero velocities and a pressure
    increasing with depth; marker count is conserved (no sources, sinks unreachable)."""
    sim = ea.Simulation(256, 256).load_half_tank()
    n0 = sim.stats().n_markers
    for _ in range(2):
        sim.step()
    st = sim.stats()
    assert st.n_markers == n0
    p = sim.get(ea.F_PRESSURE)
    col = p[2:120, 128]
    assert (np.diff(col) <= 1e-9).all()       # deeper cells (lower y) carry more pressure


def test_render_through_device_matches_reference():
    g = load("block_frames.npz")
    r = load("block_render.npz")
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL).load_text(scenario_text(g))
    sim.step()
    for key in r.files:
        if key.startswith("f0_"):
            wx, wy = (int(t) for t in key.split("_")[1][1:].split("x"))
            assert sim.draw(wx, wy) == r[key].tobytes(), key


def test_error_conventions():
    sim = ea.Simulation(X, Y)
    with pytest.raises(ea.EulerError) as e:
        sim.sim_step()                      # before a scenario is loaded
    assert e.value.code == -5
    with pytest.raises(ea.EulerError) as e:
        sim.sim_init("/nonexistent/scenario.txt")
    assert e.value.code == -3 and "Could not load" in str(e.value)   # reference main.c:213
    with pytest.raises(ea.EulerError):
        ea.Simulation(4, 4)
    with pytest.raises(ea.EulerError) as e:
        ea.Simulation(20000, 20000)         # beyond the supported maximum: refused before anything is allocated
    assert e.value.code == -1 and "supported maximum" in str(e.value)


def test_cli_dump_matches_reference_frames(tmp_path):
    """The C `euler` front end (scenario file -> step -> render loop, reference main.c:1016-1042):
    frame k of `euler --dump` = the reference's draw_rows() bytes after k sim_step() calls."""
    import os
    import subprocess
    g = load("block_frames.npz")
    r = load("block_render.npz")
    scn = tmp_path / "block.txt"
    scn.write_text(scenario_text(g))
    exe = os.path.join(os.path.dirname(ea.LIB_PATH), "..", "bin", "euler")
    out = subprocess.run([exe, "--dump", "--frames", "2", "--window", "98x38", str(scn)], capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    frames = out.stdout.split(b"--- frame ")[1:]
    assert len(frames) == 3
    for k, key in ((1, "f0_w98x38"), (2, "f1_w98x38")):
        header, body = frames[k].split(b"\n", 1)
        n = int(header.split(b"(")[1].split()[0])
        assert body[:n] == r[key].tobytes(), key
    bad = subprocess.run([exe, "--dump", "/nonexistent.txt"], capture_output=True, timeout=60)
    assert bad.returncode == 1 and b"Could not load" in bad.stderr
    # --solver: the other preconditioners draw the same frames where the solves converge (block.txt at rest: the first frames do)
    for solver in ("tile", "two-level", "multilevel"):
        alt = subprocess.run([exe, "--dump", "--frames", "2", "--window", "98x38", "--solver", solver, "--max-iterations", "500", str(scn)], capture_output=True, timeout=120)
        assert alt.returncode == 0, alt.stderr.decode()
        assert alt.stdout == out.stdout, solver
    assert subprocess.run([exe, "--dump", "--solver", "nonsense", str(scn)], capture_output=True, timeout=60).returncode == 1
    # --solver tile-fp32 (euler_config.pcg_precision: solver vectors in float, the resident solver): the same frames on an upscaled grid as the double tile-local
    # solver while the block falls freely and lands (the picture is the cell grid)
    big = [exe, "--dump", "--frames", "30", "--size", "400x160", "--upscale", "--window", "98x38"]
    a = subprocess.run(big + ["--solver", "tile", str(scn)], capture_output=True, timeout=120)
    b = subprocess.run(big + ["--solver", "tile-fp32", str(scn)], capture_output=True, timeout=120)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr.decode(), b.stderr.decode())
    fa, fb = a.stdout.split(b"--- frame ")[1:], b.stdout.split(b"--- frame ")[1:]
    assert len(fa) == len(fb) == 31 and fa[:26] == fb[:26]


# ----------------------------------------------------------------------------- randomized marker stress
def _random_marker_state(X2, Y2, seed, n_markers, solid_frac, crowd=0):
    """A synthetic state that stresses the marker stages far beyond the shipped scenarios: scattered
    solid cells (many collisions after a cell crossing -> long dt chains, main.c:501), markers that
    end up in sink/solid cells (mass deletion + swap-with-last order, main.c:112), optional crowding
    of one cell beyond 255 markers (uint8 wrap, main.c:114)."""
    rng = np.random.default_rng(seed)
    solid = np.zeros((Y2, X2), np.uint8)
    solid[2:-2, 2:-2] = rng.random((Y2 - 4, X2 - 4)) < solid_frac
    sink = np.zeros((Y2, X2), np.uint8)
    sink[0, :] = sink[-1, :] = sink[:, 0] = sink[:, -1] = 1
    sink[3:-3, 3:-3] |= ((rng.random((Y2 - 6, X2 - 6)) < 0.02) & (solid[3:-3, 3:-3] == 0)).astype(np.uint8)
    free = np.argwhere((solid == 0) & (sink == 0) & (np.indices((Y2, X2))[0] > 1) & (np.indices((Y2, X2))[0] < Y2 - 2)
                       & (np.indices((Y2, X2))[1] > 1) & (np.indices((Y2, X2))[1] < X2 - 2))
    cells = free[rng.integers(0, len(free), n_markers)]
    if crowd:
        cells[:crowd] = free[len(free) // 2]
    m = np.empty((n_markers, 2), np.float32)
    m[:, 0] = cells[:, 1] + rng.random(n_markers, dtype=np.float32) * np.float32(0.998) + np.float32(0.001)
    m[:, 1] = cells[:, 0] + rng.random(n_markers, dtype=np.float32) * np.float32(0.998) + np.float32(0.001)
    u = (rng.standard_normal((Y2, X2)) * 3).astype(np.float32)
    v = (rng.standard_normal((Y2, X2)) * 3).astype(np.float32)
    u[:, -1] = 0
    v[-1, :] = 0
    return solid, sink, m, u, v


def _load_both(X2, Y2, solid, sink, m, u, v):
    o = Oracle(X2, Y2)
    o.solid[...] = solid; o.sink[...] = sink; o.u[...] = u; o.v[...] = v
    o.set_markers(m)
    o.lib.eo_refresh_marker_counts(o.ptr)            # establishes count / prev_count / deletes bad starts
    o.lib.eo_refresh_marker_counts(o.ptr)
    sim = ea.Simulation(X2, Y2, dot_mode=ea.DOT_SEQUENTIAL)
    for f, a in ((ea.F_SOLID, solid), (ea.F_SOURCE, np.zeros_like(solid)), (ea.F_SINK, sink), (ea.F_U, u), (ea.F_V, v),
                 (ea.F_COUNT, o.count), (ea.F_PREV_COUNT, o.prev_count)):
        sim.set(f, a)
    sim.set_markers(o.markers)
    return o, sim


@pytest.mark.parametrize("shape,seed,n,solid_frac", [((96, 72), 1, 6000, 0.25), ((200, 130), 2, 60000, 0.15),
                                                      ((333, 257), 3, 200000, 0.30), ((64, 64), 4, 3000, 0.45)])
@pytest.mark.parametrize("form", ["one_pass", "two_pass", "edited"])
def test_marker_stages_random_stress_bit_exact(shape, seed, n, solid_frac, form):
    """form: one_pass - the advection pass bins what it writes and pass B moves the counts of the markers it recomputes (round 6, the default: the dt chains
    of this state recompute most of the array); two_pass - rounds 1-5's separate binning pass (EULER_OPT_MARKERS_TWO_PASS); edited - the caller writes the
    marker array between the two stages, so what the advection pass binned must be dropped."""
    X2, Y2 = shape
    solid, sink, m, u, v = _random_marker_state(X2, Y2, seed, n, solid_frac)
    o, sim = _load_both(X2, Y2, solid, sink, m, u, v)
    if form == "two_pass":
        sim.set_option(ea.OPT_MARKERS_TWO_PASS, 1)
    events = 0
    for rep in range(3):
        dt = sim.timestep(0.1)
        assert dt == o.timestep(0.1)
        sim.stage(ea.STAGE_ADVECT_MARKERS, dt)
        o.lib.eo_advect_markers(o.ptr, np.float32(dt))
        assert_bits(sim.get(ea.F_MARKERS), o.markers, "advect rep %d" % rep)
        if form == "edited":      # the same markers, back to front: the counts must come from THIS array, the deletions in ITS order
            rev = np.ascontiguousarray(o.markers[::-1])
            o.set_markers(rev)
            sim.set_markers(rev)
        sim.stage(ea.STAGE_REFRESH_COUNTS)
        o.lib.eo_refresh_marker_counts(o.ptr)
        assert sim.stats().n_markers == o.n_markers
        assert_bits(sim.get(ea.F_MARKERS), o.markers, "compaction order rep %d" % rep)
        assert_bits(sim.get(ea.F_COUNT), o.count, "count rep %d" % rep)
        assert_bits(sim.get(ea.F_PREV_COUNT), o.prev_count, "prev_count rep %d" % rep)
    st = sim.stats()
    assert st.marker_dt_events > 5, "the stress state should shorten dt many times (got %d)" % st.marker_dt_events
    assert st.marker_multi_events == 0          # CFL bound: at most one shortening collision per marker
    assert st.n_markers < n                     # deletions happened


def test_markers_entering_tiles_without_water_are_counted():
    """The tile map (round 6): k_narrow_counts<true> reads the counters of a 64 x 64 tile only if the tile held water at the last refresh or a marker has ENTERED it since
    (k_advect_bin_a2 marks the tile of a marker's new cell when it differs from the old one's).  A block of markers that drifts out of its tile into three empty ones, a
    cell per substep: counts, previous counts and the marker array against the oracle after every stage - and the empty tiles do fill."""
    X2, Y2 = 333, 257
    rng = np.random.default_rng(11)
    solid = np.zeros((Y2, X2), np.uint8)
    sink = np.zeros((Y2, X2), np.uint8)
    sink[0, :] = sink[-1, :] = sink[:, 0] = sink[:, -1] = 1
    n = 12000
    m = np.empty((n, 2), np.float32)
    m[:, 0] = np.float32(66) + rng.random(n, dtype=np.float32) * np.float32(61.9)      # the tile of columns / rows 64 .. 127 only
    m[:, 1] = np.float32(66) + rng.random(n, dtype=np.float32) * np.float32(61.9)
    u = (3.0 + rng.random((Y2, X2))).astype(np.float32)
    v = (2.0 + rng.random((Y2, X2))).astype(np.float32)
    u[:, -1] = 0
    v[-1, :] = 0
    o, sim = _load_both(X2, Y2, solid, sink, m, u, v)
    assert o.count[64:128, 64:128].sum() == n and o.count.sum() == n
    for rep in range(45):
        dt = sim.timestep(0.1)
        assert dt == o.timestep(0.1)
        sim.stage(ea.STAGE_ADVECT_MARKERS, dt)
        o.lib.eo_advect_markers(o.ptr, np.float32(dt))
        sim.stage(ea.STAGE_REFRESH_COUNTS)
        o.lib.eo_refresh_marker_counts(o.ptr)
        assert_bits(sim.get(ea.F_COUNT), o.count, "count rep %d" % rep)
        assert_bits(sim.get(ea.F_PREV_COUNT), o.prev_count, "prev_count rep %d" % rep)
        if rep % 9 == 0:
            assert_bits(sim.get(ea.F_MARKERS), o.markers, "markers rep %d" % rep)
    c = sim.get(ea.F_COUNT)
    assert c[64:128, 128:192].sum() > 100 and c[128:192, 64:128].sum() > 100 and c[128:192, 128:192].sum() > 0, "the block should have drifted into the tiles beside it"
    assert sim.stats().n_markers == o.n_markers == n


def test_marker_count_wraps_like_uint8():
    """g_marker_count is uint8_t and simply wraps (main.c:96,114): 300 markers in one cell read 44."""
    X2, Y2 = 64, 64
    solid, sink, m, u, v = _random_marker_state(X2, Y2, 9, 2000, 0.0, crowd=300)
    u[...] = 0
    v[...] = 0
    o, sim = _load_both(X2, Y2, solid, sink, m, u, v)
    sim.stage(ea.STAGE_REFRESH_COUNTS)
    o.lib.eo_refresh_marker_counts(o.ptr)
    assert_bits(sim.get(ea.F_COUNT), o.count, "count")
    cy, cx = np.unravel_index(np.argmax(np.bincount((np.floor(m[:, 1]).astype(int) * X2 + np.floor(m[:, 0]).astype(int)), minlength=X2 * Y2)), (Y2, X2))
    true_count = int(((np.floor(m[:, 0]) == cx) & (np.floor(m[:, 1]) == cy)).sum())
    assert true_count >= 300 and int(sim.get(ea.F_COUNT)[cy, cx]) == true_count % 256


# ----------------------------------------------------------------------------- extension: velocity diffusion
@pytest.mark.parametrize("size,scn,nu,frames", [((130, 70), "block", 0.05, 34), ((192, 200), "waterfall", 0.2, 10)])
def test_velocity_diffusion_extension_bit_exact_vs_oracle(size, scn, nu, frames):
    """The reference is inviscid; the diffusion stage is this build's extension (config.viscosity), defined
    by oracle eo_diffuse.  viscosity = 0 leaves the stage out (every other test); > 0 must match the
    oracle's restatement bit for bit and must actually change the flow."""
    from trajectories import oracle_for
    _, sim = make_pair(size[0], size[1], scn, oracle=False, dot_mode=ea.DOT_SEQUENTIAL, viscosity=nu)
    o = oracle_for("diffusion_%dx%d_%s" % (size[0], size[1], scn))      # (the oracle with eo_sim.viscosity = nu; recorded: tests/trajectories.py)
    _, inviscid = make_pair(size[0], size[1], scn, oracle=False, dot_mode=ea.DOT_SEQUENTIAL)
    for f in range(frames):
        o.step(); sim.step(); inviscid.step()
        compare_all(o, sim, "nu=%g frame %d" % (nu, f))
    assert max(np.abs(sim.get(ea.F_U) - inviscid.get(ea.F_U)).max(), np.abs(sim.get(ea.F_V) - inviscid.get(ea.F_V)).max()) > 1e-4
    sim.close(); inviscid.close()


def test_baseline_size_1024_half_tank_bit_exact_vs_oracle():
    """BASELINE configs[1]'s grid size with the pressure solve fully engaged from the first frame (the
    dam break falls freely for ~22 frames, the half tank's free surface makes every solve run its
    100 iterations): 16 bands, 1 M cells, sequential dot mode - every field bit for bit after each of
    two frames."""
    from trajectories import oracle_for
    o = oracle_for("half_tank_1024")
    sim = ea.Simulation(1024, 1024, dot_mode=ea.DOT_SEQUENTIAL).load_half_tank()
    compare_all(o, sim, "init")
    for f in range(2):
        o.step()
        sim.step()
        st = sim.stats()
        assert st.last_substeps == o.c.last_substeps and st.last_pcg_iterations == o.c.last_pcg_iterations >= 100, f
        compare_all(o, sim, "1024 half tank frame %d" % f)
    sim.close()


def _compare_recorded(o, sim, what):
    """compare_all for a trajectory whose frames may hold the arrays' digests only now and then (`every`): the counters always, the arrays where they are on file"""
    from trajectories import same
    for f, attr in ((ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"), (ea.F_MARKERS, "markers"), (ea.F_U, "u"), (ea.F_V, "v"), (ea.F_PRECON, "precon"), (ea.F_PRESSURE, "p")):
        if o.has(attr):
            a = sim.get(f)
            same(a, o, attr, what + " " + attr)
            del a
    st = sim.stats()
    assert st.n_markers == o.n_markers, what
    assert st.rng_state == o.cur["rng_state"], what


def test_configs2_8192_half_tank_one_substep_bit_exact_vs_recorded_oracle():
    """BASELINE configs[2] AT ITS FULL SIZE against the pinned oracle, not through properties: 8192^2 half tank (67 M cells, 134 M markers), the reference's IC(0), the
    reference's order of every sum (EULER_DOT_SEQUENTIAL), one substep = calculate_timestep + the whole of main.c:853-892 with 100 PCG iterations - count grids, the marker
    array in order, u, v, g_precon and the pressure bit for bit.  The oracle's leg (19 minutes of one core, 7.5 GB) was run once in the build container and is on file as
    digests (tests/golden/trajectories.json: half_tank_8192_ic0_substep; tests/golden/make_trajectories.py regenerates it)."""
    from trajectories import oracle_for
    o = oracle_for("half_tank_8192_ic0_substep")
    sim = ea.Simulation(8192, 8192, dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0).load_half_tank()
    dt = sim.timestep(0.1)
    sim.substep(dt)
    o.substep(dt)
    assert sim.stats().last_pcg_iterations == o.c.last_pcg_iterations == 100
    _compare_recorded(o, sim, "8192^2 half tank, one substep")
    sim.close()


def test_roofline_mode_4096_half_tank_one_substep_bit_exact_vs_recorded_oracle():
    """The roofline mode's preconditioner (tile-local IC(0), tiles of 16 records) at 4096^2, sequential sums, one substep of 100 iterations against the oracle's
    restatement of it, bit for bit (trajectory half_tank_4096_tile_substep) - what the headline's kernels compute, pinned at a BASELINE size."""
    from trajectories import oracle_for
    o = oracle_for("half_tank_4096_tile_substep")
    sim = ea.Simulation(4096, 4096, dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0_TILE, tile_records=16).load_half_tank()
    dt = sim.timestep(0.1)
    sim.substep(dt)
    o.substep(dt)
    assert sim.stats().last_pcg_iterations == o.c.last_pcg_iterations == 100
    _compare_recorded(o, sim, "4096^2 half tank, tile-local mode, one substep")
    sim.close()


def test_configs4_4096_waterfall_sources_firing_bit_exact_vs_recorded_oracle():
    """BASELINE configs[4] at its full size: 4096^2 waterfall, the first six frames - 0.27 M source cells each appending a marker per substep (main.c:276-298: the RNG
    stream, the append order), the sink column, every solve into the cap - free-running against the recorded oracle, every array bit for bit after every frame."""
    from trajectories import oracle_for
    o = oracle_for("waterfall_4096")
    sim = ea.Simulation(4096, 4096, dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0).load_text(scenarios_waterfall(), upscale=True)
    n_before = sim.stats().n_markers
    for f in range(6):
        sim.step()
        o.step()
        st = sim.stats()
        assert st.last_substeps == o.c.last_substeps and st.last_pcg_iterations == o.c.last_pcg_iterations, f
        _compare_recorded(o, sim, "4096^2 waterfall frame %d" % f)
    assert sim.stats().n_markers > n_before + 100000 and not sim.stats().source_exhausted
    sim.close()


def test_dam_break_2048_into_the_capped_phase_bit_exact_vs_recorded_oracle():
    """configs[1] / [3]'s scenario at 2048^2 (5.2 M markers), 30 frames free-running: free fall with solves that converge in a few iterations or not at all, then six frames whose
    solves all run into the reference's cap (4 substeps of 100 iterations) - counters every frame, every array on every sixth frame, bit for bit.  (The trajectory on file
    holds 36 frames; the last six - five substeps each, all capped - add a minute of sequential-sum kernels and nothing new.)"""
    from euler_amd import scenarios
    from trajectories import oracle_for
    o = oracle_for("dam_break_2048")
    sim = ea.Simulation(2048, 2048, dot_mode=ea.DOT_SEQUENTIAL, precond=ea.PRECOND_IC0).load_text(scenarios.dam_break(), upscale=True)
    for f in range(30):
        sim.step()
        o.step()
        st = sim.stats()
        assert st.last_substeps == o.c.last_substeps and st.last_pcg_iterations == o.c.last_pcg_iterations, (f, st.last_substeps, st.last_pcg_iterations)
        _compare_recorded(o, sim, "2048^2 dam break frame %d" % f)
    assert sim.stats().last_pcg_iterations == 400
    sim.close()


def test_full_size_4096_properties_without_an_oracle():
    """Size-independent properties at a size the CPU oracle cannot reach in test time (4096^2 half tank,
    tree dots, one frame = 100 PCG iterations over 16.8 M cells, 33 M markers):
    the count grid is the histogram of the marker array; the residual the solver reports is the true
    residual b - A p recomputed on the host from the cell-mask encoding of A; p is clamped >= 0 and zero
    off the fluid; velocities vanish on solid faces."""
    N = 4096
    sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE).load_half_tank()
    sim.step()
    st = sim.stats()
    assert st.last_pcg_iterations == 100 * st.last_substeps
    count, solid = sim.get(ea.F_COUNT), sim.get(ea.F_SOLID)
    mk = sim.get(ea.F_MARKERS)
    assert len(mk) == st.n_markers
    cx, cy = np.floor(mk[:, 0]).astype(np.int64), np.floor(mk[:, 1]).astype(np.int64)
    hist = np.bincount(cy * N + cx, minlength=N * N).reshape(N, N)
    assert np.array_equal((hist & 255).astype(np.uint8), count)
    del mk, cx, cy, hist
    # true residual of the LAST solve: b - A p with A read off the mask byte (bit0 fluid, bits1-4 fluid at x+1, y+1, x-1, y-1, bits 5-7 a_diag)
    p, b, m = sim.get(ea.F_PRESSURE), sim.get(ea.F_PCG_B), sim.get(ea.F_CELLMASK)
    fluid = (m & 1) != 0
    assert np.array_equal(fluid, count > 0)
    assert (p[~fluid] == 0).all() and (p >= 0).all()
    u, v = sim.get(ea.F_U), sim.get(ea.F_V)
    su = (solid[:, :-1] | solid[:, 1:]) != 0
    sv = (solid[:-1, :] | solid[1:, :]) != 0
    assert (u[:, :-1][su] == 0).all() and (v[:-1, :][sv] == 0).all()
    sim.close()


def _true_residual(sim):
    """b - A p on the host from the cell-mask encoding of A (bit0 fluid, bits1-4 fluid at x+1, y+1, x-1, y-1, bits 5-7 a_diag)."""
    p, b, m = sim.get(ea.F_PRESSURE), sim.get(ea.F_PCG_B), sim.get(ea.F_CELLMASK)
    fl = (m & 1) != 0
    ap = (m >> 5).astype(np.float64) * p
    ap[:, :-1] -= np.where((m[:, :-1] & 2) != 0, p[:, 1:], 0.0)
    ap[:-1, :] -= np.where((m[:-1, :] & 4) != 0, p[1:, :], 0.0)
    ap[:, 1:] -= np.where((m[:, 1:] & 8) != 0, p[:, :-1], 0.0)
    ap[1:, :] -= np.where((m[1:, :] & 16) != 0, p[:-1, :], 0.0)
    return np.where(fl, b - ap, 0.0), fl


def mg_record(name):
    """tests/golden/mg_records.npz (make_mg_records.py): the oracle's restatement of the multilevel mode run to tolerance at BASELINE-sized grids -> (scalars, sampled p)"""
    import os
    from golden_util import GOLDEN
    with np.load(os.path.join(GOLDEN, "mg_records.npz")) as z:
        return z[name + ".scalars"], z[name + ".p"]


def mg_sample(p, n=64):
    Y, X = p.shape
    return p[np.ix_((np.arange(n) * (Y - 1)) // (n - 1), (np.arange(n) * (X - 1)) // (n - 1))]


@pytest.mark.parametrize("precond", [ea.PRECOND_IC0, ea.PRECOND_IC0_TILE, ea.PRECOND_IC0_TILE_MG])
def test_full_size_8192_half_tank_properties(precond):
    """BASELINE configs[2] at its full size (8192^2 half tank, 134 M markers), one substep, through the size-independent properties: the count grid is the histogram
    of the marker array (checked on a 2048-row stripe to stay inside the test box's memory), the residual vector the solver carries IS b - A p recomputed on the host
    (before the p >= 0 clamp can act: a tank at rest has positive pressures), p is 0 off the fluid, velocities vanish on solid faces.
    The reference's IC(0) and the tile-local mode run the reference's 100 iterations (in the tile-local mode this also shows that p += alpha s riding one pass
    behind - k_search_apply<.., PUPD>, the fmadds left to k_velocity_update_para / k_finish_p - loses no update).
    The MULTILEVEL mode (round 6: the mode whose solves converge had no test above 1536 x 1280) runs with the cap lifted: it reaches the reference's tolerance, the
    TRUE residual on the host is <= 1e-6, in <= 60 iterations - and iteration count, max |p| and the pressure on a 64 x 64 sample grid are those of the oracle's
    restatement (mg_build / mg_vcycle) recorded in the build container (tests/golden/mg_records.npz)."""
    N = 8192
    mg = precond == ea.PRECOND_IC0_TILE_MG
    sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=precond, max_iterations=4000 if mg else 100).load_half_tank()
    dt = sim.timestep(0.1)
    sim.substep(dt)
    st = sim.stats()
    res, fl = _true_residual(sim)
    r = sim.get(ea.F_PCG_R)
    scale = np.abs(sim.get(ea.F_PCG_B)).max()
    assert np.abs(np.where(fl, r, 0.0) - res).max() <= 1e-9 * scale
    assert abs(np.abs(r[fl]).max() - st.last_residual) <= 1e-12 * scale
    if mg:
        sc, ps = mg_record("half_tank_8192_mg")
        assert dt == np.float32(sc[3]) and st.n_markers == int(sc[4])
        assert st.last_residual <= 1e-6 and np.abs(res).max() <= 1e-6 + 1e-9 * scale
        assert st.last_pcg_iterations <= 60 and abs(st.last_pcg_iterations - sc[0]) <= 0.03 * sc[0] + 2, (st.last_pcg_iterations, sc[0])
    else:
        assert st.last_pcg_iterations == 100
    del res, r
    p = sim.get(ea.F_PRESSURE)
    assert (p[~fl] == 0).all() and (p >= 0).all()
    if mg:
        assert abs(np.abs(p).max() - sc[2]) <= 1e-6 * sc[2]
        assert np.abs(mg_sample(p) - ps).max() <= 1e-6 * sc[2]
    del p
    count, solid = sim.get(ea.F_COUNT), sim.get(ea.F_SOLID)
    assert np.array_equal(fl, count > 0)
    mk = sim.get(ea.F_MARKERS)
    assert len(mk) == st.n_markers
    cy = np.floor(mk[:, 1]).astype(np.int32)
    sel = (cy >= 1024) & (cy < 3072)
    cx = np.floor(mk[sel, 0]).astype(np.int64)
    hist = np.bincount((cy[sel].astype(np.int64) - 1024) * N + cx, minlength=2048 * N).reshape(2048, N)
    assert np.array_equal((hist & 255).astype(np.uint8), count[1024:3072])
    del mk, cy, cx, hist, sel
    u, v = sim.get(ea.F_U), sim.get(ea.F_V)
    assert (u[:, :-1][(solid[:, :-1] | solid[:, 1:]) != 0] == 0).all() and (v[:-1, :][(solid[:-1, :] | solid[1:, :]) != 0] == 0).all()
    sim.close()


def test_full_size_16384_dam_break_properties_both_modes():
    """BASELINE configs[3] at its full size on one GPU (16384^2 dam break: 83 M fluid cells, 334 M markers, ~61 GB of HBM with the ring of search directions),
    rolled into the expensive phase (every substep runs PCG into the iteration cap, main.c:735), then one frame in the tile-local
    mode and one in the reference's IC(0), each through the size-independent properties: the count grid is the histogram of the
    marker array (on a 2048-row stripe through the water), the residual the solver reports is max |r| of the vector it carries,
    and that vector IS b - A p recomputed on the host from the mask encoding of A - on every row of the system none of whose
    five pressures was clamped afterwards (main.c:773-779: the clamp acts after the solve and only ever writes an exact 0) -
    p >= 0 and 0 off the fluid, velocities vanish on solid faces, nothing is NaN."""
    from euler_amd import scenarios
    N = 16384
    sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE).load_text(scenarios.dam_break(), upscale=True)
    st = sim.stats()
    assert st.n_markers > 300e6
    for _ in range(60):
        sim.step()
        if sim.stats().last_pcg_iterations >= 100:
            break
    assert sim.stats().last_pcg_iterations >= 100
    solid = sim.get(ea.F_SOLID)
    su = (solid[:, :-1] | solid[:, 1:]) != 0
    sv = (solid[:-1, :] | solid[1:, :]) != 0
    del solid
    for precond in (ea.PRECOND_IC0_TILE, ea.PRECOND_IC0, ea.PRECOND_IC0_TILE_MG):
        sim.set_precond(precond)
        if precond == ea.PRECOND_IC0_TILE_MG:      # round 6: the multilevel mode at configs[3]'s full size, the cap lifted - every substep's solve reaches the reference's tolerance
            sim.set_solver(max_iterations=4000)
        sim.step()
        st = sim.stats()
        if precond == ea.PRECOND_IC0_TILE_MG:
            assert st.last_residual <= 1e-6 and 0 < st.last_pcg_iterations <= 60 * st.last_substeps, (st.last_pcg_iterations, st.last_substeps, st.last_residual)
        else:
            assert st.last_pcg_iterations >= 10 * st.last_substeps, (precond, st.last_pcg_iterations, st.last_substeps)      # the solves are engaged (some converge, some run into the cap)
        p, b, m = sim.get(ea.F_PRESSURE), sim.get(ea.F_PCG_B), sim.get(ea.F_CELLMASK)
        fl = (m & 1) != 0
        assert (p[~fl] == 0).all() and (p >= 0).all() and np.isfinite(p).all()
        # rows of the system that the clamp did not touch: the cell and its four neighbours all kept their solved pressure
        zero = fl & (p == 0)
        touched = zero.copy()
        touched[:, 1:] |= zero[:, :-1]; touched[:, :-1] |= zero[:, 1:]; touched[1:, :] |= zero[:-1, :]; touched[:-1, :] |= zero[1:, :]
        ok = fl & ~touched
        del zero, touched
        # (at this size the block of water is still falling when the solves start to run into the cap: nearly all of it is at p = 0 after
        # the clamp, and the rows the check can use are the ~2e5 along the walls and the floor)
        # (the multilevel leg SOLVES the system: the block is in free fall, its true pressure is zero but for rounding - nearly every cell is clamped and hardly a row is left;
        # the unconverged modes' pressures are the noise of 100 iterations, positive on millions of cells)
        assert ok.sum() > 1e4 or precond == ea.PRECOND_IC0_TILE_MG, (int(ok.sum()), int(fl.sum()))
        ap = (m >> 5).astype(np.float64) * p
        ap[:, :-1] -= np.where((m[:, :-1] & 2) != 0, p[:, 1:], 0.0)
        ap[:-1, :] -= np.where((m[:-1, :] & 4) != 0, p[1:, :], 0.0)
        ap[:, 1:] -= np.where((m[:, 1:] & 8) != 0, p[:, :-1], 0.0)
        ap[1:, :] -= np.where((m[1:, :] & 16) != 0, p[:-1, :], 0.0)
        scale = np.abs(b).max()
        np.subtract(b, ap, out=ap)                      # b - A p
        del p, b
        r = sim.get(ea.F_PCG_R)
        assert np.abs(r[fl]).max() == st.last_residual or abs(np.abs(r[fl]).max() - st.last_residual) <= 1e-12 * scale
        assert not ok.any() or np.abs((r - ap)[ok]).max() <= 1e-9 * scale, precond
        del r, ap, ok
        count = sim.get(ea.F_COUNT)
        assert np.array_equal(fl, count > 0)
        del m
        mk = sim.get(ea.F_MARKERS)
        assert len(mk) == st.n_markers and np.isfinite(mk).all()
        rows = np.nonzero(fl.any(axis=1))[0]
        y0 = int(rows[len(rows) // 2]) - 1024            # a 2048-row stripe through the middle of the water
        cy = np.floor(mk[:, 1]).astype(np.int32)
        sel = (cy >= y0) & (cy < y0 + 2048)
        cx = np.floor(mk[sel, 0]).astype(np.int64)
        hist = np.bincount((cy[sel].astype(np.int64) - y0) * N + cx, minlength=2048 * N).reshape(2048, N)
        assert np.array_equal((hist & 255).astype(np.uint8), count[y0:y0 + 2048])
        del mk, cy, cx, hist, sel, count, fl
        u, v = sim.get(ea.F_U), sim.get(ea.F_V)
        assert np.isfinite(u).all() and np.isfinite(v).all()
        assert (u[:, :-1][su] == 0).all() and (v[:-1, :][sv] == 0).all()
        del u, v
    sim.close()


@pytest.mark.parametrize("precond", [ea.PRECOND_IC0, ea.PRECOND_IC0_TILE, ea.PRECOND_IC0_TILE_MG])
def test_full_size_4096_waterfall_properties_with_sources_active(precond):
    """BASELINE configs[4]'s grid and scenario (4096^2 waterfall: ~0.27 M source cells, a sink column), a few frames with the
    sources running, both modes: every substep appends one marker per eligible source cell (the marker count grows by
    their number), the count grid stays the histogram of the marker array, markers appended in the last substep lie
    inside source cells, the RNG state moved, nothing is NaN, and the reported residual is the true one."""
    N = 4096
    mg = precond == ea.PRECOND_IC0_TILE_MG      # (round 6: the multilevel mode with the cap lifted - every solve to the reference's tolerance, at most 60 iterations per substep)
    sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=precond, max_iterations=4000 if mg else 100).load_text(scenarios_waterfall(), upscale=True)
    source = sim.get(ea.F_SOURCE)
    n_src = int((source != 0).sum())
    assert n_src > 200000
    st0 = sim.stats()
    for _ in range(3):
        sim.step()
        if mg:
            stf = sim.stats()
            assert stf.last_residual <= 1e-6 and stf.last_pcg_iterations <= 60 * stf.last_substeps, (stf.last_pcg_iterations, stf.last_substeps, stf.last_residual)
    st = sim.stats()
    if mg:      # the last solve's TRUE residual on the rows the clamp (main.c:773-779) left alone
        p, b, m = sim.get(ea.F_PRESSURE), sim.get(ea.F_PCG_B), sim.get(ea.F_CELLMASK)
        res, fl = _true_residual(sim)
        zero = fl & (p == 0)
        touched = zero.copy()
        touched[:, 1:] |= zero[:, :-1]; touched[:, :-1] |= zero[:, 1:]; touched[1:, :] |= zero[:-1, :]; touched[:-1, :] |= zero[1:, :]
        ok = fl & ~touched
        # (falling water: the solved pressure is rounding noise around zero and the clamp leaves hardly a row untouched - what rows there are must hold the residual)
        assert not ok.any() or np.abs(res[ok]).max() <= 1e-6 + 1e-9 * np.abs(b).max()
        assert (p[~fl] == 0).all() and (p >= 0).all()
        del p, b, m, res, fl, zero, touched, ok
    assert st.rng_state != st0.rng_state and not st.source_exhausted
    assert st0.n_markers < st.n_markers <= st0.n_markers + st.total_substeps * n_src
    mk = sim.get(ea.F_MARKERS)
    assert len(mk) == st.n_markers and np.isfinite(mk).all()
    cx, cy = np.floor(mk[:, 0]).astype(np.int64), np.floor(mk[:, 1]).astype(np.int64)
    count = sim.get(ea.F_COUNT)
    hist = np.bincount(cy * N + cx, minlength=N * N).reshape(N, N)
    # The count grid is the histogram of the marker array EXCEPT for what the last update_fluid_sources did: it counts the new
    # marker in its source cell (main.c:289-290) while x + randf() - float arithmetic at x ~ 4096, ulp 2^-11 - can round up to the
    # next cell's edge.  So: same total, and every difference sits in a source cell or right next to one.
    diff = hist.astype(np.int64) - count.astype(np.int64)
    assert int(hist.sum()) == st.n_markers == int(count.astype(np.int64).sum())
    near = source != 0
    near[1:, :] |= source[:-1, :] != 0; near[:, 1:] |= source[:, :-1] != 0
    assert (diff[~near] == 0).all() and np.abs(diff).max() <= 2 and (diff != 0).sum() < 1e-3 * n_src * st.last_substeps + 10
    # the markers appended by the last substep's update_fluid_sources are the array's tail (no deletion happens after it)
    tail = mk[-1000:]
    assert near[np.floor(tail[:, 1]).astype(np.int64), np.floor(tail[:, 0]).astype(np.int64)].all()
    del mk, cx, cy, hist
    u, v = sim.get(ea.F_U), sim.get(ea.F_V)
    assert np.isfinite(u).all() and np.isfinite(v).all()
    sim.close()


def scenarios_waterfall():
    from euler_amd import scenarios
    return scenarios.waterfall()


def test_reported_residual_is_the_true_residual_2048():
    """b - A p recomputed on the host (A from the cell-mask encoding) against the solver's own recursively
    updated residual, 2048^2 half tank, after exactly one solve (pressure is clamped only after the solve, so
    compare before the clamp can act: the half tank's pressures are positive)."""
    N = 2048
    sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE).load_half_tank()
    dt = sim.timestep(0.1)
    sim.substep(dt)
    p, b, m, r = sim.get(ea.F_PRESSURE), sim.get(ea.F_PCG_B), sim.get(ea.F_CELLMASK), sim.get(ea.F_PCG_R)
    fl = (m & 1) != 0
    diag = (m >> 5).astype(np.float64)
    ap = diag * p
    ap[:, :-1] -= np.where((m[:, :-1] & 2) != 0, p[:, 1:], 0.0)     # fluid at x+1
    ap[:-1, :] -= np.where((m[:-1, :] & 4) != 0, p[1:, :], 0.0)     # fluid at y+1
    ap[:, 1:] -= np.where((m[:, 1:] & 8) != 0, p[:, :-1], 0.0)      # fluid at x-1
    ap[1:, :] -= np.where((m[1:, :] & 16) != 0, p[:-1, :], 0.0)     # fluid at y-1
    true_r = np.where(fl, b - ap, 0.0)
    scale = np.abs(b).max()
    assert np.abs(true_r - r).max() <= 1e-9 * scale, (np.abs(true_r - r).max(), scale)
    assert abs(np.abs(true_r).max() - sim.stats().last_residual) <= 1e-9 * scale
    sim.close()


# ----------------------------------------------------------------------------- degenerate inputs
@pytest.mark.parametrize("size,text", [
    ((100, 40), ""),                                  # empty file: nothing but the sink ring, no markers at all
    ((100, 40), "X" * 98 + "\n" + "X" * 98 + "\n"),   # only walls
    ((100, 40), "?\n"),                               # one source cell and nothing else: the fluid appears from nothing
    ((8, 8), "00\n00\n"),                             # smallest useful grid, fluid against the sink ring
    ((70, 9), "0" * 68 + "\n" + "0" * 68 + "\n"),     # one short band, very flat
    ((9, 200), "\n".join(["0000000"] * 150) + "\n"),  # very narrow, 4 bands tall: T = 72, a band is mostly skew padding
])
def test_degenerate_scenarios_bit_exact_vs_oracle(size, text):
    """Zero markers, no fluid, one cell, grids smaller than a band or narrower than the skew: every stage must cope
    with empty launches and all-padding bands, and stay bit-identical to the oracle."""
    o = Oracle(size[0], size[1]).load_text(text, upscale=False)
    sim = ea.Simulation(size[0], size[1], dot_mode=ea.DOT_SEQUENTIAL).load_text(text, upscale=False)
    compare_all(o, sim, "init")
    for f in range(6):
        o.step()
        sim.step()
        st = sim.stats()
        assert st.n_markers == o.c.n_markers and st.last_substeps == o.c.last_substeps, f
        compare_all(o, sim, "%s frame %d" % (size, f))
    assert sim.draw(size[0] - 2, size[1] - 2) == o.render(size[0] - 2, size[1] - 2) if hasattr(o, "render") else True
    sim.close()


def test_jacobi_stand_in_preconditioner_converges_to_the_same_pressure():
    """EULER_PRECOND_JACOBI (roofline comparison only, not the reference's iterates) must still solve the same system:
    with enough iterations both preconditioners converge and the pressures agree to solver tolerance; the reported
    residual is the true residual."""
    N = 128
    sims = {}
    for name, pc in (("ic0", ea.PRECOND_IC0), ("jacobi", ea.PRECOND_JACOBI)):
        sim = ea.Simulation(N, N, dot_mode=ea.DOT_TREE, precond=pc, max_iterations=4000).load_half_tank()
        dt = sim.timestep(0.1)
        sim.substep(dt)
        st = sim.stats()
        assert st.last_residual <= 1e-6, (name, st.last_pcg_iterations, st.last_residual)
        sims[name] = (sim, st.last_pcg_iterations)
    p0, p1 = sims["ic0"][0].get(ea.F_PRESSURE), sims["jacobi"][0].get(ea.F_PRESSURE)
    assert sims["jacobi"][1] > sims["ic0"][1]                      # IC(0) is the better preconditioner
    assert np.abs(p0 - p1).max() <= 1e-4 * np.abs(p0).max()
    for sim, _ in sims.values():
        sim.close()


@pytest.mark.gpu
def test_four_cells_per_thread_stage_kernels_leave_the_same_bits():
    """extrapolate / zero_bounds (main.c:158-185, 822-832) run four cells per thread on large grids (k_extrapolate4 / k_zero_bounds4: word loads of the byte grids);
    the bit-exact tests above run grids below that threshold, i.e. the cell-per-thread kernels.  EULER_OPT_GRID4_MIN_CELLS = 0 selects the four-cell kernels on any grid
    whose width is a multiple of 4.  TWO HANDLES OF ONE PROCESS, one with the option and one without, step side by side: the same bits in every field and the marker
    array after every frame (default dot mode: the reference's sequential order)."""
    from euler_amd import scenarios
    for name, size, scn, frames, kw in (("waterfall", (320, 256), "waterfall", 40, {}), ("dam_break", (256, 384), "dam_break", 40, {}), ("rainbow", (192, 128), "waterfall", 25, dict(rainbow=True))):
        a = ea.Simulation(size[0], size[1], **kw).load_text(getattr(scenarios, scn)(), upscale=True)
        b = ea.Simulation(size[0], size[1], **kw).load_text(getattr(scenarios, scn)(), upscale=True)
        b.set_option(ea.OPT_GRID4_MIN_CELLS, 0)
        assert a.get_option(ea.OPT_GRID4_MIN_CELLS) == 1 << 22 and b.get_option(ea.OPT_GRID4_MIN_CELLS) == 0
        for f in range(frames):
            a.step(); b.step()
            for fld in (ea.F_U, ea.F_V, ea.F_PRESSURE, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_MARKERS):
                assert_bits(b.get(fld), a.get(fld), "%s frame %d field %d" % (name, f, fld))
        assert a.stats().total_pcg_iterations == b.stats().total_pcg_iterations > 100
        a.close(); b.close()



@pytest.mark.gpu
@pytest.mark.parametrize("precond", [ea.PRECOND_IC0, ea.PRECOND_IC0_TILE, ea.PRECOND_IC0_TILE_MG])
def test_round_6_stage_forms_leave_the_same_bits_as_rounds_1_to_5(precond):
    """Round 6 fused stages on whole-grid handles: the advection pass bins what it writes (k_advect_bin_a2, pass B moving the counts of recomputed markers), the
    assembly is one pass over parallelograms (k_build_system_para), the end of project() is one pass that finishes and clamps the pressure in LDS and leaves the
    maxima for the next timestep (k_velocity_update_para; the pressure is finished in memory only when asked for).  EULER_OPT_MARKERS_TWO_PASS / _BUILD_TWO_PASS /
    _VELOCITY_TWO_PASS / _NO_TILE_MAP select rounds 1-5's kernels.  Two handles of one process, one with the four options, step side by side: the same bits in every field, the
    pressure included, after every frame - in the parity mode (sequential dots), the tile-local mode (the resident solver on this size) and the multilevel mode."""
    from euler_amd import scenarios
    kw = dict(precond=precond, dot_mode=ea.DOT_SEQUENTIAL if precond == ea.PRECOND_IC0 else ea.DOT_TREE, max_iterations=4000 if precond == ea.PRECOND_IC0_TILE_MG else 100)
    for name, size, scn, frames in (("waterfall", (320, 256), "waterfall", 30), ("dam_break", (256, 384), "dam_break", 40)):
        a = ea.Simulation(size[0], size[1], **kw).load_text(getattr(scenarios, scn)(), upscale=True)
        b = ea.Simulation(size[0], size[1], **kw).load_text(getattr(scenarios, scn)(), upscale=True)
        for key in (ea.OPT_MARKERS_TWO_PASS, ea.OPT_BUILD_TWO_PASS, ea.OPT_VELOCITY_TWO_PASS, ea.OPT_NO_TILE_MAP):
            b.set_option(key, 1)
        for f in range(frames):
            # the tile map (tiles of 64 x 64 cells without water are left alone by the grid passes: what they would write is there already) must survive what makes it
            # stale: a caller's write into the state (dropped, rebuilt at the next refresh) and the option that stops its readers for a while
            if f == frames // 2:
                a.set(ea.F_V, a.get(ea.F_V))
            if f == frames // 2 + 4:
                a.set_option(ea.OPT_NO_TILE_MAP, 1)
            if f == frames // 2 + 6:
                a.set_option(ea.OPT_NO_TILE_MAP, 0)
            a.step(); b.step()
            assert a.stats().last_substeps == b.stats().last_substeps and a.stats().last_pcg_iterations == b.stats().last_pcg_iterations, (name, f)
            look = (ea.F_U, ea.F_V, ea.F_UTMP, ea.F_VTMP, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_MARKERS) + ((ea.F_PRESSURE,) if f % 3 == 0 else ())      # (most frames nobody looks at the pressure: the lazy path stays lazy)
            for fld in look:
                assert_bits(a.get(fld), b.get(fld), "%s frame %d field %d" % (name, f, fld))
        assert a.stats().total_pcg_iterations == b.stats().total_pcg_iterations > 100
        a.close(); b.close()


@pytest.mark.gpu
def test_round_6_stage_forms_at_configs2_size_over_frames():
    """The same comparison at BASELINE configs[2]'s size (8192^2 half tank, 134 M markers): the bit-exact tests against the recorded oracle run ONE substep from a fresh load,
    where the tile map is not valid yet - on grids this large its readers take 64 rows per workgroup (`rep`), a form the small grids never run.  Three frames, the solves
    capped at 20 iterations (the stages around them are what differs between the two handles), every field after every frame."""
    sims = []
    for old in (False, True):
        s = ea.Simulation(8192, 8192, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, max_iterations=20).load_half_tank()
        if old:
            for key in (ea.OPT_MARKERS_TWO_PASS, ea.OPT_BUILD_TWO_PASS, ea.OPT_VELOCITY_TWO_PASS, ea.OPT_NO_TILE_MAP):
                s.set_option(key, 1)
        sims.append(s)
    a, b = sims
    for f in range(3):
        a.step(); b.step()
        assert a.stats().last_substeps == b.stats().last_substeps and a.stats().last_pcg_iterations == b.stats().last_pcg_iterations > 0, f
        for fld in (ea.F_U, ea.F_V, ea.F_UTMP, ea.F_VTMP, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_PRESSURE, ea.F_MARKERS):
            x, y = a.get(fld), b.get(fld)
            assert x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8)), "frame %d field %d" % (f, fld)
            del x, y
    a.close(); b.close()


@pytest.mark.gpu
def test_the_lazily_finished_pressure_survives_an_option_that_forgets_the_ring():
    """k_velocity_update_para leaves the last p += alpha s and the clamp to whoever asks for EULER_F_PRESSURE (eu_pressure_current: from the search directions' ring of the
    last solve).  EULER_OPT_P_STEPS / _SA_RUN make the next solve set its ring up afresh: the pressure is finished BEFORE the ring is forgotten - two handles, one reads
    the pressure directly, one after such an option call: the same bits."""
    from euler_amd import scenarios
    sims = [ea.Simulation(256, 384, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0).load_text(scenarios.dam_break(), upscale=True) for _ in range(2)]
    for f in range(34):
        for s in sims:
            s.step()
    assert sims[0].stats().last_pcg_iterations > 50
    p0 = sims[0].get(ea.F_PRESSURE)
    sims[1].set_option(ea.OPT_P_STEPS, 4)
    p1 = sims[1].get(ea.F_PRESSURE)
    assert np.abs(p0).max() > 0
    assert_bits(p1, p0, "pressure behind euler_set_option(EULER_OPT_P_STEPS)")
    for s in sims:      # ... and both go on in step (p += alpha s four at a time leaves the same bits as eight)
        s.step()
    assert_bits(sims[1].get(ea.F_U), sims[0].get(ea.F_U), "u one frame later")
    for s in sims:
        s.close()
