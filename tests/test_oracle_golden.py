"""The CPU oracle (oracle/euler_oracle.c) against the golden fixtures generated from the compiled,
unmodified reference (tests/golden/make_golden.py).  Bit-exact, free-running and teacher-forced.
Runs without a GPU and without /root/reference."""
import ctypes as C

import numpy as np
import pytest

from golden_util import SCENARIOS, X, Y, bits_equal, load, scenario_text
from oracle_lib import Oracle, fnv1a64

FRAMES = {"waterfall": 460}


def _hashes(o):
    return [fnv1a64(o.u), fnv1a64(o.v), fnv1a64(o.count), fnv1a64(o.markers)]


@pytest.mark.parametrize("scn", SCENARIOS)
def test_init_matches_reference(scn):
    g = load(scn + "_frames.npz")
    o = Oracle(X, Y).load_text(scenario_text(g))
    for n in ("solid", "source", "sink"):
        assert bits_equal(getattr(o, n), g[n]), n
    assert bits_equal(o.count, g["init_count"])
    assert bits_equal(o.markers, g["init_markers"])       # order included
    assert int(o.c.rng_state) == int(g["init_rng"])


@pytest.mark.parametrize("scn", SCENARIOS)
def test_free_running_bit_exact(scn):
    g = load(scn + "_frames.npz")
    o = Oracle(X, Y).load_text(scenario_text(g))
    keep = set(int(f) for f in g["frames_full"])
    nframes = len(g["hashes"])
    for f in range(nframes):
        nsub, _ = o.step()
        assert nsub == int(g["n_substeps"][f]), (scn, f)
        assert _hashes(o) == [int(h) for h in g["hashes"][f]], (scn, f)
        assert o.n_markers == int(g["n_markers"][f])
        if f in keep:
            for n in ("u", "v", "count", "prev_count", "precon"):
                assert bits_equal(getattr(o, n), g["f%d_%s" % (f, n)]), (scn, f, n)
            assert bits_equal(o.markers, g["f%d_markers" % f])
            assert int(o.c.source_exhausted) == int(g["f%d_exhausted" % f])
            assert int(o.c.rng_state) == int(g["f%d_rng" % f])


def test_block_known_answer_counters():
    """SURVEY.md §8c probe: block.txt, 100 frames -> 359 substeps, 15 997 PCG iterations, 4488 markers."""
    g = load("block_frames.npz")
    o = Oracle(X, Y).load_text(scenario_text(g))
    for _ in range(100):
        o.step()
    assert int(o.c.total_substeps) == 359
    assert int(o.c.total_pcg_iterations) == 15997
    assert o.n_markers == 4488
    assert int((o.count > 0).sum()) == 1117


def test_waterfall_source_exhaustion_latch():
    """main.c:281,290: the source latches off when the marker array reaches 4*X*Y-1 (frame 450)."""
    g = load("waterfall_frames.npz")
    assert int(g["f449_exhausted"]) == 0 and int(g["f450_exhausted"]) == 1
    assert int(g["n_markers"].max()) == 4 * X * Y - 1


def _apply_stage(o, name, dt):
    L, p = o.lib, o.ptr
    f = C.c_float(dt)
    {
        "advect_markers": lambda: L.eo_advect_markers(p, f),
        "refresh_marker_counts": lambda: L.eo_refresh_marker_counts(p),
        "update_fluid_sources": lambda: L.eo_update_fluid_sources(p),
        "extrapolate_u": lambda: L.eo_extrapolate(p, o.f32p(o.u), 1),
        "extrapolate_v": lambda: L.eo_extrapolate(p, o.f32p(o.v), 2),
        "zero_bounds_u": lambda: L.eo_zero_bounds(p, o.f32p(o.u), 1),
        "zero_bounds_v": lambda: L.eo_zero_bounds(p, o.f32p(o.v), 2),
        "advect_u": lambda: L.eo_advect_u(p, o.f32p(o.u), o.f32p(o.v), f, o.f32p(o.utmp)),
        "advect_v": lambda: L.eo_advect_v(p, o.f32p(o.u), o.f32p(o.v), f, o.f32p(o.vtmp)),
        "apply_body_forces": lambda: L.eo_apply_body_forces(p, o.f32p(o.vtmp), f),
        "zero_bounds_utmp": lambda: L.eo_zero_bounds(p, o.f32p(o.utmp), 1),
        "zero_bounds_vtmp": lambda: L.eo_zero_bounds(p, o.f32p(o.vtmp), 2),
        "project": lambda: L.eo_project(p, f, o.f32p(o.utmp), o.f32p(o.vtmp), o.f32p(o.u), o.f32p(o.v)),
    }[name]()


def load_substep_state(o, g):
    """Teacher forcing: put the reference's state before the recorded substep into an Oracle."""
    for n in ("solid", "source", "sink"):
        getattr(o, n)[...] = g[n]
    for n in ("u", "v", "utmp", "vtmp", "count", "prev_count", "precon"):
        getattr(o, n)[...] = g["before_" + n]
    o.set_markers(g["before_markers"])
    o.c.rng_state = int(g["rng_before"])
    o.c.source_exhausted = int(g["exhausted_before"])


@pytest.mark.parametrize("scn", SCENARIOS)
def test_teacher_forced_stage_by_stage(scn):
    g = load(scn + "_substep.npz")
    o = Oracle(X, Y)
    load_substep_state(o, g)
    dt = float(g["dt"])
    assert np.float32(o.timestep(0.1)) >= np.float32(dt)  # dt = min(cfl, remaining frame time)
    expect = {n: g["before_" + n] for n in ("u", "v", "utmp", "vtmp", "count", "prev_count", "precon", "markers")}
    for i, name in enumerate(g["stage_names"]):
        name = str(name)
        _apply_stage(o, name, dt)
        for k in expect:
            key = "s%02d_%s" % (i, k)
            if key in g:
                expect[k] = g[key]
        for k, want in expect.items():
            got = o.markers if k == "markers" else getattr(o, k)
            assert bits_equal(got, want), (scn, i, name, k)
    assert int(o.c.rng_state) == int(g["rng_after"])
    assert int(o.c.source_exhausted) == int(g["exhausted_after"])


def test_filter_substep_has_dt_shortening_collision():
    """The filter fixture was cut at a substep where a marker collides after a cell crossing, so the
    reference's `dt -= t_prev` on its parameter (main.c:501,518) shortens dt for later markers.
    A per-marker dt (the 'fixed' behaviour) must NOT reproduce the fixture."""
    g = load("filter_substep.npz")
    o = Oracle(X, Y)
    load_substep_state(o, g)
    before = g["before_markers"].copy()
    want = g["s00_markers"]
    # advect each marker alone (fresh dt each): differs from the reference for later markers
    dt = float(g["dt"])
    alone = np.empty_like(before)
    for i in range(len(before)):
        o.set_markers(before[i:i + 1])
        o.lib.eo_advect_markers(o.ptr, C.c_float(dt))
        alone[i] = o.markers[0]
    assert not bits_equal(alone, want)
    first = int(np.argwhere((alone != want).any(1)).ravel()[0])
    assert bits_equal(alone[:first], want[:first])


@pytest.mark.parametrize("scn", SCENARIOS)
def test_render_rows_match_reference(scn):
    g = load(scn + "_frames.npz")
    r = load(scn + "_render.npz")
    o = Oracle(X, Y).load_text(scenario_text(g))
    for key in r.files:
        f, w = key.split("_")
        f = int(f[1:])
        wx, wy = (int(t) for t in w[1:].split("x"))
        for n in ("count",):
            o.count[...] = g["f%d_count" % f]
        assert o.render(wx, wy) == r[key].tobytes(), key


RAINBOW = ("block", "waterfall", "filter")


@pytest.mark.parametrize("scn", RAINBOW)
def test_rainbow_dye_bit_exact(scn):
    """--rainbow: colorize at init, extrapolate(P) x3, the source colour, advect_p x3 with the
    whole-array memcpy (main.c:187-201, 292-294, 859-863, 873-882) and the coloured draw_rows bytes
    (main.c:902-951) against the compiled reference run with g_rainbow_enabled."""
    g = load(scn + "_rainbow.npz")
    o = Oracle(X, Y, rainbow=True).load_text(scenario_text(load(scn + "_frames.npz")))
    for n, a in (("r", o.cr), ("g", o.cg), ("b", o.cb)):
        assert bits_equal(a, g["init_" + n]), n
    keep = set(int(f) for f in g["frames_full"])
    for f in range(len(g["hashes"])):
        o.step()
        assert [fnv1a64(o.cr), fnv1a64(o.cg), fnv1a64(o.cb), fnv1a64(o.u), fnv1a64(o.count)] == [int(h) for h in g["hashes"][f]], (scn, f)
        if f in keep:
            for n, a in (("r", o.cr), ("g", o.cg), ("b", o.cb)):
                assert bits_equal(a, g["f%d_%s" % (f, n)]), (scn, f, n)
            for (wx, wy) in ((98, 38), (40, 10)):
                assert o.render(wx, wy) == g["f%d_w%dx%d" % (f, wx, wy)].tobytes(), (scn, f, wx, wy)


def test_rainbow_does_not_disturb_the_flow():
    g, gr = load("block_frames.npz"), load("block_rainbow.npz")
    assert [int(h) for h in gr["hashes"][:, 3]] == [int(h) for h in g["hashes"][:, 0]]
