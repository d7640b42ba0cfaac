"""The tile-local IC(0) EXTENSION of the oracle (oracle/euler_oracle.h eo_tile_start; include/euler.h
EULER_PRECOND_IC0_TILE).  The reference has no such mode: these tests pin the restatement to the reference's IC(0) where
the two coincide (one band, one tile) and check the properties PCG needs of a preconditioner (symmetric, positive)."""
import numpy as np
import pytest

from golden_util import load, scenario_text
from oracle_lib import Oracle, oracle_lib


def test_tile_starts():
    L = oracle_lib()
    for w in (8, 16, 32):
        assert [t for t in range(100) if L.eo_tile_start(w, t)] == list(range(0, 100, w))
    assert [t for t in range(100) if L.eo_tile_start(0, t)] == [0]


def _system(X, Y, scn="block", frames=3):
    o = Oracle(X, Y).load_text(scenario_text(load(scn + "_frames.npz")), upscale=True)
    for _ in range(frames):
        o.step()
    o.utmp[...] = o.u; o.vtmp[...] = o.v
    o.lib.eo_build_system(o.ptr, np.float32(0.05), o.f32p(o.utmp), o.f32p(o.vtmp))
    return o


def _apply(o, units, r):
    o.c.tile_records = units
    o.precon[...] = 0
    o.r[...] = r
    o.lib.eo_apply_preconditioner(o.ptr, o.f64p(o.r), o.f64p(o.z))
    return o.z.copy()


def test_one_band_one_tile_is_the_reference_preconditioner():
    """Y <= 64 and one tile per band: no coupling is cut, so the tiled recurrences ARE main.c:586-626."""
    o = _system(130, 60)
    rng = np.random.default_rng(3)
    r = np.where(o.count > 0, rng.standard_normal(o.count.shape), 0.0)
    z_ref, pre_ref = _apply(o, 0, r), o.precon.copy()
    z_tile = _apply(o, 1 << 20, r)
    assert np.array_equal(z_tile, z_ref) and np.array_equal(o.precon, pre_ref)


@pytest.mark.parametrize("units", [8, 16, 32])
def test_tile_preconditioner_is_symmetric_positive(units):
    o = _system(300, 200)
    rng = np.random.default_rng(units)
    fluid = o.count > 0
    a = np.where(fluid, rng.standard_normal(fluid.shape), 0.0)
    b = np.where(fluid, rng.standard_normal(fluid.shape), 0.0)
    Ma, Mb = _apply(o, units, a), _apply(o, units, b)
    assert abs((Ma * b).sum() - (a * Mb).sum()) < 1e-10 * np.abs(Ma * b).sum()
    assert (Ma * a).sum() > 0 and (Mb * b).sum() > 0
    # it differs from the reference's preconditioner exactly when something is cut (3 bands here)
    assert not np.array_equal(Ma, _apply(o, 0, a))


def test_pcg_with_tile_preconditioner_reaches_the_same_pressure():
    """Tolerance parity where PCG converges: same tol, enough iterations -> |dp| <= 1e-5 max|p|, and the tile-local mode
    needs at most 1.6x the iterations of the reference's IC(0) on this system."""
    res = {}
    for units in (0, 16):
        o = Oracle(256, 256).load_half_tank()
        o.c.tile_records = units
        o.c.max_iterations = 3000
        o.step()
        res[units] = (o.p.copy(), int(o.c.total_pcg_iterations), o.c.last_residual)
    (p0, it0, r0), (p1, it1, r1) = res[0], res[16]
    assert r0 <= 1e-6 and r1 <= 1e-6
    assert np.abs(p1 - p0).max() <= 1e-5 * np.abs(p0).max()
    assert it0 < it1 <= 1.6 * it0, (it0, it1)
