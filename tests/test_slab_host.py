"""Host-side pieces of the row-slab decomposition (no GPU): the streaming marker seeding every rank runs over the ONE
sequential RNG stream (main.c:255-266) and the stacked-tank generator of the weak-scaling workload, through the C library."""
import ctypes as C

import numpy as np

import euler_amd as ea
from euler_amd.slab import slab_bands


def _lib():
    L = ea.load_library()
    L.euler_seed_markers_rows.restype = C.c_int
    L.euler_seed_markers_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p,
                                          C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.euler_half_tanks_grids.restype = C.c_int
    L.euler_half_tanks_grids.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def test_row_slabs_partition_the_marker_stream():
    """Every rank walks the whole stream and keeps its rows: together the ranks hold every marker of the single-GPU array
    exactly once, bit for bit, under its array index (key); the RNG ends in the same state everywhere."""
    L = _lib()
    X, Y = 96, 200
    rng = np.random.default_rng(5)
    fluid = (rng.random((Y, X)) < 0.4).astype(np.uint8)
    fluid[0, :] = fluid[-1, :] = 0; fluid[:, 0] = fluid[:, -1] = 0
    ref, st_ref = ea.seed_markers(fluid)
    nb = (Y + 63) // 64
    for world in (1, 2, 3):
        seen = np.zeros(len(ref), np.int32)
        for rank in range(world):
            lo, hi = slab_bands(nb, rank, world)
            row_lo, row_hi = 64 * lo, min(64 * hi, Y)
            st, n, k = C.c_uint64(0x9bd185c449534b91), C.c_uint64(0), C.c_uint64(0)
            assert L.euler_seed_markers_rows(fluid.ctypes.data, X, Y, row_lo, row_hi, C.byref(st), None, None, 0, C.byref(n), C.byref(k)) == 0
            assert n.value == len(ref)
            xy = np.zeros((k.value + 1, 2), np.float32)
            keys = np.zeros(k.value + 1, np.uint32)
            st = C.c_uint64(0x9bd185c449534b91)
            assert L.euler_seed_markers_rows(fluid.ctypes.data, X, Y, row_lo, row_hi, C.byref(st), xy.ctypes.data, keys.ctypes.data, k.value,
                                             C.byref(n), C.byref(k)) == 0
            assert st.value == st_ref
            xy, keys = xy[:k.value], keys[:k.value]
            assert np.array_equal(xy.view(np.uint32), ref[keys].view(np.uint32))
            assert ((np.floor(xy[:, 1]) >= row_lo) & (np.floor(xy[:, 1]) < row_hi)).all()
            seen[keys] += 1
        assert (seen == 1).all()


def test_stacked_half_tanks_are_copies_of_the_single_tank():
    L = _lib()
    X, H = 40, 64
    one = [np.zeros((H, X), np.uint8) for _ in range(4)]
    assert L.euler_half_tanks_grids(X, H, 1, *[a.ctypes.data for a in one]) == 0
    for tanks in (2, 3):
        g = [np.zeros((H * tanks, X), np.uint8) for _ in range(4)]
        assert L.euler_half_tanks_grids(X, H * tanks, tanks, *[a.ctypes.data for a in g]) == 0
        solid, source, sink, fluid = g
        for k in range(tanks):
            assert np.array_equal(fluid[k * H:(k + 1) * H], one[3])              # same water in every tank
            assert solid[k * H + 1:(k + 1) * H - 1, 1:-1][one[0][1:-1, 1:-1] > 0].all()   # the single tank's walls are there
        assert not source.any()
        ring = np.zeros_like(sink); ring[0, :] = ring[-1, :] = 1; ring[:, 0] = ring[:, -1] = 1
        assert np.array_equal(sink, ring)                                        # sinks only on the grid's border (main.c:244-252)
        assert (solid[[k * H for k in range(1, tanks)], 1:-1] == 1).all()        # closed where two tanks meet
    assert L.euler_half_tanks_grids(X, 100, 3, *[a.ctypes.data for a in one]) != 0  # rows must divide


def test_fluid_balanced_partition_tiles_the_grid():
    """bench.py's band ranges for a strong-scaling run: contiguous, in rank order, at least one band each, and as even in weight as a
    prefix split over whole bands gets (euler_config.slab_band_lo / hi take them; euler_set_comm* verifies the tiling again)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(11)
    for nb, world in ((4, 4), (16, 4), (256, 8), (100, 7), (9, 8)):
        for trial in range(20):
            settled = rng.integers(1, nb + 1)                       # water in the lowest `settled` bands, air above
            w = [1.02 if b < settled else 0.02 for b in range(nb)]
            if trial % 3 == 0:
                w = list(rng.random(nb) + 0.02)
            part = bench.balanced_partition(w, world)
            assert len(part) == world and part[0][0] == 0 and part[-1][1] == nb
            assert all(hi > lo for lo, hi in part) and all(part[r][1] == part[r + 1][0] for r in range(world - 1))
            loads = [sum(w[lo:hi]) for lo, hi in part]
            # no rank carries more than the ideal share plus one band's worth (a band cannot be split)
            assert max(loads) <= sum(w) / world + max(w) + 1e-9 or nb - world < 2, (nb, world, part, loads)
    even = bench.balanced_partition([1.0] * 256, 8)
    assert even == [(32 * r, 32 * r + 32) for r in range(8)]
