"""euler_amd — host-side binding of libeuler_hip.so, the MI355X-native simulation path of
cgmb/euler (reference main.c: sim_init :209, sim_step :843, draw :953).

This module is plumbing over the C ABI declared in include/euler.h (ctypes; no torch types cross
the boundary).  There is NO CPU implementation behind it: if the shared library is missing, or
no gfx950 device is present, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libeuler_hip.so")

# --- enums of include/euler.h ------------------------------------------------------------------
DOT_AUTO, DOT_SEQUENTIAL, DOT_TREE = 0, 1, 2
PRECOND_IC0, PRECOND_JACOBI, PRECOND_IC0_TILE, PRECOND_IC0_TILE2, PRECOND_IC0_TILE_MG = 0, 1, 2, 3, 4
(OPT_P_STEPS, OPT_TILE_STORE_AS, OPT_TILE_REVERSE, OPT_RESIDENT_CAP, OPT_GRID4_MIN_CELLS, OPT_SLAB_FUSION, OPT_RCCL_SMALL, OPT_RCCL_NO_EXCHANGE, OPT_MARKERS_ROWMAJOR,
 OPT_SA_RUN, OPT_NO_INTERIOR, OPT_BUILD_GATHER, OPT_RESIDENT_FORCE_TIMEOUT, OPT_MG_SPLIT_LEVEL, OPT_MG_SPLIT_ACTIVE, OPT_MARKERS_TWO_PASS, OPT_BUILD_TWO_PASS,
 OPT_VELOCITY_TWO_PASS, OPT_NO_TILE_MAP, OPT_PROFILE_STRIDE) = range(1, 21)      # include/euler.h EULER_OPT_*
SWEEP_AUTO, SWEEP_BAND, SWEEP_SIMPLE = 0, 1, 2
PCG_F64, PCG_F32 = 0, 1
RESIDENT_AUTO, RESIDENT_OFF = 0, 1
(F_U, F_V, F_UTMP, F_VTMP, F_SOLID, F_SOURCE, F_SINK, F_COUNT, F_PREV_COUNT, F_MARKERS, F_PRECON,
 F_PRESSURE, F_PCG_B, F_PCG_R, F_PCG_Z, F_PCG_S, F_PCG_Q, F_CELLMASK,
 F_DYE_R, F_DYE_G, F_DYE_B, F_DYE_RTMP, F_DYE_GTMP, F_DYE_BTMP, F_MARKER_KEYS) = range(25)
(STAGE_ADVECT_MARKERS, STAGE_REFRESH_COUNTS, STAGE_SOURCES, STAGE_EXTRAPOLATE, STAGE_ADVECT_VELOCITY,
 STAGE_PROJECT) = range(6)
(OP_BUILD_SYSTEM, OP_PRECON_FACTOR, OP_FORWARD_SOLVE, OP_BACKWARD_SOLVE, OP_APPLY_A, OP_DOT_ZR, OP_DOT_ZS,
 OP_INF_NORM_R, OP_UPDATE_PR, OP_UPDATE_SEARCH) = range(10)

_FIELD_DTYPE = {
    F_U: np.float32, F_V: np.float32, F_UTMP: np.float32, F_VTMP: np.float32,
    F_SOLID: np.uint8, F_SOURCE: np.uint8, F_SINK: np.uint8, F_COUNT: np.uint8, F_PREV_COUNT: np.uint8,
    F_MARKERS: np.float32, F_PRECON: np.float64, F_PRESSURE: np.float64, F_PCG_B: np.float64,
    F_PCG_R: np.float64, F_PCG_Z: np.float64, F_PCG_S: np.float64, F_PCG_Q: np.float64, F_CELLMASK: np.uint8,
    F_DYE_R: np.float32, F_DYE_G: np.float32, F_DYE_B: np.float32,
    F_DYE_RTMP: np.float32, F_DYE_GTMP: np.float32, F_DYE_BTMP: np.float32, F_MARKER_KEYS: np.uint32,
}


class EulerError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libeuler_hip error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("X", C.c_int32), ("Y", C.c_int32), ("device", C.c_int32),
        ("max_iterations", C.c_int32), ("tol", C.c_double), ("dot_mode", C.c_int32), ("precond", C.c_int32),
        ("sweep_mode", C.c_int32), ("max_substeps", C.c_int32), ("frame_time", C.c_float),
        ("viscosity", C.c_float), ("pcg_poll_interval", C.c_int32), ("rainbow", C.c_int32),
        ("precond_tile_records", C.c_int32), ("slab_rank", C.c_int32), ("slab_nranks", C.c_int32),
        ("slab_band_lo", C.c_int32), ("slab_band_hi", C.c_int32), ("pcg_precision", C.c_int32), ("resident", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64), ("total_substeps", C.c_uint64), ("total_pcg_iterations", C.c_uint64),
        ("last_substeps", C.c_int32), ("last_pcg_iterations", C.c_int32), ("last_residual", C.c_double),
        ("last_dt", C.c_float), ("n_markers", C.c_uint64), ("source_exhausted", C.c_int32),
        ("rng_state", C.c_uint64), ("marker_dt_events", C.c_uint64), ("marker_multi_events", C.c_uint64),
        ("fluid_cells", C.c_uint64),
    ]


_lib = None

EXPORTS = [
    "euler_config_default", "euler_create", "euler_destroy", "euler_last_error", "euler_abi_version",
    "euler_load_scenario_mem", "euler_load_scenario_file", "euler_load_half_tank", "euler_load_half_tanks", "euler_parse_scenario",
    "euler_seed_markers", "euler_step", "euler_timestep", "euler_substep", "euler_stage", "euler_pcg_op", "euler_set_precond", "euler_set_solver",
    "euler_get_field", "euler_set_field", "euler_set_markers", "euler_set_rng", "euler_get_stats",
    "euler_field_bytes", "euler_render", "euler_render_grids", "euler_render_grids_rgb", "euler_colorize", "euler_profile_enable",
    "euler_profile_class_count", "euler_profile_class_name", "euler_profile_get", "euler_profile_reset",
    "euler_measure_copy_bandwidth", "euler_measure_exchange", "euler_device_name", "euler_hbm_bytes", "euler_sweep_timeline", "euler_save_state", "euler_load_state", "euler_set_comm", "euler_set_stream", "euler_slab_info",
    "euler_rccl_unique_id", "euler_rccl_version", "euler_set_comm_rccl", "euler_comm_calls",
    "euler_p2p_export", "euler_p2p_connect", "euler_p2p_disconnect", "euler_p2p_calls", "euler_resident_info",
    "euler_set_option", "euler_get_option",
]


def load_library():
    """dlopen libeuler_hip.so (built in-tree by `make -C euler_amd/csrc`). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("EULER_HIP_LIB", LIB_PATH)   # development: an alternative build of the same library
    if not os.path.exists(path):
        raise ImportError(
            "euler_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C euler_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(path)
    vp, i32, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_float, C.c_double
    sig = {
        "euler_config_default": (C.c_int, [C.POINTER(Config)]),
        "euler_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
        "euler_destroy": (None, [vp]),
        "euler_last_error": (C.c_char_p, []),
        "euler_abi_version": (C.c_int, []),
        "euler_load_scenario_mem": (C.c_int, [vp, C.c_char_p, i32, i32]),
        "euler_load_scenario_file": (C.c_int, [vp, C.c_char_p, i32]),
        "euler_load_half_tank": (C.c_int, [vp]),
        "euler_load_half_tanks": (C.c_int, [vp, i32]),
        "euler_parse_scenario": (C.c_int, [C.c_char_p, i32, i32, i32, i32, vp, vp, vp, vp]),
        "euler_seed_markers": (C.c_int, [vp, i32, i32, C.POINTER(u64), vp, C.POINTER(u64)]),
        "euler_step": (C.c_int, [vp]),
        "euler_timestep": (C.c_int, [vp, f32, C.POINTER(f32)]),
        "euler_substep": (C.c_int, [vp, f32]),
        "euler_stage": (C.c_int, [vp, i32, f32]),
        "euler_pcg_op": (C.c_int, [vp, i32, f32, f64, C.POINTER(f64)]),
        "euler_set_precond": (C.c_int, [vp, i32, i32]),
        "euler_set_solver": (C.c_int, [vp, i32, f64]),
        "euler_get_field": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "euler_set_field": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "euler_set_markers": (C.c_int, [vp, vp, u64]),
        "euler_set_rng": (C.c_int, [vp, u64, i32]),
        "euler_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "euler_field_bytes": (C.c_size_t, [vp, i32]),
        "euler_render": (C.c_int, [vp, i32, i32, C.c_char_p, i32, C.POINTER(i32)]),
        "euler_render_grids": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, C.c_char_p, i32, C.POINTER(i32)]),
        "euler_render_grids_rgb": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_char_p, i32, C.POINTER(i32)]),
        "euler_colorize": (C.c_int, [vp]),
        "euler_profile_enable": (C.c_int, [vp, u64]),
        "euler_profile_class_count": (C.c_int, []),
        "euler_profile_class_name": (C.c_char_p, [i32]),
        "euler_profile_get": (C.c_int, [vp, i32, C.POINTER(f64), C.POINTER(u64)]),
        "euler_profile_reset": (C.c_int, [vp]),
        "euler_resident_info": (C.c_int, [vp, C.POINTER(u64)]),
        "euler_measure_copy_bandwidth": (C.c_int, [vp, C.c_size_t, i32, C.POINTER(f64)]),
        "euler_measure_exchange": (C.c_int, [vp, i32, i32, i32, C.POINTER(f64)]),
        "euler_device_name": (C.c_int, [vp, C.c_char_p, i32]),
        "euler_hbm_bytes": (u64, [vp]),
        "euler_sweep_timeline": (C.c_int, [vp, C.POINTER(C.c_uint64), i32]),
        "euler_save_state": (C.c_int, [vp, C.c_char_p]),
        "euler_load_state": (C.c_int, [vp, C.c_char_p]),
        "euler_set_comm": (C.c_int, [vp, vp, i32]),                 # euler_amd/slab.py passes a CommOps struct
        "euler_set_stream": (C.c_int, [vp, vp]),
        "euler_slab_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "euler_rccl_unique_id": (C.c_int, [vp, i32]),
        "euler_rccl_version": (C.c_int, []),
        "euler_set_comm_rccl": (C.c_int, [vp, vp, i32, i32, i32, i32]),
        "euler_comm_calls": (C.c_int, [vp, C.POINTER(u64)]),
        "euler_p2p_export": (C.c_int, [vp, vp, i32]),
        "euler_p2p_connect": (C.c_int, [vp, vp, i32]),
        "euler_p2p_disconnect": (C.c_int, [vp]),
        "euler_p2p_calls": (C.c_int, [vp, C.POINTER(u64)]),
        "euler_set_option": (C.c_int, [vp, i32, C.c_int64]),
        "euler_get_option": (C.c_int, [vp, i32, C.POINTER(C.c_int64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = a symbol include/euler.h declares is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise EulerError(rc, load_library().euler_last_error().decode(errors="replace"))


# --- host-only helpers (no GPU) ------------------------------------------------------------------
def parse_scenario(text, X, Y, upscale=False):
    """Scenario text -> (solid, source, sink, fluid) uint8 [Y][X] (reference main.c:217-252)."""
    if isinstance(text, str):
        text = text.encode()
    out = [np.zeros((Y, X), np.uint8) for _ in range(4)]
    _check(load_library().euler_parse_scenario(text, len(text), X, Y, int(upscale), *[a.ctypes.data for a in out]))
    return tuple(out)


def seed_markers(fluid, rng_state=0x9bd185c449534b91):
    """Four jittered markers per fluid cell (reference main.c:255-266). Returns (markers[n,2], rng_state)."""
    fluid = np.ascontiguousarray(fluid, np.uint8)
    Y, X = fluid.shape
    m = np.zeros((4 * X * Y, 2), np.float32)
    st, n = C.c_uint64(rng_state), C.c_uint64(0)
    _check(load_library().euler_seed_markers(fluid.ctypes.data, X, Y, C.byref(st), m.ctypes.data, C.byref(n)))
    return m[: n.value].copy(), st.value


def render_grids(solid, sink, count, wx, wy, rgb=None):
    """draw_rows() over host grids (reference main.c:914-951); rgb = (r, g, b) float grids for --rainbow."""
    Y, X = count.shape
    L = load_library()
    n = C.c_int32(0)
    a = [np.ascontiguousarray(g, np.uint8) for g in (solid, sink, count)]
    ptrs = [g.ctypes.data for g in a]
    fn = L.euler_render_grids
    if rgb is not None:
        a += [np.ascontiguousarray(g, np.float32) for g in rgb]
        ptrs = [g.ctypes.data for g in a]
        fn = L.euler_render_grids_rgb
    _check(fn(*ptrs, X, Y, wx, wy, None, 0, C.byref(n)))
    buf = C.create_string_buffer(max(n.value, 1))
    _check(fn(*ptrs, X, Y, wx, wy, buf, n.value, C.byref(n)))
    return buf.raw[: n.value]


SNAPSHOT_F32 = ("u", "v", "utmp", "vtmp")
SNAPSHOT_U8 = ("solid", "source", "sink", "count", "prev_count")
SNAPSHOT_DYE = ("dye_r", "dye_g", "dye_b", "dye_rtmp", "dye_gtmp", "dye_btmp")   # version 2 (euler_config.rainbow)


def _fnv1a64(data, h=14695981039346656037):
    # vectorised FNV-1a is not possible (sequential dependence); snapshots of test size only
    for b in memoryview(data).cast("B"):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def read_snapshot(path, verify=True):
    """Parse a state snapshot written by euler_save_state (layout: include/euler.h) into a dict of
    numpy arrays + scalars.  verify=True recomputes the FNV-1a-64 checksum (pure Python: slow beyond a
    few MB)."""
    import struct
    raw = open(path, "rb").read()
    magic, version, X, Y, _, n, rng, exhausted, _, frames, substeps, iters = struct.unpack_from("<8sIiiIQQiiQQQ", raw, 0)
    if magic != b"EULERSNP" or version not in (1, 2):
        raise ValueError("%s is not an euler state snapshot (version 1 or 2)" % path)
    out = {"X": X, "Y": Y, "n_markers": n, "rng_state": rng, "source_exhausted": exhausted, "frames": frames,
           "total_substeps": substeps, "total_pcg_iterations": iters}
    off, Cn = 72, X * Y
    for name in SNAPSHOT_F32:
        out[name] = np.frombuffer(raw, np.float32, Cn, off).reshape(Y, X).copy(); off += 4 * Cn
    for name in SNAPSHOT_U8:
        out[name] = np.frombuffer(raw, np.uint8, Cn, off).reshape(Y, X).copy(); off += Cn
    out["precon"] = np.frombuffer(raw, np.float64, Cn, off).reshape(Y, X).copy(); off += 8 * Cn
    if version == 2:
        for name in SNAPSHOT_DYE:
            out[name] = np.frombuffer(raw, np.float32, Cn, off).reshape(Y, X).copy(); off += 4 * Cn
    out["markers"] = np.frombuffer(raw, np.float32, 2 * n, off).reshape(n, 2).copy(); off += 8 * n
    (want,) = struct.unpack_from("<Q", raw, off)
    if len(raw) != off + 8:
        raise ValueError("%s: %d trailing bytes" % (path, len(raw) - off - 8))
    if verify and _fnv1a64(raw[:off]) != want:
        raise ValueError("%s: checksum mismatch" % path)
    return out


def write_snapshot(path, st):
    """Inverse of read_snapshot (e.g. to turn a golden fixture into a resumable state)."""
    import struct
    Y, X = st["u"].shape
    mk = np.ascontiguousarray(st["markers"], np.float32).reshape(-1, 2)
    dye = "dye_r" in st
    body = struct.pack("<8sIiiIQQiiQQQ", b"EULERSNP", 2 if dye else 1, X, Y, 0, len(mk), int(st["rng_state"]), int(st.get("source_exhausted", 0)), 0,
                       int(st.get("frames", 0)), int(st.get("total_substeps", 0)), int(st.get("total_pcg_iterations", 0)))
    body += b"".join(np.ascontiguousarray(st[k], np.float32).tobytes() for k in SNAPSHOT_F32)
    body += b"".join(np.ascontiguousarray(st[k], np.uint8).tobytes() for k in SNAPSHOT_U8)
    body += np.ascontiguousarray(st["precon"], np.float64).tobytes()
    if dye:
        body += b"".join(np.ascontiguousarray(st[k], np.float32).tobytes() for k in SNAPSHOT_DYE)
    body += mk.tobytes()
    with open(path, "wb") as f:
        f.write(body + struct.pack("<Q", _fnv1a64(body)))


def profile_class_names():
    L = load_library()
    return [L.euler_profile_class_name(i).decode() for i in range(L.euler_profile_class_count())]


class Simulation:
    """One simulation on one MI355X.  Mirrors the reference's surface: sim_init / sim_step / draw,
    plus state access for parity tests."""

    def __init__(self, X=100, Y=40, device=0, dot_mode=DOT_AUTO, precond=PRECOND_IC0, sweep_mode=SWEEP_AUTO,
                 max_iterations=100, tol=None, pcg_poll_interval=8, viscosity=0.0, rainbow=False, tile_records=0, slab=None,
                 pcg_precision=0, resident=0):
        self.L = load_library()
        cfg = Config()
        _check(self.L.euler_config_default(C.byref(cfg)))
        cfg.X, cfg.Y, cfg.device = X, Y, device
        cfg.dot_mode, cfg.precond, cfg.sweep_mode = dot_mode, precond, sweep_mode
        cfg.max_iterations = max_iterations
        if tol is not None:
            cfg.tol = tol
        cfg.pcg_poll_interval = pcg_poll_interval
        cfg.viscosity = viscosity            # extension: 0 = the inviscid reference
        cfg.rainbow = int(rainbow)           # args_t.rainbow (main.c:54): carry and advect the dye fields
        cfg.precond_tile_records = tile_records  # PRECOND_IC0_TILE: records per tile, 8 / 16 / 32 (0 = default 16)
        if slab is not None:                 # (rank, nranks[, band_lo, band_hi]): row slabs for every stage, this process holds one slab only
            cfg.slab_rank, cfg.slab_nranks = slab[0], slab[1]
            if len(slab) == 4:               # an explicit (fluid-balanced) partition instead of the even split
                cfg.slab_band_lo, cfg.slab_band_hi = slab[2], slab[3]
        self.slab = slab if slab is not None and slab[1] >= 1 else None
        cfg.pcg_precision = pcg_precision    # PCG_F32: solver vectors in float (BASELINE configs[1]'s "fp32"; resident solver only)
        cfg.resident = resident              # RESIDENT_OFF: never the one-launch resident solver (k_resident.hip)
        self.cfg = cfg
        self.X, self.Y = X, Y
        self.h = C.c_void_p()
        _check(self.L.euler_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.L.euler_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- reference surface
    def sim_init(self, scenario_file, upscale=False):
        _check(self.L.euler_load_scenario_file(self.h, os.fsencode(scenario_file), int(upscale)))
        return self

    def load_text(self, text, upscale=False):
        if isinstance(text, str):
            text = text.encode()
        _check(self.L.euler_load_scenario_mem(self.h, text, len(text), int(upscale)))
        return self

    def load_half_tank(self, tanks=1):
        _check(self.L.euler_load_half_tanks(self.h, tanks))
        return self

    def sim_step(self):
        _check(self.L.euler_step(self.h))

    step = sim_step

    def draw(self, wx, wy):
        n = C.c_int32(0)
        _check(self.L.euler_render(self.h, wx, wy, None, 0, C.byref(n)))
        buf = C.create_string_buffer(max(n.value, 1))
        _check(self.L.euler_render(self.h, wx, wy, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    def colorize(self):
        """The reference's 'r' key (main.c:970-973): colour the current fluid afresh."""
        _check(self.L.euler_colorize(self.h))

    # --- finer control
    def timestep(self, frame_time_left=0.1):
        dt = C.c_float(0)
        _check(self.L.euler_timestep(self.h, frame_time_left, C.byref(dt)))
        return dt.value

    def substep(self, dt):
        _check(self.L.euler_substep(self.h, dt))

    def stage(self, stage, dt=0.0):
        _check(self.L.euler_stage(self.h, stage, dt))

    def set_precond(self, precond, tile_records=0):
        _check(self.L.euler_set_precond(self.h, precond, tile_records))

    def set_solver(self, max_iterations=0, tol=-1.0):
        """iteration budget / tolerance of the following solves (<= 0 / < 0: unchanged)"""
        _check(self.L.euler_set_solver(self.h, max_iterations, tol))

    def pcg_op(self, op, dt=0.0, scalar=0.0):
        out = C.c_double(0)
        _check(self.L.euler_pcg_op(self.h, op, dt, scalar, C.byref(out)))
        return out.value

    # --- state
    def get(self, field):
        nbytes = self.L.euler_field_bytes(self.h, field)
        dt = np.dtype(_FIELD_DTYPE[field])
        a = np.empty(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(self.L.euler_get_field(self.h, field, a.ctypes.data, nbytes))
        elif field not in (F_MARKERS, F_MARKER_KEYS):      # an unknown or unavailable field: let the library say why
            _check(self.L.euler_get_field(self.h, field, np.empty(8, np.uint8).ctypes.data, 0) or -1)
        if field == F_MARKERS:
            return a.reshape(-1, 2)
        if field == F_MARKER_KEYS:
            return a
        return a.reshape(-1, self.X)        # a row-slab handle returns its own rows (euler_slab_info)

    def set(self, field, arr):
        a = np.ascontiguousarray(arr, _FIELD_DTYPE[field])
        _check(self.L.euler_set_field(self.h, field, a.ctypes.data, a.nbytes))

    def slab_rows(self):
        """[row_lo, row_hi) this handle owns (the whole grid without slabs)."""
        lo, hi, nb = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self.L.euler_slab_info(self.h, C.byref(lo), C.byref(hi), C.byref(nb)))
        return (64 * lo.value, min(64 * hi.value, self.Y)) if self.slab else (0, self.Y)

    def set_markers(self, m):
        a = np.ascontiguousarray(m, np.float32).reshape(-1, 2)
        _check(self.L.euler_set_markers(self.h, a.ctypes.data, len(a)))

    def set_rng(self, state, exhausted=0):
        _check(self.L.euler_set_rng(self.h, int(state), int(exhausted)))

    def set_option(self, key, value):
        """include/euler.h EULER_OPT_*: a per-handle option (what used to be an EULER_* environment variable)"""
        _check(self.L.euler_set_option(self.h, int(key), C.c_int64(int(value))))
        return self

    def get_option(self, key):
        v = C.c_int64(0)
        _check(self.L.euler_get_option(self.h, int(key), C.byref(v)))
        return int(v.value)

    def resident_info(self):
        """(eligible, solves run by the resident solver, solves that fell back to the multi-kernel path)"""
        out = (C.c_uint64 * 3)()
        _check(self.L.euler_resident_info(self.h, out))
        return bool(out[0]), int(out[1]), int(out[2])

    def stats(self):
        s = Stats()
        _check(self.L.euler_get_stats(self.h, C.byref(s)))
        return s

    # --- measurement
    def profile_enable(self, classes):
        names = profile_class_names()
        mask = 0
        for c in classes:
            mask |= 1 << (names.index(c) if isinstance(c, str) else c)
        _check(self.L.euler_profile_enable(self.h, mask))

    def profile_reset(self):
        _check(self.L.euler_profile_reset(self.h))

    def profile(self):
        out = {}
        for i, n in enumerate(profile_class_names()):
            ms, k = C.c_double(0), C.c_uint64(0)
            _check(self.L.euler_profile_get(self.h, i, C.byref(ms), C.byref(k)))
            if k.value:
                out[n] = (ms.value, k.value)
        return out

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        g = C.c_double(0)
        _check(self.L.euler_measure_copy_bandwidth(self.h, nbytes, reps, C.byref(g)))
        return g.value

    def exchange_latency(self, reps=200, row_doubles=0, nsmall=2):
        """collective: microseconds per exchange point of a distributed PCG iteration over the installed communicator"""
        g = C.c_double(0)
        _check(self.L.euler_measure_exchange(self.h, reps, row_doubles, nsmall, C.byref(g)))
        return g.value

    def save_state(self, path):
        """Checkpoint (include/euler.h "state snapshots"): everything sim_step() depends on."""
        _check(self.L.euler_save_state(self.h, os.fsencode(path)))

    def load_state(self, path):
        _check(self.L.euler_load_state(self.h, os.fsencode(path)))
        return self

    def sweep_timeline(self, raw=False):
        """[(entry_us, first_ready_us, exit_us, blocks, stalled_blocks)] per band of the last sweep launch,
        times relative to the first band's entry.  raw=True: additionally the 4 development words as integers
        (stall counts per helper, or 100 MHz time stamps in an SW_TRACE_HANDOFF build) and the origin tick."""
        nb = (self.Y + 63) // 64
        buf = (C.c_uint64 * (8 * nb))()
        n = self.L.euler_sweep_timeline(self.h, buf, nb)
        if n < 0:
            _check(n)
        rows = [tuple(buf[8 * i + k] for k in range(8)) for i in range(n)]
        t0 = min(r[0] for r in rows) if rows else 0
        out = []
        for r in rows:
            row = ((r[0] - t0) / 100.0, (r[1] - t0) / 100.0, (r[2] - t0) / 100.0, r[3] >> 32, r[3] & 0xffffffff)
            if raw:
                row += tuple(int(x) for x in r[4:8]) + (int(t0),)
            out.append(row)
        return out

    def hbm_bytes(self):
        """device memory this handle allocated (a row-slab handle: its slab only)"""
        return int(self.L.euler_hbm_bytes(self.h))

    def device_name(self):
        buf = C.create_string_buffer(256)
        _check(self.L.euler_device_name(self.h, buf, 256))
        return buf.value.decode()
