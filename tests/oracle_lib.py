"""ctypes bindings for the TEST ORACLE (oracle/liboracle.so) and, when it was built in the
authoring container, the compiled unmodified reference (oracle/_ref/libeuler_ref.so).

Test infrastructure only: nothing under euler_amd/ imports this module.
"""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P, U, V = 0, 1, 2


class Vec2f(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class EoSim(C.Structure):
    _fields_ = [
        ("X", C.c_int), ("Y", C.c_int),
        ("u", C.POINTER(C.c_float)), ("v", C.POINTER(C.c_float)),
        ("utmp", C.POINTER(C.c_float)), ("vtmp", C.POINTER(C.c_float)),
        ("solid", C.POINTER(C.c_uint8)), ("source", C.POINTER(C.c_uint8)), ("sink", C.POINTER(C.c_uint8)),
        ("count", C.POINTER(C.c_uint8)), ("prev_count", C.POINTER(C.c_uint8)),
        ("markers", C.POINTER(Vec2f)),
        ("n_markers", C.c_size_t), ("max_markers", C.c_size_t),
        ("source_exhausted", C.c_int),
        ("rng_state", C.c_uint64),
        ("a_diag", C.POINTER(C.c_int8)),
        ("precon", C.POINTER(C.c_double)), ("q", C.POINTER(C.c_double)),
        ("b", C.POINTER(C.c_double)), ("p", C.POINTER(C.c_double)),
        ("r", C.POINTER(C.c_double)), ("z", C.POINTER(C.c_double)), ("s", C.POINTER(C.c_double)),
        ("max_iterations", C.c_int), ("tol", C.c_double), ("viscosity", C.c_float),
        ("total_substeps", C.c_uint64), ("total_pcg_iterations", C.c_uint64),
        ("last_substeps", C.c_int), ("last_pcg_iterations", C.c_int),
        ("last_residual", C.c_double), ("last_dt", C.c_float), ("frame_count", C.c_uint32),
        ("rainbow", C.c_int),
        ("cr", C.POINTER(C.c_float)), ("cg", C.POINTER(C.c_float)), ("cb", C.POINTER(C.c_float)),
        ("crtmp", C.POINTER(C.c_float)), ("cgtmp", C.POINTER(C.c_float)), ("cbtmp", C.POINTER(C.c_float)),
        ("tile_records", C.c_int),
        ("coarse_m", C.c_int), ("coarse_n", C.c_int), ("coarse_nx", C.c_int), ("coarse_chol", C.POINTER(C.c_double)),
        ("coarse_mg", C.c_int), ("mg", C.c_void_p),
        ("coarse_npinned", C.c_int), ("coarse_pinned", C.c_int * 16), ("coarse_null", C.POINTER(C.c_double)),
        ("pcg_f32", C.c_int), ("coarse_bw", C.c_int), ("mg_theta", C.c_double),
    ]


def build_oracle(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    path = os.path.join(ORACLE_DIR, name)
    src = os.path.join(ORACLE_DIR, "euler_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name], stdout=subprocess.DEVNULL)
    return path


_LIBS = {}


def oracle_lib(fast=False, lib_path=None):
    """lib_path: load this exact build of oracle/euler_oracle.c (bench.py's cpu_baseline leg compiles
    one for the host it runs on); otherwise the in-tree liboracle.so / liboracle_fast.so."""
    key = lib_path or fast
    if key in _LIBS:
        return _LIBS[key]
    lib = C.CDLL(lib_path or build_oracle(fast))
    sp = C.POINTER(EoSim)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    lib.eo_create.restype = sp
    lib.eo_create.argtypes = [C.c_int, C.c_int]
    lib.eo_destroy.argtypes = [sp]
    lib.eo_load_scenario_mem.argtypes = [sp, C.c_char_p, C.c_int, C.c_int]
    lib.eo_load_scenario_file.argtypes = [sp, C.c_char_p, C.c_int]
    lib.eo_load_half_tank.argtypes = [sp]
    lib.eo_step.argtypes = [sp]
    lib.eo_substep.argtypes = [sp, C.c_float]
    lib.eo_substep.restype = C.c_int
    lib.eo_calculate_timestep.argtypes = [sp, C.c_float]
    lib.eo_calculate_timestep.restype = C.c_float
    lib.eo_advect_markers.argtypes = [sp, C.c_float]
    lib.eo_refresh_marker_counts.argtypes = [sp]
    lib.eo_update_fluid_sources.argtypes = [sp]
    lib.eo_extrapolate.argtypes = [sp, fp, C.c_int]
    lib.eo_zero_bounds.argtypes = [sp, fp, C.c_int]
    lib.eo_advect_u.argtypes = [sp, fp, fp, C.c_float, fp]
    lib.eo_advect_v.argtypes = [sp, fp, fp, C.c_float, fp]
    lib.eo_apply_body_forces.argtypes = [sp, fp, C.c_float]
    lib.eo_project.argtypes = [sp, C.c_float, fp, fp, fp, fp]
    lib.eo_project.restype = C.c_int
    lib.eo_build_system.argtypes = [sp, C.c_float, fp, fp]
    lib.eo_apply_preconditioner.argtypes = [sp, dp, dp]
    lib.eo_apply_a.argtypes = [sp, dp, dp]
    lib.eo_dot.argtypes = [sp, dp, dp]
    lib.eo_dot.restype = C.c_double
    lib.eo_inf_norm.argtypes = [sp, dp]
    lib.eo_inf_norm.restype = C.c_double
    lib.eo_tile_start.argtypes = [C.c_int, C.c_int]
    lib.eo_tile_start.restype = C.c_int
    lib.eo_coarse_m.argtypes = [C.c_int, C.c_int]
    lib.eo_coarse_m.restype = C.c_int
    lib.eo_colorize.argtypes = [sp]
    lib.eo_advect_p.argtypes = [sp, fp, fp, fp, C.c_float, fp]
    lib.eo_render_rows.argtypes = [sp, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.eo_render_rows.restype = C.c_int
    lib.eo_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
    lib.eo_fnv1a64.restype = C.c_uint64
    _LIBS[key] = lib
    return lib


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(oracle_lib().eo_fnv1a64(a.ctypes.data_as(C.c_void_p), a.nbytes))


class Oracle:
    """Thin numpy view over one eo_sim. Arrays alias the C memory (no copies)."""

    FIELDS_F32 = ("u", "v", "utmp", "vtmp", "cr", "cg", "cb", "crtmp", "cgtmp", "cbtmp")
    FIELDS_U8 = ("solid", "source", "sink", "count", "prev_count")
    FIELDS_F64 = ("precon", "q", "b", "p", "r", "z", "s")

    def __init__(self, X, Y, fast=False, lib_path=None, rainbow=False):
        self.lib = oracle_lib(fast, lib_path)
        self.ptr = self.lib.eo_create(X, Y)
        if not self.ptr:
            raise MemoryError("eo_create failed")
        self.ptr.contents.rainbow = int(rainbow)   # before loading: sim_init colours the initial fluid (main.c:270-273)
        self.X, self.Y = X, Y
        c = self.ptr.contents
        shape = (Y, X)
        for n in self.FIELDS_F32 + self.FIELDS_U8 + self.FIELDS_F64:
            setattr(self, n, np.ctypeslib.as_array(getattr(c, n), shape=shape))
        self.a_diag = np.ctypeslib.as_array(c.a_diag, shape=shape)
        self._markers = np.ctypeslib.as_array(C.cast(c.markers, C.POINTER(C.c_float)), shape=(4 * X * Y, 2))

    def close(self):
        if self.ptr:
            self.lib.eo_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state
    @property
    def c(self):
        return self.ptr.contents

    @property
    def n_markers(self):
        return int(self.c.n_markers)

    @property
    def markers(self):
        return self._markers[: self.n_markers]

    def set_markers(self, m):
        m = np.asarray(m, dtype=np.float32).reshape(-1, 2)
        self._markers[: len(m)] = m
        self.c.n_markers = len(m)

    def sorted_markers(self):
        return sort_markers(self.markers)

    # --- loading
    def load_text(self, text, upscale=False):
        if isinstance(text, str):
            text = text.encode()
        rc = self.lib.eo_load_scenario_mem(self.ptr, text, len(text), int(upscale))
        if rc:
            raise ValueError("scenario parse failed: %d" % rc)
        return self

    def load_file(self, path, upscale=False):
        with open(path, "rb") as f:
            return self.load_text(f.read(), upscale)

    def load_half_tank(self):
        self.lib.eo_load_half_tank(self.ptr)
        return self

    # --- stepping
    def step(self):
        self.lib.eo_step(self.ptr)
        return self.c.last_substeps, self.c.last_pcg_iterations

    def substep(self, dt):
        return self.lib.eo_substep(self.ptr, C.c_float(dt))

    def timestep(self, frame_time=0.1):
        return float(self.lib.eo_calculate_timestep(self.ptr, C.c_float(frame_time)))

    def f32p(self, a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def f64p(self, a):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    def render(self, wx, wy):
        cap = (self.X + 32) * (self.Y + 2) * 24   # a coloured cell takes up to 20 bytes
        buf = C.create_string_buffer(cap)
        n = self.lib.eo_render_rows(self.ptr, wx, wy, buf, cap)
        return buf.raw[:n]


def sort_markers(m):
    """Canonical order for comparing marker multisets: by (y bits, x bits)."""
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(-1, 2)
    bits = m.view(np.uint32).astype(np.uint64)
    key = (bits[:, 1] << np.uint64(32)) | bits[:, 0]
    return m[np.argsort(key, kind="stable")]


# ----------------------------------------------------------------------------- compiled reference

REF_SO = os.path.join(ORACLE_DIR, "_ref", "libeuler_ref.so")
REF_X, REF_Y = 100, 40


class ArgsT(C.Structure):
    _fields_ = [("scenario_file", C.c_char_p), ("rainbow", C.c_bool)]


def have_ref():
    return os.path.exists(REF_SO)


class Reference:
    """One private instance of the compiled reference (state is process-global inside the .so,
    so each instance dlopens its own temporary copy)."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        fd, self._tmp = tempfile.mkstemp(suffix=".so", prefix="euler_ref_")
        os.close(fd)
        shutil.copyfile(REF_SO, self._tmp)
        self.lib = C.CDLL(self._tmp)
        os.unlink(self._tmp)
        L = self.lib
        shape = (REF_Y, REF_X)

        def arr(name, ctype):
            return np.ctypeslib.as_array((ctype * (REF_X * REF_Y)).in_dll(L, name)).reshape(shape)

        for n in ("g_u", "g_v", "g_utmp", "g_vtmp"):
            setattr(self, n[2:], arr(n, C.c_float))
        for n in ("g_solid", "g_source", "g_sink"):
            setattr(self, n[2:], arr(n, C.c_uint8))
        self.count = arr("g_marker_count", C.c_uint8)
        self.prev_count = arr("g_prev_marker_count", C.c_uint8)
        self.precon = arr("g_precon", C.c_double)
        self.q = arr("g_q", C.c_double)
        self.a_diag = arr("g_a", C.c_int8)
        self._markers = np.ctypeslib.as_array((C.c_float * (8 * REF_X * REF_Y)).in_dll(L, "g_markers")).reshape(-1, 2)
        self._len = C.c_size_t.in_dll(L, "g_markers_length")
        self._exhausted = C.c_bool.in_dll(L, "g_source_exhausted")
        for n in ("g_r", "g_g", "g_b", "g_rtmp", "g_gtmp", "g_btmp"):
            setattr(self, "c" + n[2:], arr(n, C.c_float))
        self._rainbow = C.c_bool.in_dll(L, "g_rainbow_enabled")
        self._frame_count = C.c_uint16.in_dll(L, "g_frame_count")
        self._wx = C.c_int.in_dll(L, "g_wx")
        self._wy = C.c_int.in_dll(L, "g_wy")
        L.sim_init.argtypes = [ArgsT]
        L.calculate_timestep.argtypes = [C.c_float]
        L.calculate_timestep.restype = C.c_float
        L.advect_markers.argtypes = [C.c_float]
        fp = C.POINTER(C.c_float)
        L.extrapolate.argtypes = [fp, C.c_int]
        L.zero_bounds.argtypes = [fp, C.c_int]
        L.advect_u.argtypes = [fp, fp, C.c_float, fp]
        L.advect_v.argtypes = [fp, fp, C.c_float, fp]
        L.apply_body_forces.argtypes = [fp, C.c_float]
        L.project.argtypes = [C.c_float, fp, fp, fp, fp]
        L.interpolate.restype = C.c_float

    X, Y = REF_X, REF_Y

    @property
    def n_markers(self):
        return int(self._len.value)

    @property
    def markers(self):
        return self._markers[: self.n_markers]

    @property
    def source_exhausted(self):
        return bool(self._exhausted.value)

    def sorted_markers(self):
        return sort_markers(self.markers)

    def init(self, path, rainbow=False):
        self._path = path.encode()
        self._rainbow.value = bool(rainbow)     # main() sets the global before sim_init (main.c:1020)
        self.lib.sim_init(ArgsT(self._path, bool(rainbow)))
        return self

    def step(self):
        self.lib.sim_step()

    def fp(self, a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def render(self, wx, wy):
        """draw_rows() into a buffer_t; returns the bytes."""
        class BufT(C.Structure):
            _fields_ = [("data", C.c_void_p), ("len", C.c_int)]
        self._wx.value, self._wy.value = wx, wy
        b = BufT(None, 0)
        self.lib.draw_rows(C.byref(b))
        out = C.string_at(b.data, b.len) if b.len else b""
        self.lib.buffer_free(C.byref(b))
        return out
