"""The reference's own entry points over libeuler_hip.so (SURVEY §8b "compatibility shim").

oracle/_ref/libeuler_ref_on_hip.so (oracle/Makefile `ref_on_hip`, built where /root/reference exists and carried to
the GPU box prebuilt) = the UNMODIFIED reference main.c + misc/*.c with its own sim_init / sim_step / draw_rows /
colorize made weak, linked with the product's shim euler_amd/compat/euler_compat.c.  The reference's main loop, key
handling, draw() and its globals are all there, unchanged - and drive the GPU.  The tests call it exactly the way
tests/golden/make_golden.py drives the compiled reference."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import load, scenario_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libeuler_ref_on_hip.so")
needs_so = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libeuler_ref_on_hip.so not built (needs the reference tree)")


class ArgsT(C.Structure):
    _fields_ = [("scenario_file", C.c_char_p), ("rainbow", C.c_bool)]


class BufT(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_int)]


@needs_so
def test_the_references_main_reaches_the_shim_without_a_gpu(tmp_path):
    """No GPU here: the reference's own main() (parse_args, window size, sim_init ...) must end in the shim's
    sim_init, which reports the library's error and exits 1 like main.c:212-215 - never a CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    scn = tmp_path / "block.txt"
    scn.write_text(scenario_text(load("block_frames.npz")))
    code = ("import ctypes as C, sys\n"
            "L = C.CDLL(%r)\n"
            "C.c_int.in_dll(L, 'g_wx').value = 98; C.c_int.in_dll(L, 'g_wy').value = 38\n"
            "class A(C.Structure): _fields_ = [('f', C.c_char_p), ('r', C.c_bool)]\n"
            "L.sim_init.argtypes = [A]\n"
            "L.sim_init(A(%r, False))\n" % (SO, str(scn).encode()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "no HIP device" in r.stderr and "no CPU path" in r.stderr


@needs_so
@pytest.mark.gpu
@pytest.mark.parametrize("scn,rainbow", [("block", False), ("waterfall", True)])
def test_reference_names_drive_the_gpu(tmp_path, scn, rainbow):
    """sim_init(args_t) / sim_step() / draw_rows(buffer_t*) / colorize() with the reference's own types and globals:
    frame k's bytes = the compiled reference's draw_rows() bytes; the pause gate and g_frame_count behave like main.c."""
    g = load(scn + "_frames.npz")
    want = load(scn + ("_rainbow.npz" if rainbow else "_render.npz"))
    path = tmp_path / (scn + ".txt")
    path.write_text(scenario_text(g))
    L = C.CDLL(SO)
    wx, wy = C.c_int.in_dll(L, "g_wx"), C.c_int.in_dll(L, "g_wy")
    wx.value, wy.value = 98, 38
    C.c_bool.in_dll(L, "g_rainbow_enabled").value = rainbow         # main() sets it before sim_init (main.c:1020)
    # the library's state is process-global like the reference's (one dlopen per process): start from main()'s initial values
    C.c_uint16.in_dll(L, "g_frame_count").value = 0
    C.c_bool.in_dll(L, "g_pause").value = False
    C.c_uint32.in_dll(L, "g_temp_unpause_counter").value = 0
    L.sim_init.argtypes = [ArgsT]
    L.euler_compat_handle.restype = C.c_void_p
    L.sim_init(ArgsT(str(path).encode(), rainbow))                   # a second sim_init replaces the handle
    assert L.euler_compat_handle()                                   # the three names sit on a libeuler_hip handle

    def frame():
        b = BufT(None, 0)
        L.draw_rows(C.byref(b))                                      # appends through the reference's buffer_append
        out = C.string_at(b.data, b.len) if b.len else b""
        L.buffer_free(C.byref(b))
        return out

    frames = C.c_uint16.in_dll(L, "g_frame_count")
    pause = C.c_bool.in_dll(L, "g_pause")
    unpause = C.c_uint32.in_dll(L, "g_temp_unpause_counter")
    L.sim_step()
    assert frames.value == 1 and frame() == want["f0_w98x38"].tobytes()
    pause.value = True
    L.sim_step(); L.sim_step()
    assert frames.value == 1 and frame() == want["f0_w98x38"].tobytes()      # paused: nothing moves (main.c:844-846)
    unpause.value = 1
    L.sim_step()
    assert frames.value == 2 and unpause.value == 0                           # 'f': exactly one frame (main.c:896-898)
    pause.value = False
    for _ in range(9):
        L.sim_step()
    assert frames.value == 11 and frame() == want["f10_w98x38"].tobytes()
    if rainbow:
        L.colorize()                                                  # the 'r' key (main.c:970-973)
        assert frame() != want["f10_w98x38"].tobytes()
    else:
        # the same names over another preconditioner (EULER_COMPAT_SOLVER): block.txt's solves converge within the cap in the first frames, so the frames are the same
        os.environ["EULER_COMPAT_SOLVER"], os.environ["EULER_COMPAT_MAX_ITERATIONS"] = "multilevel", "400"
        try:
            frames.value = 0
            L.sim_init(ArgsT(str(path).encode(), rainbow))
            L.sim_step()
            assert frames.value == 1 and frame() == want["f0_w98x38"].tobytes()
        finally:
            del os.environ["EULER_COMPAT_SOLVER"], os.environ["EULER_COMPAT_MAX_ITERATIONS"]
