"""No-GPU tests of the product library's boundary: libeuler_hip.so loads, exports every symbol
include/euler.h declares, refuses to run without a gfx950 device, and its host-only C pieces
(scenario parser, marker seeding, frame formatter) reproduce the reference's golden data."""
import os
import re

import numpy as np
import pytest

import euler_amd as ea
from golden_util import SCENARIOS, X, Y, bits_equal, load, scenario_text
from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "euler.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(euler_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = ea.load_library()
    for name in sorted(declared):
        assert hasattr(L, name), "include/euler.h declares %s but libeuler_hip.so does not export it" % name
    assert declared == set(ea.EXPORTS)
    assert L.euler_abi_version() == 2


def test_python_constants_mirror_the_header_enums():
    """the ctypes wrapper's EULER_* values are the header's (preconditioner modes, dot modes, sweep modes, error codes)"""
    hdr = open(os.path.join(ROOT, "include", "euler.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    values = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(EULER_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr)}
    for py, c in (("PRECOND_IC0", "EULER_PRECOND_IC0"), ("PRECOND_JACOBI", "EULER_PRECOND_JACOBI"), ("PRECOND_IC0_TILE", "EULER_PRECOND_IC0_TILE"),
                  ("PRECOND_IC0_TILE2", "EULER_PRECOND_IC0_TILE2"), ("PRECOND_IC0_TILE_MG", "EULER_PRECOND_IC0_TILE_MG"),
                  ("DOT_TREE", "EULER_DOT_TREE"), ("DOT_SEQUENTIAL", "EULER_DOT_SEQUENTIAL")):
        assert c in values, c
        assert getattr(ea, py) == values[c], (py, getattr(ea, py), values[c])
    # the per-handle options (euler_set_option): every EULER_OPT_* of the header has its OPT_* twin with the same number, and the wrapper invents none
    opts = {k[len("EULER_"):]: v for k, v in values.items() if k.startswith("EULER_OPT_")}
    assert len(opts) >= 20
    for name, v in opts.items():
        assert getattr(ea, name) == v, (name, getattr(ea, name, None), v)
    assert {k for k in dir(ea) if k.startswith("OPT_")} == set(opts)


def test_no_cpu_fallback():
    """Without a GPU the handle cannot be created: the product has no CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ea.EulerError) as e:
        ea.Simulation(100, 40)
    assert e.value.code == -4 and "no CPU path" in str(e.value)


def test_product_never_links_the_oracle():
    import subprocess
    out = subprocess.check_output(["nm", "-D", ea.LIB_PATH]).decode()
    assert "eo_" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "euler_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip")):
                src = open(os.path.join(root, f)).read()
                assert "euler_oracle" not in src and "liboracle" not in src and "oracle_lib" not in src, f


@pytest.mark.parametrize("scn", SCENARIOS)
def test_parser_and_seeding_match_reference(scn):
    g = load(scn + "_frames.npz")
    solid, source, sink, fluid = ea.parse_scenario(scenario_text(g), X, Y)
    assert bits_equal(solid, g["solid"]) and bits_equal(source, g["source"]) and bits_equal(sink, g["sink"])
    assert bits_equal(fluid, (g["init_count"] > 0).astype(np.uint8))
    m, rng = ea.seed_markers(fluid)
    assert bits_equal(m, g["init_markers"])
    assert rng == int(g["init_rng"])


def test_parser_edge_cases_match_oracle():
    cases = [
        "",                                   # empty file: nothing but the sink ring
        "\n\n\n",                             # blank lines
        "X" * 300 + "\n0?=\n",                # over-long first line is truncated, its tail discarded
        "0" * 98,                             # exactly X-2 chars, no newline at EOF
        "0" * 98 + "\n" + "?" * 98 + "\n",    # exact-width lines consume their own newline
        "ab\r\n0\r\n",                        # '\r' and letters are "other" = empty
        "\n".join(["0?X= "] * 60) + "\n",     # more lines than rows: surplus ignored
    ]
    for text in cases:
        got = ea.parse_scenario(text, X, Y)
        o = Oracle(X, Y).load_text(text)
        assert bits_equal(got[0], o.solid) and bits_equal(got[1], o.source) and bits_equal(got[2], o.sink), repr(text[:20])
        m, _ = ea.seed_markers(got[3])
        assert len(m) == o.n_markers and bits_equal(m, o.markers)
        assert bits_equal(got[3], (o.count > 0).astype(np.uint8))


@pytest.mark.parametrize("size", [(256, 256), (333, 127), (1024, 64)])
def test_upscale_rule_matches_oracle(size):
    X2, Y2 = size
    for scn in ("block", "waterfall"):
        text = scenario_text(load(scn + "_frames.npz"))
        got = ea.parse_scenario(text, X2, Y2, upscale=True)
        o = Oracle(X2, Y2).load_text(text, upscale=True)
        assert bits_equal(got[0], o.solid) and bits_equal(got[1], o.source) and bits_equal(got[2], o.sink)
        m, rng = ea.seed_markers(got[3])
        assert len(m) == o.n_markers and bits_equal(m, o.markers)
        assert rng == int(o.c.rng_state)


@pytest.mark.parametrize("scn", SCENARIOS)
def test_render_matches_reference(scn):
    g = load(scn + "_frames.npz")
    r = load(scn + "_render.npz")
    for key in r.files:
        f, w = key.split("_")
        wx, wy = (int(t) for t in w[1:].split("x"))
        got = ea.render_grids(g["solid"], g["sink"], g["%s_count" % f], wx, wy)
        assert got == r[key].tobytes(), key


def test_coloured_frame_formatter_matches_reference_bytes():
    """euler_render_grids_rgb (host C, no GPU) on the dye arrays the compiled reference produced
    (tests/golden/*_rainbow.npz): the 24-bit colour escapes of main.c:902-912 byte for byte."""
    import numpy as np
    from golden_util import load
    for scn in ("block", "waterfall"):
        g, gr = load(scn + "_frames.npz"), load(scn + "_rainbow.npz")
        f = int(gr["frames_full"][1])
        if "f%d_count" % f not in g.files:
            continue
        rgb = tuple(gr["f%d_%s" % (f, c)] for c in "rgb")
        for (wx, wy) in ((98, 38), (40, 10)):
            got = ea.render_grids(g["solid"], g["sink"], g["f%d_count" % f], wx, wy, rgb=rgb)
            assert got == gr["f%d_w%dx%d" % (f, wx, wy)].tobytes(), (scn, f, wx, wy)
