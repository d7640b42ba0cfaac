"""--rainbow dye on the GPU: g_r/g_g/g_b through the C ABI against the golden fixtures
generated from the compiled reference run with g_rainbow_enabled (100x40, bit-exact, free-running), and
against the CPU oracle on larger ragged grids.  The coloured draw_rows bytes are compared too."""
import numpy as np
import pytest

import euler_amd as ea
from euler_amd import scenarios
from golden_util import X, Y, bits_equal, load, scenario_text
from oracle_lib import Oracle, fnv1a64
from test_gpu_parity import assert_bits

pytestmark = pytest.mark.gpu

DYE = ((ea.F_DYE_R, "r"), (ea.F_DYE_G, "g"), (ea.F_DYE_B, "b"))


@pytest.mark.parametrize("scn", ["block", "waterfall", "filter"])
def test_dye_free_running_bit_exact_vs_reference(scn):
    g = load(scn + "_rainbow.npz")
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL, rainbow=True).load_text(scenario_text(load(scn + "_frames.npz")))
    for fld, n in DYE:
        assert_bits(sim.get(fld), g["init_" + n], "init " + n)       # colorize at sim_init (main.c:270-273)
    keep = set(int(f) for f in g["frames_full"])
    for f in range(len(g["hashes"])):
        sim.step()
        got = [fnv1a64(sim.get(fld)) for fld, _ in DYE] + [fnv1a64(sim.get(ea.F_U)), fnv1a64(sim.get(ea.F_COUNT))]
        if f in keep:
            for fld, n in DYE:
                assert_bits(sim.get(fld), g["f%d_%s" % (f, n)], "%s frame %d %s" % (scn, f, n))
            for (wx, wy) in ((98, 38), (40, 10)):
                assert sim.draw(wx, wy) == g["f%d_w%dx%d" % (f, wx, wy)].tobytes(), (scn, f, wx, wy)
        assert got == [int(h) for h in g["hashes"][f]], (scn, f)


@pytest.mark.parametrize("X_,Y_,workload,frames", [(200, 150, "waterfall", 40), (256, 160, "dam_break", 30)])
def test_dye_vs_oracle_on_larger_grids(X_, Y_, workload, frames):
    import hashlib
    from trajectories import Recorded, oracle_for, same, vmax
    text = getattr(scenarios, workload)()
    o = oracle_for("dye_%dx%d_%s" % (X_, Y_, workload))      # (Oracle(X_, Y_, rainbow=True) on this scenario; recorded: tests/trajectories.py)
    sim = ea.Simulation(X_, Y_, dot_mode=ea.DOT_SEQUENTIAL, rainbow=True).load_text(text, upscale=True)
    moved = False
    for f in range(frames):
        o.step()
        sim.step()
        for fld, a in ((ea.F_DYE_R, "cr"), (ea.F_DYE_G, "cg"), (ea.F_DYE_B, "cb"), (ea.F_DYE_RTMP, "crtmp"), (ea.F_U, "u")):
            same(sim.get(fld), o, a, "%s frame %d field %d" % (workload, f, fld), nan_class=True)
        moved = moved or vmax(o) > 0
    assert moved
    if isinstance(o, Recorded):
        assert hashlib.sha1(sim.draw(X_, Y_)).hexdigest()[:20] == o.render_digest() or sim.draw(X_, Y_) == o.live().render(X_, Y_)
    else:
        assert sim.draw(X_, Y_) == o.render(X_, Y_)


def test_colorize_key_and_snapshot_round_trip(tmp_path):
    """The 'r' key (euler_colorize, main.c:970-973) and a version-2 snapshot: a resumed handle continues bit for bit."""
    text = scenarios.dam_break()
    a = ea.Simulation(160, 96, dot_mode=ea.DOT_SEQUENTIAL, rainbow=True).load_text(text, upscale=True)
    o = Oracle(160, 96, rainbow=True).load_text(text, upscale=True)
    for _ in range(12):
        a.step(); o.step()
    a.colorize(); o.lib.eo_colorize(o.ptr)
    assert_bits(a.get(ea.F_DYE_R), o.cr, "r after colorize")
    assert not bits_equal(a.get(ea.F_DYE_R), a.get(ea.F_DYE_RTMP))     # only now do g_r and g_rtmp differ
    p = str(tmp_path / "dye.snap")
    a.save_state(p)
    snap = ea.read_snapshot(p)
    assert bits_equal(snap["dye_g"], o.cg) and bits_equal(snap["dye_rtmp"], o.crtmp)
    b = ea.Simulation(160, 96, dot_mode=ea.DOT_SEQUENTIAL, rainbow=True).load_state(p)
    for _ in range(5):
        a.step(); b.step(); o.step()
    for fld, want in ((ea.F_DYE_R, o.cr), (ea.F_DYE_G, o.cg), (ea.F_DYE_B, o.cb), (ea.F_U, o.u), (ea.F_V, o.v)):
        assert_bits(b.get(fld), want, "resumed field %d" % fld)
        assert_bits(a.get(fld), want, "original field %d" % fld)
    plain = ea.Simulation(160, 96)
    with pytest.raises(ea.EulerError):
        plain.load_state(p)                   # a handle without the dye cannot take a version-2 snapshot
    with pytest.raises(ea.EulerError):
        plain.get(ea.F_DYE_R)


def _cli_frames(args, timeout=180):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(ea.LIB_PATH), "..", "bin", "euler")
    out = subprocess.run([exe, "--dump"] + args, capture_output=True, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()
    frames = []
    for chunk in out.stdout.split(b"--- frame ")[1:]:
        header, body = chunk.split(b"\n", 1)
        frames.append(body[: int(header.split(b"(")[1].split()[0])])
    return frames, out.stderr.decode()


def test_cli_rainbow_dump_matches_reference_frames(tmp_path):
    """`euler --rainbow` (main.c:992, 1020): frame k = the reference's coloured draw_rows() bytes after k steps."""
    gr = load("waterfall_rainbow.npz")
    scn = tmp_path / "waterfall.txt"
    scn.write_text(scenario_text(load("waterfall_frames.npz")))
    frames, _ = _cli_frames(["--rainbow", "--frames", "11", "--window", "98x38", str(scn)])
    assert len(frames) == 12
    assert frames[1] == gr["f0_w98x38"].tobytes()
    assert frames[11] == gr["f10_w98x38"].tobytes()


def test_cli_keys_pause_gate_and_quit(tmp_path):
    """process_keypress + the pause gate of sim_step (main.c:844-846, 896-898, 961-980), scripted with --keys:
    p pauses (frames repeat, no step is taken), f advances exactly one frame while paused, p resumes, q quits."""
    r = load("block_render.npz")
    scn = tmp_path / "block.txt"
    scn.write_text(scenario_text(load("block_frames.npz")))
    #            frame:  1  2  3  4  5  6  7   (frame 0 = the initial draw)
    frames, err = _cli_frames(["--keys", ".p.f.pq", "--frames", "20", "--window", "98x38", str(scn)])
    assert len(frames) == 7                                  # 'q' at frame 7 ends the loop before a step or a draw
    f0, f1 = r["f0_w98x38"].tobytes(), r["f1_w98x38"].tobytes()
    assert frames[1] == f0                                   # '.': one step
    assert frames[2] == f0 and frames[3] == f0               # 'p', '.': paused, nothing moves
    assert frames[4] == f1                                   # 'f': exactly one more step
    assert frames[5] == f1                                   # '.': still paused
    assert frames[6] != f1                                   # 'p': resumed, third step taken
    assert "frames 3 " in err


def test_cli_interactive_loop_on_a_pty(tmp_path):
    """The reference's terminal loop (main.c:1016-1042, misc/terminal.c) for real: `euler --rainbow` on a
    pseudo-terminal - window size from TIOCGWINSZ, raw mode, frames drawn with reposition / hide-cursor codes at
    10 Hz, the keys p (pause), f (one frame), r (recolour), q (quit: screen cleared, cursor shown, exit 0)."""
    import fcntl
    import os
    import pty
    import select
    import struct
    import subprocess
    import termios
    import time
    scn = tmp_path / "waterfall.txt"
    scn.write_text(scenario_text(load("waterfall_frames.npz")))
    exe = os.path.join(os.path.dirname(ea.LIB_PATH), "..", "bin", "euler")
    try:
        master, slave = pty.openpty()
    except OSError as e:                     # containers without /dev/ptmx: the key handler and the gate are covered by --keys above
        pytest.skip("no pseudo-terminal available: %s" % e)
    fcntl.ioctl(slave, termios.TIOCSWINSZ, struct.pack("HHHH", 30, 90, 0, 0))     # 30 rows x 90 columns
    p = subprocess.Popen([exe, "--rainbow", str(scn)], stdin=slave, stdout=slave, stderr=subprocess.PIPE, close_fds=True)
    os.close(slave)
    out = b""

    def pump(seconds):
        nonlocal out
        end = time.time() + seconds
        while time.time() < end:
            r, _, _ = select.select([master], [], [], 0.05)
            if r:
                try:
                    out += os.read(master, 1 << 16)
                except OSError:
                    return

    try:
        pump(1.5)                      # a dozen frames
        n_running = out.count(b"\x1b[H")
        os.write(master, b"p")
        pump(0.6)
        os.write(master, b"f")
        pump(0.4)
        os.write(master, b"r")
        pump(0.4)
        os.write(master, b"q")
        p.wait(timeout=20)
    finally:
        if p.poll() is None:
            p.kill()
        pump(0.2)
        os.close(master)
    err = p.stderr.read().decode()
    assert p.returncode == 0, err
    assert n_running >= 5                                   # frames were being drawn at ~10 Hz
    assert b"\x1b[?25l" in out and b"\x1b[38;2;" in out     # cursor hidden, 24-bit colour escapes of the dye
    assert out.rstrip().endswith(b"\x1b[?25h")              # quit: cursor shown again (after the final clear)
    frames = int(err.split("frames ")[1].split()[0])
    assert 5 <= frames < 60                                 # paused for a while: far fewer frames than wall time x 10 Hz would give
    rows = [l for l in out.split(b"\x1b[H")[2].split(b"\r\n")]
    assert len(rows) == 30 - 0 or len(rows) <= 38           # the window height bounds the rows drawn (Y - 2 = 38 at most)
