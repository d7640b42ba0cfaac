"""State snapshots (include/euler.h "state snapshots"; SURVEY §8f item 2: checkpoint / resume).

CPU: the numpy mirror of the on-disk format round-trips a reference-generated state.
GPU: (1) a state written from the compiled reference's golden fixture resumes on the GPU and lands,
30 frames later, bit for bit on the reference's own later fixture; (2) save -> destroy -> load ->
continue equals the uninterrupted run bit for bit, and the C writer's file parses with the numpy
reader."""
import os

import numpy as np
import pytest

import euler_amd as ea
from golden_util import X, Y, bits_equal, load


def fixture_state(frames, tag):
    st = {k: frames["%s_%s" % (tag, k)] for k in ("u", "v", "count", "prev_count", "precon", "markers")}
    st.update(solid=frames["solid"], source=frames["source"], sink=frames["sink"], utmp=np.zeros((Y, X), np.float32),
              vtmp=np.zeros((Y, X), np.float32), rng_state=int(frames[tag + "_rng"]), source_exhausted=int(frames[tag + "_exhausted"]))
    return st


def test_snapshot_format_round_trip(tmp_path):
    st = fixture_state(load("waterfall_frames.npz"), "f10")
    p = str(tmp_path / "s.snap")
    ea.write_snapshot(p, st)
    back = ea.read_snapshot(p)
    assert (back["X"], back["Y"], back["n_markers"]) == (X, Y, len(st["markers"]))
    assert back["rng_state"] == st["rng_state"] and back["source_exhausted"] == st["source_exhausted"]
    for k in ea.SNAPSHOT_F32 + ea.SNAPSHOT_U8 + ("precon", "markers"):
        assert bits_equal(back[k], np.ascontiguousarray(st[k], back[k].dtype).reshape(back[k].shape)), k
    raw = bytearray(open(p, "rb").read())
    raw[200] ^= 1                                            # one flipped bit must be caught
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        ea.read_snapshot(p)


@pytest.mark.gpu
@pytest.mark.parametrize("scn,start,end", [("block", 10, 40), ("waterfall", 10, 100), ("filter", 10, 28), ("waterfall", 449, 459)])
def test_resume_from_reference_state_reaches_reference_state(tmp_path, scn, start, end):
    frames = load(scn + "_frames.npz")
    p = str(tmp_path / "ref.snap")
    ea.write_snapshot(p, fixture_state(frames, "f%d" % start))       # the compiled reference's state after frame `start`
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_SEQUENTIAL).load_state(p)
    for _ in range(end - start):
        sim.step()
    for k, f in (("u", ea.F_U), ("v", ea.F_V), ("count", ea.F_COUNT), ("prev_count", ea.F_PREV_COUNT), ("precon", ea.F_PRECON),
                 ("markers", ea.F_MARKERS)):
        assert bits_equal(sim.get(f), frames["f%d_%s" % (end, k)]), "%s after resume: %s" % (scn, k)
    st = sim.stats()
    assert st.rng_state == int(frames["f%d_rng" % end]) and st.source_exhausted == int(frames["f%d_exhausted" % end])
    sim.close()


@pytest.mark.gpu
def test_save_load_continue_is_bit_identical(tmp_path):
    from golden_util import scenario_text
    text = scenario_text(load("waterfall_frames.npz"))
    a = ea.Simulation(192, 136, dot_mode=ea.DOT_SEQUENTIAL).load_text(text, upscale=True)
    for _ in range(6):
        a.step()
    p = str(tmp_path / "mid.snap")
    a.save_state(p)
    snap = ea.read_snapshot(p)                               # the C writer's file through the numpy reader
    assert bits_equal(snap["u"], a.get(ea.F_U)) and bits_equal(snap["markers"], a.get(ea.F_MARKERS))
    assert snap["frames"] == 6 and snap["rng_state"] == a.stats().rng_state
    b = ea.Simulation(192, 136, dot_mode=ea.DOT_SEQUENTIAL).load_state(p)
    for _ in range(5):
        a.step(); b.step()
    for f in (ea.F_U, ea.F_V, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_PRECON, ea.F_MARKERS, ea.F_PRESSURE):
        assert bits_equal(a.get(f), b.get(f)), f
    assert a.stats().frames == b.stats().frames == 11
    with pytest.raises(ea.EulerError):
        ea.Simulation(100, 40).load_state(p)                 # wrong grid size
    a.close(); b.close()


@pytest.mark.gpu
def test_cli_checkpoint_and_resume(tmp_path):
    """`euler --checkpoint` after 2 frames, `euler --resume` for 1 more = the reference's frame after 3 steps."""
    import subprocess
    from golden_util import scenario_text
    g, r = load("block_frames.npz"), load("block_render.npz")
    scn = tmp_path / "block.txt"
    scn.write_text(scenario_text(g))
    exe = os.path.join(os.path.dirname(ea.LIB_PATH), "..", "bin", "euler")
    snap = str(tmp_path / "two.snap")
    a = subprocess.run([exe, "--dump", "--frames", "1", "--checkpoint", snap, str(scn)], capture_output=True, timeout=120)
    assert a.returncode == 0, a.stderr.decode()
    assert ea.read_snapshot(snap)["frames"] == 1
    b = subprocess.run([exe, "--dump", "--frames", "1", "--window", "98x38", "--resume", snap], capture_output=True, timeout=120)
    assert b.returncode == 0, b.stderr.decode()
    header, body = b.stdout.split(b"--- frame ")[2].split(b"\n", 1)
    n = int(header.split(b"(")[1].split()[0])
    assert body[:n] == r["f1_w98x38"].tobytes()              # draw_rows() bytes after the 2nd sim_step()
