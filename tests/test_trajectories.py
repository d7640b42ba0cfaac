"""tests/golden/trajectories.json (the recorded oracle legs of the GPU parity tests, tests/trajectories.py) against the live oracle: every trajectory is on file
with the right number of frames, and the cheap ones are replayed here digest by digest - a stale record fails on the CPU before it can pass or fail anything on the GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

import trajectories as T
from golden_util import GOLDEN

CHEAP = ["ragged_112x48_filter", "ragged_130x70_block", "ragged_257x129_filter", "diffusion_192x200_waterfall", "dye_200x150_waterfall", "ragged_320x192_weird-edges"]


def test_every_trajectory_is_on_file():
    rec = T.records()
    for name, spec in T.SPECS.items():
        assert name in rec, "run tests/golden/make_trajectories.py"
        assert len(rec[name]["frames"]) == (spec.get("substeps") or spec["frames"])
        assert ("init" in rec[name]) == bool(spec.get("init")) and ("render" in rec[name]) == bool(spec.get("render"))
        assert set(spec["fields"]) <= set(rec[name]["frames"][-1])      # (`every` = n: the arrays' digests on every n-th frame and the last)


@pytest.mark.parametrize("name", CHEAP)
def test_recorded_digests_are_the_live_oracle(name):
    spec, rec = T.SPECS[name], T.records()[name]
    o = T.make_oracle(spec)
    if spec.get("init"):
        assert T.snapshot(o, spec) == rec["init"]
    for f in range(spec["frames"]):
        o.step()
        assert T.snapshot(o, spec) == rec["frames"][f], (name, f)
    o.close()


BIG = sorted(n for n, spec in T.SPECS.items() if spec.get("big"))
ORACLE_SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "euler_oracle.c")


def oracle_sha1():
    with open(ORACLE_SRC, "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def test_the_recorded_oracle_legs_belong_to_this_oracle_source():
    """The big records (BASELINE-sized trajectories: an hour of one core) are never replayed on an ordinary run, so nothing would say that an edit of
    oracle/euler_oracle.c has made them stale - a stale digest can only ever FAIL a GPU test, but it fails it for the wrong reason.  tests/golden/MANIFEST.json
    therefore holds the SHA-1 of the source the records were last replayed against (`oracle_source`), and mg_records.npz the one it was generated from: after an
    edit of the oracle, run `EULER_REPLAY_BIG=1 python -m pytest tests/test_trajectories.py -k big` (and tests/golden/make_mg_records.py), then
    `python tests/golden/make_trajectories.py --stamp`."""
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        man = json.load(f)
    src = oracle_sha1()
    assert man.get("oracle_source", {}).get("sha1") == src, "oracle/euler_oracle.c changed since the recorded trajectories were last replayed (see this test's docstring)"
    assert sorted(man["oracle_source"]["big_records_replayed"]) == BIG
    with np.load(os.path.join(GOLDEN, "mg_records.npz")) as z:
        assert bytes(z["oracle_sha1"]).decode() == src, "tests/golden/mg_records.npz was generated from another oracle/euler_oracle.c: run tests/golden/make_mg_records.py"


@pytest.mark.skipif(os.environ.get("EULER_REPLAY_BIG") != "1", reason="opt-in (EULER_REPLAY_BIG=1): minutes to an hour of one core and up to 12 GB per trajectory")
@pytest.mark.parametrize("name", BIG)
def test_big_recorded_digests_are_the_live_oracle(name):
    spec, rec = T.SPECS[name], T.records()[name]
    o = T.make_oracle(spec)
    n, every = spec.get("substeps") or spec["frames"], spec.get("every", 1)
    for f in range(n):
        T.advance(o, spec)
        full = (f + 1) % every == 0 or f == n - 1
        assert T.snapshot(o, spec if full else dict(spec, fields=())) == rec["frames"][f], (name, f)
    o.close()


def test_a_mismatch_falls_back_to_the_live_oracle_with_the_same_diagnostics():
    import numpy as np
    o = T.Recorded("ragged_130x70_block")
    o.step()
    live = T.make_oracle(T.SPECS["ragged_130x70_block"])
    live.step()
    T.same(live.u.copy(), o, "u", "u")                       # the right array: the digest matches, no replay
    assert o._live is None
    bad = live.u.copy()
    bad[12, 34] += 1.0
    with pytest.raises(AssertionError, match=r"1 of 9100 entries differ, first at .*12.*34"):
        T.same(bad, o, "u", "u")
    assert np.array_equal(o.live().u, live.u)
