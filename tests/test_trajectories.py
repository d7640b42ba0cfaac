"""tests/golden/trajectories.json (the recorded oracle legs of the GPU parity tests, tests/trajectories.py) against the live oracle: every trajectory is on file
with the right number of frames, and the cheap ones are replayed here digest by digest - a stale record fails on the CPU before it can pass or fail anything on the GPU."""
import pytest

import trajectories as T

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
