"""Oracle vs the compiled reference itself (oracle/_ref/libeuler_ref.so), longer than the
committed fixtures.  Skipped where the reference build is absent."""
import glob
import os

import numpy as np
import pytest

from golden_util import bits_equal, load, scenario_text, SCENARIOS
from oracle_lib import Oracle, Reference, have_ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("scn,frames", [("basic", 200), ("block", 150), ("filter", 200), ("weird-edges", 150)])
def test_lockstep(tmp_path, scn, frames):
    g = load(scn + "_frames.npz")
    path = tmp_path / (scn + ".txt")
    path.write_text(scenario_text(g))
    ref = Reference().init(str(path))
    o = Oracle(100, 40).load_file(str(path))
    for f in range(frames):
        ref.step()
        o.step()
        for n in ("u", "v", "count", "prev_count", "precon"):
            assert bits_equal(getattr(ref, n), getattr(o, n)), (scn, f, n)
        assert bits_equal(ref.markers, o.markers), (scn, f)
