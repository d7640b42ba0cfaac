#!/usr/bin/env python3
"""Regenerate tests/golden/*_rainbow.npz from the COMPILED, UNMODIFIED reference run with
g_rainbow_enabled = true (main.c:1020; `--rainbow`): the dye fields g_r/g_g/g_b (main.c:76-78).

Runs only in the authoring container (needs /root/reference and oracle/_ref/libeuler_ref.so).
Data only: per-frame FNV-1a-64 hashes of g_r, g_g, g_b (and of g_u, g_marker_count as a guard that
the dye does not disturb the flow), the three arrays in full after selected frames, and the
draw_rows() bytes with the 24-bit colour escapes (main.c:902-912) for two window sizes.  The scenario
is rebuilt by the tests from the cell grids stored in <scn>_frames.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Reference, fnv1a64  # noqa: E402

PLAN = {"block": (100, [0, 10, 40, 99]), "waterfall": (130, [0, 10, 60, 129]), "filter": (60, [0, 29, 59])}


def main():
    for scn, (nframes, keep) in PLAN.items():
        ref = Reference().init("/root/reference/scenarios/%s.txt" % scn, rainbow=True)
        out = {"init_r": ref.cr.copy(), "init_g": ref.cg.copy(), "init_b": ref.cb.copy(), "frames_full": np.array(keep, dtype=np.int64)}
        hs = []
        for f in range(nframes):
            ref.step()
            hs.append([fnv1a64(ref.cr), fnv1a64(ref.cg), fnv1a64(ref.cb), fnv1a64(ref.u), fnv1a64(ref.count)])
            if f in keep:
                out["f%d_r" % f], out["f%d_g" % f], out["f%d_b" % f] = ref.cr.copy(), ref.cg.copy(), ref.cb.copy()
                for (wx, wy) in ((98, 38), (40, 10)):
                    out["f%d_w%dx%d" % (f, wx, wy)] = np.frombuffer(ref.render(wx, wy), dtype=np.uint8)
        out["hashes"] = np.array(hs, dtype=np.uint64)
        np.savez_compressed(os.path.join(HERE, scn + "_rainbow.npz"), **out)
        print(scn, nframes, "hash_r %016x" % hs[-1][0])


if __name__ == "__main__":
    main()
