"""Helpers shared by the golden-vector tests (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENARIOS = ("basic", "block", "filter", "waterfall", "weird-edges")
X, Y = 100, 40


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def scenario_text(frames_npz):
    """Rebuild scenario text (reference format, README.md:53-56: X wall, 0 fluid, ? source,
    = sink) from the parsed cell grids stored in a *_frames.npz fixture."""
    solid, source, sink = frames_npz["solid"], frames_npz["source"], frames_npz["sink"]
    fluid = frames_npz["init_count"] > 0
    lines = []
    for y in range(Y - 2, 0, -1):
        row = []
        for x in range(1, X - 1):
            if solid[y, x]:
                row.append("X")
            elif source[y, x]:
                row.append("?")
            elif fluid[y, x]:
                row.append("0")
            elif sink[y, x]:
                row.append("=")
            else:
                row.append(" ")
        lines.append("".join(row))
    return "\n".join(lines) + "\n"


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
