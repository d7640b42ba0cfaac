"""Scenario text generators in the reference's format (README.md:53-56: `X` wall, `0` fluid,
`?` source, `=` sink; first text line is the top row).  These are this build's own inputs for the
BASELINE.json configurations; tests check that `dam_break()` and `waterfall()` parse to the same
cell grids as the reference's scenarios/block.txt and scenarios/waterfall.txt."""


def _canvas(w, h):
    return [[" "] * w for _ in range(h)]


def _text(rows):
    return "\n".join("".join(r) for r in rows) + "\n"


def dam_break():
    """95x38 picture: a walled box with a block of water hanging in the left half (the layout of
    the reference's block.txt: water at picture columns 2..52, lines 8..29)."""
    w, h = 95, 38
    g = _canvas(w, h)
    for x in range(w):
        g[0][x] = g[h - 1][x] = "X"
    for y in range(h):
        g[y][0] = g[y][w - 1] = "X"
    for y in range(8, 30):
        for x in range(2, 53):
            g[y][x] = "0"
    return _text(g)


def waterfall():
    """98x38 picture with a source patch top-left, two ledges and a sink column bottom-left (the
    layout of the reference's waterfall.txt)."""
    w, h = 98, 38
    g = _canvas(w, h)
    for x in range(w):
        g[0][x] = g[h - 1][x] = "X"
    for y in range(h):
        g[y][w - 1] = "X"
    for y in range(1, 28):
        g[y][0] = "X"
    for y in range(1, 4):
        for x in range(1, 21):
            g[y][x] = "?"
    for y in range(24, 27):
        g[y][25] = "X"
    for x in range(0, 26):
        g[27][x] = "X"
    for y in range(28, 32):
        g[y][25] = "X"
        g[y][50] = "X"
    for x in range(0, 51):
        g[32][x] = "X"
    for y in range(33, 37):
        g[y][0] = "="
    return _text(g)


def stacked(text, copies):
    """`copies` pictures on top of each other (weak-scaling workloads: upscaled onto an X x (copies*Y) grid,
    every row slab of a multi-GPU job gets one copy of the single-GPU scenario - closed tanks, walls in between)."""
    if not text.endswith("\n"):
        text += "\n"
    return text * max(1, int(copies))

# (config 3, the half-filled tank, is generated on the grid directly: euler_load_half_tank / euler_load_half_tanks)
