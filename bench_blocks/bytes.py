"""bench_blocks.bytes - the byte table: algorithmic bytes per fluid cell of every launch of a PCG iteration, per mode (the single source of `roofline.achieved` and `bytes_per_cell_iteration`).

Split out of bench.py in round 5 (the contract line and the driver stay there); nothing here is imported by the product."""
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


# Algorithmic bytes per FLUID cell and launch (w = 8 for double vectors, one mask byte per kernel), per mode, for the launches of ONE
# PCG iteration as this build runs it.  ITER_BYTES is the single table: a class's `bytes_per_cell`, the iteration's `bytes_per_cell_iteration`
# (= the sum over the per-iteration classes, checked in summarize()) and `roofline.achieved` all come from it - "never the larger one".
W = 8


_TILE = 4 * W + 1         # k_precond_tile: r -= alpha A s', max |r|, z = M_tile^-1 r, dot(z, r): read r and s' (A s' is formed again from s', not read back); write r, z (E^-1 of an interior tile is a table in LDS) -> 33


_RUPD = 3 * W + 1         # parity mode: the first half of that pass alone (read r, s'; write r; max |r|) -> 25


_SWEEP = 3 * W + 1        # one IC(0) sweep of the reference's factor: read rhs, precon; write the result -> 25 each way


ITER_BYTES, PCG_BYTES = {}, {}


def set_as_stored(stored, p_steps=8):
    """The byte table of the build's launches.  stored False (the default wherever this bench runs: tree dots, one GPU or compact ghost rows): A s' never goes to
    memory - k_search_apply reads s, z and writes s' (3w+1), plus p += alpha s of EIGHT iterations on every eighth pass (read p and the s of eight .. two
    iterations ago, write p: 9w / 8 = 1.125w) -> 34.  stored True (sequential dots, mailboxes, EULER_OPT_TILE_STORE_AS): it also writes A s' -> 42, and the r update reads
    that instead of s'.  p_steps N (mailboxes: 2; EULER_OPT_P_STEPS): p on every N-th pass, (N + 1) w / N -> 35 (4), 37 (2)."""
    apply_ = (4 if stored else 3) * W + 1 + (p_steps + 1.0) / p_steps * W
    ITER_BYTES.clear()
    ITER_BYTES.update({
        "ic0": {"forward_solve": _SWEEP, "backward_solve": _SWEEP, "apply_a": apply_, "update_pr": _RUPD},      # 109 (the reference's five loops: 18w+5 = 149; main.c as written: 212)
        "ic0_tile": {"apply_a": apply_, "precond_tile": _TILE},                                                   # 67
        "ic0_tile2": {"apply_a": apply_, "precond_tile": _TILE, "coarse_cycle": 0.0},
        # multilevel: + the V-cycle: 72 doubles of partial sums per 16 x 64 tile written and read (1.1 B/cell) and the node grids - a node per 64 cells, per node of level 0 the
        # right-hand side written and read, omega / d and the "deep water" byte read on the way down and again on the way up, the result written and read by k_search_apply
        # (~58 B per node: nine stencil entries only where the byte says so), a third more for the levels above: ~1.2 B/cell
        "ic0_tile_mg": {"apply_a": apply_, "precond_tile": _TILE, "coarse_cycle": 2.3},
        "jacobi": {"apply_a": apply_, "update_pr": _RUPD, "jacobi": 2 * W + 1, "dot": 2 * W + 1},
    })
    PCG_BYTES.clear()
    PCG_BYTES.update({m: sum(c.values()) for m, c in ITER_BYTES.items()})


ONCE_PER_SOLVE_BYTES = {"update_pr": 3 * W + 1}      # k_finish_p in the tile modes: the last one or two p += alpha s (read s, p; write p)


PCG_CLASSES = ["forward_solve", "backward_solve", "apply_a", "dot", "update_pr", "update_search", "precond_tile", "coarse_cycle", "jacobi", "resident_pcg"]


KERNEL_OF_CLASS = {"forward_solve": "k_sweep_skew<1", "backward_solve": "k_sweep_skew<2", "precon_factor": "k_sweep_skew<0",
                   "apply_a": "k_search_apply", "dot": "k_dot_partial", "update_pr": "k_update_pr", "precond_tile": "k_precond_tile"}


MODE_NAME = {"ic0": "parity mode: the reference's IC(0), bit-identical iterates",
             "ic0_tile": "roofline mode: tile-local IC(0) (64x%d-cell blocks), NOT the reference's iterates (tolerance parity where PCG converges)",
             "ic0_tile2": "two-level mode: tile-local IC(0) (64x%d-cell blocks) + a coarse correction (<= 256 aggregates, dense inverse), NOT the reference's iterates",
             "ic0_tile_mg": "multilevel mode: tile-local IC(0) (64x%d-cell blocks) + one V-cycle over node grids of 8, 16, 32, ... cells spacing (bilinear interpolation, nine-point Galerkin stencils, dense top level), NOT the reference's iterates",
             "jacobi": "Jacobi stand-in, NOT the reference's iterates"}


TILE_MODES = ("ic0_tile", "ic0_tile2", "ic0_tile_mg")
