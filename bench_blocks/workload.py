"""bench_blocks.workload - handles and workloads: creating a handle (whole grid or row slab), attaching the exchanges, pre-rolling into the expensive phase, fluid-balanced partitions.

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


# ------------------------------------------------------------------------------------------------ one measured case
def load_workload(sim, scenarios, workload, tiles=1):
    if workload == "dam_break":
        sim.load_text(scenarios.stacked(scenarios.dam_break(), tiles), upscale=True)
    elif workload == "waterfall":
        sim.load_text(scenarios.stacked(scenarios.waterfall(), tiles), upscale=True)
    else:
        sim.load_half_tank(tiles)      # `tiles` closed tanks on top of each other (weak scaling: one per row slab)


def preroll_into_solves(sim, max_preroll, saturate=False):
    """untimed: advance to the first frame whose substeps run PCG iterations at all (a dam break first falls freely for ~22
    frames: zero divergence, the reference's all_zero(r) test skips the solve, main.c:742).
    saturate (the half tank, tol 0): go on until two frames in a row take the reference's maximum of 8 CFL substeps, at most 16
    frames.  From rest the tank's velocities are the rounding noise of unconverged solves; it grows for ~7 frames, during which a
    frame takes 1, 2, 3, 5, 7 ... substeps - a ramp whose shape depends on every summation order (mode, run length, number of
    ranks).  Timing frames of the ramp would make cells*steps/s a lottery; in the saturated phase a frame is 8 substeps."""
    n = 0
    while n < max_preroll:
        sim.step()
        n += 1
        if sim.stats().last_pcg_iterations >= 100:
            break
    full = 0
    while saturate and n < min(max_preroll, 16) and full < 2:
        sim.step()
        n += 1
        full = full + 1 if sim.stats().last_substeps >= 8 else 0
    return n


def balanced_partition(weights, world):
    """contiguous band ranges [lo, hi) per rank, at least one band each, whose weights are as even as a prefix split gets"""
    nb, total = len(weights), float(sum(weights))
    cuts, acc, b = [0], 0.0, 0
    for r in range(1, world):
        target = total * r / world
        lo_min, hi_max = cuts[-1] + 1, nb - (world - r)
        while b < hi_max and (b < lo_min or acc + 0.5 * weights[b] < target):
            acc += weights[b]
            b += 1
        cuts.append(b)
    cuts.append(nb)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def make_handle(ctx, GX, GY, workload, tiles, slab_arg, precond, tol):
    """create a handle (the whole grid, or this rank's slab), attach the exchanges, load the workload -> (sim, comm, p2p, HBM bytes)"""
    import torch
    args, ea, rank = ctx["args"], ctx["ea"], ctx["rank"]
    sm = ea.Simulation(GX, GY, device=ctx["local_rank"], dot_mode=ctx["dot_mode"], precond=ctx["PC"][precond], tile_records=args.tile_records, tol=tol,
                       slab=slab_arg, max_iterations=args.max_iterations, pcg_poll_interval=8)
    hbm = sm.hbm_bytes()      # what THIS handle allocated (free-memory differences are confounded when ranks share a device)
    cm, p2p = None, False
    if ctx["sharded"]:
        from euler_amd.slab import SLAB_EXACT, SLAB_LOCAL, RcclComm, TorchComm, attach_p2p
        coupling = SLAB_EXACT if args.slab == "exact" else SLAB_LOCAL
        if args.comm == "rccl":
            from euler_amd.slab import RcclUnavailable
            try:
                cm = RcclComm(sm, coupling)
            except RcclUnavailable as e:          # raised on every rank alike: the job goes on over torch.distributed, and says so
                if rank == 0:
                    print("bench: %s; exchanges fall back to torch.distributed callbacks" % e, file=sys.stderr)
                args.comm = "torch"
        if args.comm == "torch":
            cm = TorchComm(sm, coupling)
        p2p = args.p2p and attach_p2p(sm)
        if rank == 0 and args.p2p and not p2p and not ctx["comm_note"]:
            ctx["comm_note"].append(1)
            print("bench: peer-to-peer mailboxes unavailable (%s); exchanges stay on %s" % (sm._p2p_error, args.comm), file=sys.stderr)
    load_workload(sm, ctx["scenarios"], workload, tiles)
    return sm, cm, p2p, hbm


def pilot_partition(ctx, GX, GY, workload, tiles, saturate):
    """ONE picture over all ranks: even row slabs leave the ranks above (or below) the water with air.  A pilot pass with even
    slabs runs the untimed preroll, every rank reports the fluid cells of its bands, and the timed pass is created with band
    ranges that balance them (euler_config.slab_band_lo / hi) and pre-rolled by the same number of frames.
    -> (band ranges per rank, preroll frames)"""
    args, ea, grp, rank, world = ctx["args"], ctx["ea"], ctx["grp"], ctx["rank"], ctx["world"]
    pilot, pcomm, _, _ = make_handle(ctx, GX, GY, workload, tiles, (rank, world), args.precond, ctx["tol"] if workload == args.workload else None)
    preroll = preroll_into_solves(pilot, args.max_preroll, saturate)
    r0, r1 = pilot.slab_rows()
    fl_rows = (pilot.get(ea.F_COUNT) > 0).sum(axis=1)                # own rows
    mine = [[(r0 + k) // 64, int(fl_rows[k:k + 64].sum())] for k in range(0, r1 - r0, 64)]
    if pcomm is not None and getattr(pcomm, "error", None):
        raise RuntimeError(pcomm.error)
    pilot.close()
    del pilot, pcomm
    allb = [None] * world
    grp.dist.all_gather_object(allb, mine)
    nb = (GY + 63) // 64
    weights = [0.02 * 64 * GX] * nb                                  # an air cell costs a few dozen bytes per substep, a fluid cell ~9 KB
    for lst in allb:
        for bnd, cnt in lst:
            weights[bnd] += cnt
    return balanced_partition(weights, world), preroll


def rank_balance(ctx, sim, partition):
    import numpy as np  # noqa: F401
    grp, world = ctx["grp"], ctx["world"]
    per_rank = [None] * world
    grp.dist.all_gather_object(per_rank, [int(sim.stats().fluid_cells), list(sim.slab_rows())])
    fl = [p[0] for p in per_rank]
    return {"partition": "fluid-balanced band ranges (from a pilot pass with even slabs)" if partition else "even rows",
            "rows_per_rank": [p[1] for p in per_rank], "fluid_cells_per_rank": fl,
            "max_over_mean": round(max(fl) / max(sum(fl) / len(fl), 1.0), 3)}
