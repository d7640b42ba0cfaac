"""bench_blocks.timing - the timed region: K frames between barriers, HIP-event kernel classes, and the summary of one block (roofline object, per-iteration bytes and times).

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
from bench_blocks.bytes import HBM_PEAK_GBPS, ITER_BYTES, MODE_NAME, ONCE_PER_SOLVE_BYTES, PCG_BYTES, PCG_CLASSES, TILE_MODES, W  # noqa: F401
from bench_blocks.pmc import traffic_of  # noqa: F401


def kernel_rows(prof, precond, cells_fluid, traffic, iters):
    """per kernel class: launch time and the byte counts -> GB/s.  A class that runs once per iteration takes its bytes from ITER_BYTES[precond]."""
    rows = {}
    for name, (ms, launches) in prof.items():
        if not launches:
            continue
        e = {"ms_total": round(ms, 3), "launches": int(launches), "avg_us": round(1e3 * ms / launches, 2)}
        per_iteration = launches >= 0.5 * max(iters, 1)
        b = ITER_BYTES[precond].get(name) if per_iteration else ONCE_PER_SOLVE_BYTES.get(name)
        e["per_iteration"] = bool(per_iteration)
        if b is not None:
            sec = ms / launches * 1e-3
            e["bytes_per_cell"] = b
            e["GBps_active"] = round(b * cells_fluid / sec / 1e9, 1)
            t = traffic_of(traffic, name, precond)
            if t:
                e["traffic_bytes_per_launch"] = int(t)
                e["GBps_traffic"] = round(t / sec / 1e9, 1)
        rows[name] = e
    return rows


PROFILE_STRIDE = 7      # large grids: every 7th launch of the dominant class carries its event pair in the timed region (coprime with the p-update's period of 8 iterations)


def time_frames(sim, ea, grp, args, precond, steps, warmup_done, warmup, big):
    """the timed region + per-kernel HIP-event timing.  An event pair serialises the stream for a few microseconds: with every launch of every class bracketed the 8192^2
    headline ran 3.9 % slower than without (399.8 against 384.4 ms per frame).  So INSIDE the timed region only the dominant class is bracketed - on large grids every
    PROFILE_STRIDE-th launch of it (EULER_OPT_PROFILE_STRIDE: ~2300 samples over the default 20 frames, every phase of the iteration's 8-cycle alike; the resident solver's
    one launch per solve: every one) - and the other classes are timed in a second pass BEHIND it (large grids: two frames, every launch; their totals are scaled to the
    timed region's substeps, so averages stand and per-substep sums compare; small grids: as many frames as the timed region)."""
    for _ in range(max(warmup - warmup_done, 0)):
        sim.step()
    dominant = {"ic0": "backward_solve", "ic0_tile": "apply_a", "ic0_tile2": "apply_a", "ic0_tile_mg": "apply_a", "jacobi": "update_pr"}[precond]
    # round 6: on large grids EVERY kernel class is timed (second pass): the line then carries the stages around the solve as well (`stages`), not only the iteration's classes
    classes_all = ea.profile_class_names() if (args.profile_all or big) else PCG_CLASSES
    if not big and precond == "ic0_tile" and sim.resident_info()[0]:
        dominant = "resident_pcg"
    timed = [] if args.no_kernel_timing else [dominant]
    stride = PROFILE_STRIDE if (timed and dominant != "resident_pcg") else 1      # (the resident solver is ONE launch per solve)
    sim.profile_reset()
    sim.set_option(ea.OPT_PROFILE_STRIDE, stride)
    sim.profile_enable(timed)
    st0 = sim.stats()
    elapsed = grp.timed(sim.step, steps)
    st1 = sim.stats()
    prof = sim.profile() if timed else {}
    sim.profile_enable([])
    sim.set_option(ea.OPT_PROFILE_STRIDE, 1)
    if stride > 1 and dominant in prof:      # the sample stands for every launch of the class: the same average, the launch count of the region
        ms, n = prof[dominant]
        prof[dominant] = (ms * stride, n * stride)
    iters = st1.total_pcg_iterations - st0.total_pcg_iterations
    substeps = st1.total_substeps - st0.total_substeps
    iters2 = iters
    if not args.no_kernel_timing:
        steps2 = min(steps, 2) if big else steps
        sim.profile_reset()
        sim.profile_enable([k for k in classes_all if k != dominant])
        for _ in range(steps2):
            sim.step()
        st2 = sim.stats()
        iters2 = st2.total_pcg_iterations - st1.total_pcg_iterations
        f = (substeps / max(st2.total_substeps - st1.total_substeps, 1)) if big else 1.0
        for k, (ms, n) in sim.profile().items():
            prof.setdefault(k, (ms * f, n * f))
        sim.profile_enable([])
        if big:
            iters2 = iters      # (the second pass's counts are scaled to the timed region)
    # one iteration = the mode's per-iteration classes (ITER_BYTES); once-per-solve launches (s = z, k_finish_p, the factor) are not in it
    per_iter_ms = sum(prof[k][0] / prof[k][1] for k in ITER_BYTES[precond]
                      if k in prof and prof[k][1] >= 0.5 * max(iters if (big or k == dominant) else iters2, 1))
    resident = None
    if "resident_pcg" in prof and prof["resident_pcg"][1]:      # the solves ran as ONE persistent launch each (csrc/k_resident.hip): its time / the iterations it ran
        resident = dict(ms_total=prof["resident_pcg"][0], solves=int(prof["resident_pcg"][1]), iters=iters if (big or dominant == "resident_pcg") else iters2,
                        f32=bool(sim.cfg.pcg_precision))
        if "apply_a" not in prof:
            per_iter_ms = prof["resident_pcg"][0] / max(resident["iters"], 1)
    return dict(elapsed=elapsed, st0=st0, st1=st1, prof=prof, iters=iters, per_iter_ms=per_iter_ms, dominant=dominant,
                substeps=substeps, resident=resident)


def summarize(t, size_x, size_y, precond, tile_w, traffic, traffic_note, steps, fused_search=True, rank_cells=None, rank_fluid_share=1.0):
    """value + per-kernel rows + roofline object (dominant kernel = the per-iteration PCG class with the largest total time) + whole-iteration aggregate.
    The byte rates are THIS rank's: its kernels cover rank_cells cells (row slabs: the own rows, whose fluid cells the handle counts
    itself; band slabs of a replicated handle: 1 / world of the grid's fluid cells)."""
    cells_job = size_x * size_y
    cells = rank_cells if rank_cells else cells_job
    fluid = int(t["st1"].fluid_cells * rank_fluid_share)
    rows = kernel_rows(t["prof"], precond, fluid, traffic, t["iters"])
    per_iter = [k for k in rows if rows[k]["per_iteration"] and k in ITER_BYTES[precond]]
    roof = None
    if per_iter:
        dom = max(per_iter, key=lambda k: rows[k]["ms_total"])
        r = rows[dom]
        sec = r["avg_us"] * 1e-6
        b = r["bytes_per_cell"]
        tr = r.get("traffic_bytes_per_launch")
        active = b * fluid / sec / 1e9
        # `achieved` / `frac` = ALGORITHMIC bytes (SURVEY 8d's per-cell figure for this variant x the fluid cells one launch processes)
        # / average launch time; the PMC traffic (FETCH_SIZE counts Infinity-Cache hits too) stays beside it as traffic / frac_traffic
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(active, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(active / HBM_PEAK_GBPS, 4), "traffic": int(tr) if tr else None,
                "frac_traffic": round(tr / sec / 1e9 / HBM_PEAK_GBPS, 4) if tr else None,
                "traffic_over_algorithmic": round(tr / (b * fluid), 3) if tr else None,
                "frac_dense": round(b * cells / sec / 1e9 / HBM_PEAK_GBPS, 4),
                "achieved_is": "algorithmic bytes per cell x fluid cells of one launch / average launch time (HIP events in the timed region; large grids: around every 7th launch of the class)",
                "algorithmic_bytes_per_cell": b, "algorithmic_bytes_per_launch": int(b * fluid),
                "avg_launch_us": r["avg_us"], "launches": r["launches"], "fluid_fraction": round(fluid / cells, 4),
                "traffic_source": traffic_note,
                "note": "frac_dense counts ALL X*Y cells like the reference's dense loops and may exceed 1 on sparse scenes; the kernels "
                        "visit fluid cells only, so frac / frac_traffic are what the memory system did"}
    agg = None
    if t["per_iter_ms"] and per_iter:
        sec = t["per_iter_ms"] * 1e-3
        bpc = sum(rows[k]["bytes_per_cell"] for k in per_iter)
        # the accounting follows the kernels: the iteration's bytes are the sum over the classes that were timed once per iteration - and that is the mode's table
        assert abs(bpc - PCG_BYTES[precond]) < 1e-9 or set(per_iter) != set(ITER_BYTES[precond]), (bpc, PCG_BYTES[precond], per_iter)
        tsum = None
        if traffic:
            parts = [rows[k].get("traffic_bytes_per_launch") for k in per_iter if rows[k]["bytes_per_cell"] >= 1]
            tsum = sum(parts) if parts and all(parts) else None
        agg = {"us_per_iteration": round(1e3 * t["per_iter_ms"], 2), "bytes_per_cell_iteration": round(bpc, 2),
               "classes": {k: rows[k]["bytes_per_cell"] for k in per_iter},
               "complete": set(per_iter) == set(ITER_BYTES[precond]),      # every per-iteration class of the mode was timed
               "launches_per_iteration": len(per_iter),      # (coarse_cycle: one class, three launches in the multilevel mode)
               "GBps_active": round(bpc * fluid / sec / 1e9, 1), "frac_active": round(bpc * fluid / sec / 1e9 / HBM_PEAK_GBPS, 4),
               "GBps_traffic": round(tsum / sec / 1e9, 1) if tsum else None,
               "frac_traffic": round(tsum / sec / 1e9 / HBM_PEAK_GBPS, 4) if tsum else None,
               "frac_dense": round(bpc * cells / sec / 1e9 / HBM_PEAK_GBPS, 4)}
    if t.get("resident") and not per_iter:
        # the resident solver: r, s, p, E^-1 stay in registers / LDS for the whole solve - no HBM traffic inside it.  For comparison the figure the multi-kernel
        # form would need for the same time: its algorithmic bytes (67 B per cell and iteration in double, 34.5 in float) / this time
        rs = t["resident"]
        sec = rs["ms_total"] / max(rs["iters"], 1) * 1e-3
        w = 4 if rs.get("f32") else W
        bpc = 8.125 * w + 2
        agg = {"us_per_iteration": round(1e6 * sec, 2), "resident": True, "solves": rs["solves"], "launches_per_iteration": 0,
               "hbm_bytes_inside_the_solve": 0, "equivalent_bytes_per_cell_iteration": bpc,
               "GBps_active": round(bpc * fluid / sec / 1e9, 1), "frac_active": round(bpc * fluid / sec / 1e9 / HBM_PEAK_GBPS, 4),
               "note": "one persistent launch per solve, vectors in registers: bound by two grid-wide synchronisations per iteration, not by bytes; frac_active = what the "
                       "multi-kernel form's algorithmic bytes would need at this speed"}
        roof = {"bound": "hbm", "kernel": "resident_pcg", "achieved": agg["GBps_active"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": agg["frac_active"], "traffic": 0,
                "algorithmic_bytes_per_cell": bpc, "avg_launch_us": round(1e3 * rs["ms_total"] / max(rs["solves"], 1), 2), "launches": rs["solves"],
                "note": agg["note"]}
    stages = None
    if t["substeps"] and len(rows) > len(per_iter) + 2:      # every class was timed: the time per substep of the stages around the iterations
        ms = {k: round(rows[k]["ms_total"] / t["substeps"], 3) for k in rows if k not in ITER_BYTES[precond] and k != "resident_pcg"}
        stages = {"ms_per_substep": dict(sorted(ms.items(), key=lambda kv: -kv[1])), "non_pcg_ms_per_substep": round(sum(ms.values()), 3),
                  "pcg_ms_per_substep": round(sum(rows[k]["ms_total"] for k in per_iter) / t["substeps"], 3),
                  "note": "kernel time by class (HIP events: the dominant class sampled inside the timed region, the others in two frames behind it) / substeps; a solve's first k_apply_a and first k_precond_tile launch sit inside their per-iteration classes"}
    return {"mode": MODE_NAME[precond] % tile_w if precond in TILE_MODES else MODE_NAME[precond],
            "stages": stages,
            "value": cells_job * steps / t["elapsed"], "unit": "cells*steps/s", "ms_per_step": 1e3 * t["elapsed"] / steps,
            "substeps": int(t["substeps"]), "pcg_iterations": int(t["iters"]),
            "cells_substeps_per_s": cells_job * t["substeps"] / t["elapsed"],
            "fluid_cells": fluid, "markers": int(t["st1"].n_markers), "last_residual": float(t["st1"].last_residual),
            "roofline": roof, "pcg_iteration": agg, "kernels": rows}
