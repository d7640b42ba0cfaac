"""bench_blocks.studies - the secondary blocks of bench_full.json: equal-residual scans, parity against the reference's preconditioner, what 100 iterations are worth, the converged frames, the strong-scaling point.

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
from bench_blocks.bytes import ITER_BYTES, MODE_NAME  # noqa: F401
from bench_blocks.cpu import oracle_from_sim  # noqa: F401
from bench_blocks.timing import summarize, time_frames  # noqa: F401
from bench_blocks.workload import load_workload, make_handle, pilot_partition, preroll_into_solves, rank_balance  # noqa: F401


def strong_block(ctx, size, steps, tile_w):
    """BASELINE configs[3] (SURVEY 8d Config 4): ONE size x size dam break, main.c:843-900 per substep, the unit of the north star's
    strong-scaling target.  N > 1: split into fluid-balanced row slabs (collective: every rank calls this); N = 1: the same scenario
    on one GPU, the curve's denominator.  Timed like the headline (barrier + sync on both sides, MAX over ranks).  Rank 0 gets the
    block, the others None."""
    args, ea, grp, rank, world = ctx["args"], ctx["ea"], ctx["grp"], ctx["rank"], ctx["world"]
    multi = ctx["sharded"] and world > 1
    t_setup = time.perf_counter()
    if multi:
        partition, preroll = pilot_partition(ctx, size, size, "dam_break", 1, False)
        sim, comm, p2p_on, hbm = make_handle(ctx, size, size, "dam_break", 1, (rank, world, partition[rank][0], partition[rank][1]), args.precond, None)
        for _ in range(preroll):
            sim.step()
    else:
        partition = None
        sim, comm, p2p_on, hbm = make_handle(dict(ctx, sharded=False), size, size, "dam_break", 1, None, args.precond, None)
        preroll = preroll_into_solves(sim, args.max_preroll, False)
    setup_s = time.perf_counter() - t_setup
    t = time_frames(sim, ea, grp, args, args.precond, steps, 0, 1, True)
    if comm is not None and getattr(comm, "error", None):
        raise RuntimeError(comm.error)
    balance = rank_balance(ctx, sim, partition) if multi else None
    # the same job with every solve run to the reference's tolerance (multilevel mode, cap lifted): what "a 16384^2 dam break, simulated" costs per frame
    conv = None
    if args.precond in ("ic0_tile", "ic0_tile_mg") and not p2p_on:      # (the multilevel mode runs on the default transport, not over the mailboxes)
        try:
            sim.set_precond(ea.PRECOND_IC0_TILE_MG, args.tile_records)
            sim.set_solver(20000, 1e-6)
            sim.step()
            st0 = sim.stats()
            el = grp.timed(sim.step, 1)
            st1 = sim.stats()
            nsub = int(st1.total_substeps - st0.total_substeps)
            conv = {"mode": MODE_NAME["ic0_tile_mg"] % tile_w, "value": size * size / el, "unit": "cells*steps/s", "ms_per_step": 1e3 * el, "steps": 1, "tol": 1e-6,
                    "substeps": nsub, "pcg_iterations": int(st1.total_pcg_iterations - st0.total_pcg_iterations),
                    "last_residual": float(st1.last_residual)}
            if not args.no_kernel_timing:      # the NEXT frame with every kernel class bracketed (an event pair per launch costs the frame a few per cent: not the timed one):
                sim.profile_reset()            # how a converged frame divides into iterations and the stages around them
                sim.profile_enable(ea.profile_class_names())
                el_p = grp.timed(sim.step, 1)
                st2 = sim.stats()
                prof = sim.profile()
                sim.profile_enable([])
                nsub_p = int(st2.total_substeps - st1.total_substeps)
                if prof and nsub_p:
                    pcg = sum(ms for k, (ms, n) in prof.items() if k in ITER_BYTES["ic0_tile_mg"] and n)
                    rest = {k: round(ms / nsub_p, 3) for k, (ms, n) in prof.items() if k not in ITER_BYTES["ic0_tile_mg"] and n}
                    conv["stages"] = {"pcg_ms_per_substep": round(pcg / nsub_p, 3), "non_pcg_ms_per_substep": round(sum(rest.values()), 3),
                                      "non_pcg_share_of_kernel_time": round(sum(rest.values()) / max(sum(rest.values()) + pcg / nsub_p, 1e-9), 3),
                                      "non_pcg_share_of_frame": round(sum(rest.values()) * nsub_p / max(1e3 * el_p, 1e-9), 3),      # (of THAT frame's wall time: the iterations, these stages, the gaps)
                                      "bracketed_frame_ms": round(1e3 * el_p, 2),
                                      "ms_per_substep": dict(sorted(rest.items(), key=lambda kv: -kv[1])[:8]),
                                      "note": "kernel time by class in the frame BEHIND the timed one (every launch bracketed by a HIP event pair)"}
        except Exception as e:      # (collective: a failure here is every rank's)
            conv = {"error": repr(e)}
    out = None
    if rank == 0:
        rank_cells = None
        if multi:
            r0, r1 = sim.slab_rows()
            rank_cells = size * (r1 - r0)
        blk = summarize(t, size, size, args.precond, tile_w, None, None, steps, rank_cells=rank_cells)
        out = {k: blk[k] for k in ("mode", "value", "unit", "ms_per_step", "substeps", "pcg_iterations", "cells_substeps_per_s", "fluid_cells",
                                   "markers", "last_residual", "roofline", "pcg_iteration", "stages")}
        out.update({"workload": "%dx%d dam break (block layout upscaled; BASELINE configs[3]), %d timed frames after %d preroll frames "
                                "(into the phase where every substep runs PCG to the iteration cap)" % (size, size, steps, preroll),
                    "n_gpus": world if multi else 1, "scaling": "strong", "balance": balance, "hbm_bytes_this_rank": int(hbm),
                    "setup_and_preroll_seconds": round(setup_s, 1), "converged_frames_multilevel": conv,
                    "note": ("rank 0's kernels cover its slab; `value` is the whole job" if multi else
                             "one GPU: the denominator of the strong-scaling curve (run `bench.py --gpus N` for the N-GPU points of the same scenario)")})
    sim.close()
    del sim
    return out


def equal_residual_scan(sim, ea, tile_records, limit=1200, two_level=False, solver_tol=-1.0):
    """ONE pressure system - the stages of a substep up to project(), main.c:855-889, run once; project() reads utmp / vtmp / the
    cell grid and can be repeated - solved with the reference's IC(0) and the reference's budget of 100 iterations (main.c:735):
    its residual is the bar.  Then the tile-local mode gets the smallest budget (steps of 4) whose residual on the SAME system is at
    or below that bar.  Leaves the handle mid-substep (the caller goes on with whole frames), in the tile-local mode, budget 100."""
    dt = sim.timestep(0.1)
    for st in (ea.STAGE_ADVECT_MARKERS, ea.STAGE_REFRESH_COUNTS, ea.STAGE_SOURCES, ea.STAGE_EXTRAPOLATE, ea.STAGE_ADVECT_VELOCITY):
        sim.stage(st, dt)

    def solve(precond, budget):
        sim.set_precond(precond, tile_records)
        sim.set_solver(budget)              # (synchronises the handle's stream; euler_stage returns with its work done)
        t0 = time.perf_counter()
        sim.stage(ea.STAGE_PROJECT, dt)
        st = sim.stats()
        return {"ms": round(1e3 * (time.perf_counter() - t0), 2), "iterations": int(st.last_pcg_iterations), "residual": float(st.last_residual)}

    solve(ea.PRECOND_IC0, 100)                      # (untimed: first launches of the sweep kernels on this handle)
    exact = solve(ea.PRECOND_IC0, 100)
    tile100 = solve(ea.PRECOND_IC0_TILE, 100)
    budget, tile = 100, tile100
    scan = [[100, tile100["residual"]]]
    while tile["residual"] > exact["residual"] and budget < limit and exact["iterations"] >= 100:
        budget = budget + 4 if budget < 160 else int(budget * 1.06) // 4 * 4 + 4      # (the inf-norm residual of CG is not monotone: a scan, not a bisection)
        tile = solve(ea.PRECOND_IC0_TILE, budget)
        scan.append([budget, tile["residual"]])
    two = None
    multi = None

    def coarse_scan(precond):      # the same bar for a mode with a coarse correction: budgets from 8 up (it needs fewer than the reference's 100)
        out = {"scan": []}
        b2 = 4
        while True:
            b2 += 4 if b2 < 160 else 16
            t2 = solve(precond, b2)
            out["scan"].append([b2, t2["residual"]])
            if t2["residual"] <= exact["residual"] or b2 >= limit or exact["iterations"] < 100:
                break
        ok2 = t2["residual"] <= exact["residual"]
        out.update({"budget_for_equal_residual": b2 if ok2 else None, "at_that_budget": t2, "at_100_iterations": solve(precond, 100),
                    "solve_speedup_at_equal_residual": round(exact["ms"] / t2["ms"], 2) if ok2 else None})
        out["scan"] = out["scan"][::max(1, len(out["scan"]) // 16)] + out["scan"][-1:]
        return out

    if two_level:
        two = coarse_scan(ea.PRECOND_IC0_TILE2)
        multi = coarse_scan(ea.PRECOND_IC0_TILE_MG)
        sim.set_precond(ea.PRECOND_IC0_TILE, tile_records)
    errors = None
    if two_level and two is not None:
        # the residual's inf-norm is a noisy yardstick: the same budgets judged by the ERROR of the pressure against the converged solution
        # of this system (two-level mode to the reference's tolerance 1e-6, cap lifted): ||p_k - p*||_2 / ||p*||_2
        try:
            import numpy as np
            sim.set_precond(ea.PRECOND_IC0_TILE_MG, tile_records)
            sim.set_solver(20000, 1e-6)
            sim.stage(ea.STAGE_PROJECT, dt)
            st = sim.stats()
            pstar = sim.get(ea.F_PRESSURE).astype(np.float64)
            nstar = float(np.sqrt((pstar * pstar).sum()))
            errors = {"converged": {"iterations": int(st.last_pcg_iterations), "residual": float(st.last_residual)}, "unit": "||p - p*||_2 / ||p*||_2"}
            sim.set_solver(100, solver_tol)

            def err(precond, budget):
                solve(precond, budget)
                d = sim.get(ea.F_PRESSURE).astype(np.float64) - pstar
                return float(np.sqrt((d * d).sum()) / nstar)

            errors["reference_ic0_100"] = err(ea.PRECOND_IC0, 100)
            errors["tile_100"] = err(ea.PRECOND_IC0_TILE, 100)
            if budget != 100:
                errors["tile_%d" % budget] = err(ea.PRECOND_IC0_TILE, budget)
            errors["two_level_100"] = err(ea.PRECOND_IC0_TILE2, 100)
            b2 = two.get("budget_for_equal_residual")
            if b2 and b2 != 100:
                errors["two_level_%d" % b2] = err(ea.PRECOND_IC0_TILE2, b2)
            # the smallest two-level budget whose error is at or below the reference's after its 100 iterations
            k, ek = 8, None
            while k < 400:
                ek = err(ea.PRECOND_IC0_TILE2, k)
                if ek <= errors["reference_ic0_100"]:
                    break
                k += 4 if k < 64 else 16
            errors["two_level_budget_for_equal_error"] = k if ek is not None and ek <= errors["reference_ic0_100"] else None
            errors["two_level_at_that_budget"] = ek
            errors["multilevel_100"] = err(ea.PRECOND_IC0_TILE_MG, 100)
            bm = multi.get("budget_for_equal_residual") if multi else None
            if bm and bm != 100:
                errors["multilevel_%d" % bm] = err(ea.PRECOND_IC0_TILE_MG, bm)
            del pstar
        except Exception as e:
            errors = {"error": repr(e)}
        sim.set_precond(ea.PRECOND_IC0_TILE, tile_records)
    sim.set_solver(100, solver_tol)
    reached = tile["residual"] <= exact["residual"]
    return {"dt": dt, "reference_ic0_100_iterations": exact, "tile_100_iterations": tile100, "two_level": two, "multilevel": multi, "pressure_error_vs_converged": errors,
            "tile_budget_for_equal_residual": budget if reached else None, "tile_at_that_budget": tile,
            "solve_speedup_at_equal_residual": round(exact["ms"] / tile["ms"], 2) if reached else None, "residual_scan": scan[::max(1, len(scan) // 16)] + scan[-1:]}


def equal_residual(sim, ea, grp, args, GX, GY, tile_w, solver_tol):
    """Reference-quality throughput of the roofline mode on the headline workload: equal_residual_scan on the state the timed frames
    left (the saturated tank), then frames timed with the budget it found: cells*steps/s at equal residual."""
    out = equal_residual_scan(sim, ea, args.tile_records, two_level=True, solver_tol=solver_tol)
    out["system"] = "%dx%d %s, the state behind the timed frames, one substep's pressure system (dt %.3g)" % (GX, GY, args.workload, out.pop("dt"))

    def frames(precond, budget):
        sim.set_precond(precond, args.tile_records)
        sim.set_solver(budget)
        sim.step()
        k = max(1, args.steps // 2)
        st0 = sim.stats()
        el = grp.timed(sim.step, k)
        st1 = sim.stats()
        return {"value": GX * GY * k / el, "unit": "cells*steps/s", "ms_per_step": 1e3 * el / k, "steps": k,
                "substeps": int(st1.total_substeps - st0.total_substeps),
                "pcg_iterations": int(st1.total_pcg_iterations - st0.total_pcg_iterations),
                "cells_substeps_per_s": GX * GY * (st1.total_substeps - st0.total_substeps) / el}

    # frames with that budget: the one number for "reference-quality throughput", per mode
    if out["tile_budget_for_equal_residual"]:
        out["frames_at_that_budget"] = frames(ea.PRECOND_IC0_TILE, out["tile_budget_for_equal_residual"])
    for key, pc in (("two_level", ea.PRECOND_IC0_TILE2), ("multilevel", ea.PRECOND_IC0_TILE_MG)):
        blk = out.get(key)
        if blk and blk.get("budget_for_equal_residual"):
            blk["frames_at_that_budget"] = frames(pc, blk["budget_for_equal_residual"])
    mg = out.get("multilevel")
    if mg is not None:      # and the thing the reference cannot do at this size at all: frames whose solves reach its tolerance 1e-6
        try:
            sim.set_precond(ea.PRECOND_IC0_TILE_MG, args.tile_records)
            sim.set_solver(20000, 1e-6)
            sim.step()
            st0 = sim.stats()
            el = grp.timed(sim.step, 1)
            st1 = sim.stats()
            mg["converged_frames"] = {"value": GX * GY / el, "unit": "cells*steps/s", "ms_per_step": 1e3 * el, "steps": 1, "tol": 1e-6,
                                      "substeps": int(st1.total_substeps - st0.total_substeps),
                                      "pcg_iterations": int(st1.total_pcg_iterations - st0.total_pcg_iterations),
                                      "last_residual": float(st1.last_residual)}
        except Exception as e:
            mg["converged_frames"] = {"error": repr(e)}
        sim.set_solver(100, solver_tol)
    sim.set_precond(ea.PRECOND_IC0_TILE, args.tile_records)
    sim.set_solver(100)
    return out


# (name, N, workload, preroll to the first capped solve?, further frames before the state is taken)
PARITY_CASES = (("1024x1024 dam break, first frame whose solves run into the cap (BASELINE configs[1]; the block is in free fall: p ~ 0)", 1024, "dam_break", 400, 0),
                ("1024x1024 dam break, 60 frames later (the water has hit the floor: real pressures)", 1024, "dam_break", 400, 60),
                ("2048x2048 half tank from rest (configs[2] at 1/16 of its cells)", 2048, "half_tank", 0, 0),
                ("1024x1024 waterfall (configs[4] at 1/16 of its cells)", 1024, "waterfall", 30, 40))


def parity_vs_reference(ea, scenarios, so, device, dot_mode, tile_records, cases=PARITY_CASES):
    """What the roofline mode's fields are worth against the REFERENCE's preconditioner: from ONE state per BASELINE workload, one
    frame in the tile-local mode on the GPU and one frame with the reference's IC(0) on the oracle (CPU restatement, pinned to
    the compiled reference).  Where the solves converge the two agree to solver tolerance; where they run into the reference's
    100-iteration cap (main.c:735) both are unconverged and differ by what the last iterations would still have moved."""
    import numpy as np
    out = []
    for name, n, workload, preroll, more in cases:
        # (the state is reached in the roofline mode - any state will do, and it gets there several times sooner)
        sim = ea.Simulation(n, n, device=device, dot_mode=dot_mode, precond=ea.PRECOND_IC0_TILE, tile_records=tile_records)
        load_workload(sim, scenarios, workload, 1)
        pre = preroll_into_solves(sim, preroll) if preroll else 0
        for _ in range(more):
            sim.step()
        pre += more
        snap = os.path.join(tempfile.mkdtemp(prefix="euler_bench_"), "state.bin")
        sim.save_state(snap)      # (the same state again below, for the frames in other modes)
        o = oracle_from_sim(sim, ea, so)
        t0 = time.perf_counter()
        o.step()
        cpu_s = time.perf_counter() - t0
        sim.set_precond(ea.PRECOND_IC0_TILE, tile_records)
        sim.step()
        st = sim.stats()
        tile_uv = (sim.get(ea.F_U), sim.get(ea.F_V), sim.get(ea.F_COUNT) > 0)
        pr = o.p
        pmax = float(np.abs(pr).max())
        gfl, ofl = sim.get(ea.F_COUNT) > 0, o.count > 0
        e = {"state": "%s, after %d frames" % (name, pre),
             "substeps": [int(st.last_substeps), int(o.c.last_substeps)], "pcg_iterations": [int(st.last_pcg_iterations), int(o.c.last_pcg_iterations)],
             "capped": bool(o.c.last_pcg_iterations >= 100 * o.c.last_substeps),
             "residual_last_solve": [float(st.last_residual), float(o.c.last_residual)],
             "max_abs_du": float(np.abs(sim.get(ea.F_U) - o.u).max()), "max_abs_dv": float(np.abs(sim.get(ea.F_V) - o.v).max()),
             "max_abs_velocity": float(max(np.abs(o.u).max(), np.abs(o.v).max())),
             "dp_over_max_p": float(np.abs(sim.get(ea.F_PRESSURE) - pr).max() / pmax) if pmax > 0 else 0.0, "max_p": pmax,
             "fluid_cells": int(ofl.sum()), "fluid_cells_differing": int((gfl != ofl).sum()),
             "markers": [int(st.n_markers), int(o.n_markers)], "oracle_seconds": round(cpu_s, 2),
             "order": "[GPU tile-local mode, oracle with the reference's IC(0)]"}
        try:      # ... and what the tile-local mode needs to match the reference's residual on the NEXT substep's system of this state
            sc = equal_residual_scan(sim, ea, tile_records)
            e["next_system"] = {"residual_reference_ic0_100": sc["reference_ic0_100_iterations"]["residual"],
                                "residual_tile_100": sc["tile_100_iterations"]["residual"],
                                "tile_budget_for_equal_residual": sc["tile_budget_for_equal_residual"],
                                "residual_scan": sc["residual_scan"],
                                "solve_ms": [sc["reference_ic0_100_iterations"]["ms"], sc["tile_at_that_budget"]["ms"]],
                                "solve_speedup_at_equal_residual": sc["solve_speedup_at_equal_residual"]}
        except Exception as ex:
            e["next_system"] = {"error": repr(ex)}
        try:      # ... and all three against the CONVERGED frame of the same state (multilevel mode, cap lifted: every solve to the reference's tolerance)
            def frame(precond, cap):
                sim.load_state(snap)
                sim.set_precond(precond, tile_records)
                sim.set_solver(cap, 1e-6)
                t0 = time.perf_counter()
                sim.step()
                s2 = sim.stats()
                return sim.get(ea.F_U), sim.get(ea.F_V), sim.get(ea.F_COUNT) > 0, s2, time.perf_counter() - t0
            us, vs, fs, sst, secs = frame(ea.PRECOND_IC0_TILE_MG, 20000)
            um, vm, fm, mst, msecs = frame(ea.PRECOND_IC0_TILE_MG, 100)

            def dist(u, v, f):
                return {"max_abs_du": float(np.abs(u - us).max()), "max_abs_dv": float(np.abs(v - vs).max()), "fluid_cells_differing": int((f != fs).sum())}
            e["against_converged"] = {"converged": {"substeps": int(sst.last_substeps), "pcg_iterations": int(sst.last_pcg_iterations), "residual_last_solve": float(sst.last_residual),
                                                    "frame_seconds": round(secs, 3)},
                                      "reference_ic0_cap_100": dist(o.u, o.v, ofl), "tile_local_cap_100": dist(*tile_uv),
                                      "multilevel_cap_100": dict(dist(um, vm, fm), frame_seconds=round(msecs, 3), pcg_iterations=int(mst.last_pcg_iterations))}
        except Exception as ex:
            e["against_converged"] = {"error": repr(ex)}
        try:
            os.remove(snap); os.rmdir(os.path.dirname(snap))
        except OSError:
            pass
        out.append(e)
        o.close()
        sim.close()
        del sim
    return out


# ------------------------------------------------------------------------------------------------ quality of the headline's speed (default run: summaries)
def quality_summary(sim, ea, tile_records, solver_tol):
    """ONE pressure system - the state behind the timed frames, the stages of a substep up to project() (main.c:855-889) run once; project()
    reads utmp / vtmp / the cell grid and can be repeated - solved with each preconditioner under the reference's budget of 100 iterations
    (main.c:735): residual, time, and the ERROR of the pressure against the converged solution of the same system (multilevel mode to the
    reference's tolerance 1e-6, cap lifted).  Leaves the handle mid-substep (the caller goes on with whole frames)."""
    import numpy as np
    dt = sim.timestep(0.1)
    for st in (ea.STAGE_ADVECT_MARKERS, ea.STAGE_REFRESH_COUNTS, ea.STAGE_SOURCES, ea.STAGE_EXTRAPOLATE, ea.STAGE_ADVECT_VELOCITY):
        sim.stage(st, dt)

    def solve(precond, budget, tol=solver_tol):
        sim.set_precond(precond, tile_records)
        sim.set_solver(budget, tol)
        t0 = time.perf_counter()
        sim.stage(ea.STAGE_PROJECT, dt)
        st = sim.stats()
        return {"ms": round(1e3 * (time.perf_counter() - t0), 2), "iterations": int(st.last_pcg_iterations), "residual": float(st.last_residual)}

    solve(ea.PRECOND_IC0_TILE_MG, 8)      # (untimed: first launches / allocations of the coarse levels on this handle)
    conv = solve(ea.PRECOND_IC0_TILE_MG, 20000, 1e-6)
    pstar = sim.get(ea.F_PRESSURE).astype(np.float64)
    nstar = float(np.sqrt((pstar * pstar).sum())) or 1.0
    out = {"system": "one substep's pressure system of the state behind the timed frames (dt %.3g)" % dt, "budget": 100,
           "converged": dict(conv, mode="ic0_tile_mg", tol=1e-6), "error_unit": "||p - p*||_2 / ||p*||_2 against the converged solution p*", "modes": {}}
    solve(ea.PRECOND_IC0, 4)              # (untimed: first launches of the sweep kernels)
    for name, pc in (("ic0", ea.PRECOND_IC0), ("ic0_tile", ea.PRECOND_IC0_TILE), ("ic0_tile2", ea.PRECOND_IC0_TILE2), ("ic0_tile_mg", ea.PRECOND_IC0_TILE_MG)):
        r = solve(pc, 100)
        d = sim.get(ea.F_PRESSURE).astype(np.float64) - pstar
        r["pressure_error"] = float(np.sqrt((d * d).sum()) / nstar)
        out["modes"][name] = r
    del pstar
    sim.set_precond(ea.PRECOND_IC0_TILE, tile_records)
    sim.set_solver(100, solver_tol)
    return out


def converged_block(sim, ea, grp, args, GX, GY, tile_w, steps, solver_tol, traffic=None, traffic_note=None):
    """The headline workload with EVERY solve run to the reference's tolerance 1e-6 (main.c:736) - the multilevel mode, iteration cap lifted: what
    "this grid, actually solved" costs.  Timed like the headline, per-kernel HIP events in the timed region, its own roofline object."""
    sim.set_precond(ea.PRECOND_IC0_TILE_MG, args.tile_records)
    sim.set_solver(20000, 1e-6)
    sim.step()      # (untimed: the tank calms down from the capped frames' noise; allocations of the coarse levels)
    t = time_frames(sim, ea, grp, args, "ic0_tile_mg", steps, 0, 1, True)
    blk = summarize(t, GX, GY, "ic0_tile_mg", tile_w, traffic, traffic_note, steps)
    out = {k: blk[k] for k in ("mode", "value", "unit", "ms_per_step", "substeps", "pcg_iterations", "cells_substeps_per_s", "fluid_cells", "last_residual",
                               "roofline", "pcg_iteration", "kernels", "stages")}
    out.update({"steps": steps, "tol": 1e-6, "max_iterations": 20000,
                "iterations_per_solve": round(blk["pcg_iterations"] / max(blk["substeps"], 1), 1),
                "workload": "%dx%d %s, the frames behind the headline's, every solve converged" % (GX, GY, args.workload)})
    sim.set_precond(ea.PRECOND_IC0_TILE, args.tile_records)
    sim.set_solver(args.max_iterations, solver_tol)
    return out


def converged_deviation(ea, scenarios, so, device, dot_mode, tile_records, n=512, more=40):
    """How far the converged multilevel frame on the GPU is from the REFERENCE's algorithm run to convergence: from one state of the n x n dam
    break at impact, one frame on the GPU (multilevel mode, tol 1e-6, cap lifted) and one on the oracle with the reference's own IC(0), same
    tolerance, cap lifted (main.c:735 raised; nothing else changed).  Both converge to the same pressure, so the fields agree to solver tolerance."""
    import numpy as np
    sim = ea.Simulation(n, n, device=device, dot_mode=dot_mode, precond=ea.PRECOND_IC0_TILE_MG, tile_records=tile_records, max_iterations=20000, pcg_poll_interval=8)
    load_workload(sim, scenarios, "dam_break", 1)
    pre = preroll_into_solves(sim, 400)
    for _ in range(more):
        sim.step()
    o = oracle_from_sim(sim, ea, so)
    o.c.max_iterations = 20000
    t0 = time.perf_counter()
    o.step()
    cpu_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    sim.step()
    gpu_s = time.perf_counter() - t0
    st = sim.stats()
    gfl, ofl = sim.get(ea.F_COUNT) > 0, o.count > 0
    pmax = float(np.abs(o.p).max()) or 1.0
    out = {"state": "%dx%d dam break after %d frames (the water has hit the floor)" % (n, n, pre + more),
           "vs": "oracle with the reference's IC(0), tol 1e-6, cap lifted, from the same state",
           "substeps": [int(st.last_substeps), int(o.c.last_substeps)], "pcg_iterations": [int(st.last_pcg_iterations), int(o.c.last_pcg_iterations)],
           "max_abs_du": float(np.abs(sim.get(ea.F_U) - o.u).max()), "max_abs_dv": float(np.abs(sim.get(ea.F_V) - o.v).max()),
           "max_abs_velocity": float(max(np.abs(o.u).max(), np.abs(o.v).max())),
           "dp_over_max_p": float(np.abs(sim.get(ea.F_PRESSURE) - o.p).max() / pmax),
           "fluid_cells": int(ofl.sum()), "fluid_cells_differing": int((gfl != ofl).sum()),
           "frame_seconds": [round(gpu_s, 3), round(cpu_s, 2)], "order": "[GPU multilevel mode, oracle]"}
    o.close()
    sim.close()
    return out
