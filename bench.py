#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configurations.

metric    cells*steps/s of the whole sim_step() path (one step = one 0.1 s frame = up to 8 CFL substeps, each with a
          PCG pressure projection), plus the achieved HBM rate of the pressure solve against the MI355X roofline.

output    ONE compact JSON line on stdout (rank 0; < 8 KB: compact_line, tests/test_bench_line.py - round 3's 25.7 KB line came back unparsed from the driver):
          the contract's keys, `roofline`, `pcg_iteration`, `kernels`, `cpu_baseline`, the `converged` block and one-number summaries of the secondary blocks.
          The whole object goes to bench_full.json beside it (and under gpurun_out/).  Inputs are resident in HBM when a timed region starts.

headline  N=1: configs[2], the 8192x8192 half-filled tank, "pressure-solve roofline run" (SURVEY 8d config 3): tol = 0 and
          max_iterations = 100, so every substep runs exactly 100 PCG iterations; the timed frames lie in the tank's saturated
          phase (the reference's maximum of 8 CFL substeps per frame, preroll_into_solves).  It runs in the ROOFLINE MODE SURVEY 7 (hard
          part 1b) and 8d name: the tile-local IC(0) preconditioner (EULER_PRECOND_IC0_TILE, include/euler.h) - the reference's
          recurrences restricted to 64-row x 16-column blocks, one pass over memory per iteration - which is NOT the reference's
          sequence of iterates (same solution where PCG converges; tests/test_gpu_tile_precond.py).  Every number carries its mode.

roofline  `roofline.frac` / `achieved` = ALGORITHMIC bytes of the dominant launch (ITER_BYTES: bytes per fluid cell of every launch of an iteration, per mode) x the
          fluid cells it processes / its average launch time (HIP events inside the timed region).  `traffic` = HBM bytes rocprofv3's FETCH_SIZE / WRITE_SIZE
          counters saw for the kernel in a live PMC pass of this very workload (child processes of this run, calibrated on a copy of known size in the same
          pass), with `frac_traffic` / `traffic_over_algorithmic` beside it.  An iteration's bytes are the SUM of its launches' bytes (asserted in summarize()).

beside it `converged` - the same workload with EVERY solve run to the reference's tolerance (multilevel mode, cap lifted): its own roofline, the CPU at equal tolerance
          (the oracle's IC(0) to 1e-6), the deviation from the reference's algorithm run to convergence; `summary.exact_ic0` - the same frames in the parity mode (the
          reference's own IC(0), bit-identical iterates); `summary.quality_100_iterations` - what the reference's budget is worth in every mode against the converged
          solution; `summary.configs1_1024_dam_break` - BASELINE configs[1]: parity mode checked in-run against the oracle, the resident solver (one persistent launch
          per solve), its float variant ("fp32") and the multi-kernel form; `summary.projection_16384` with its own PMC pass; `summary.time_to_solution_2048_ms`;
          `summary.strong_16384_dam_break` - configs[3] on this many GPUs (N = 1: the strong-scaling denominator).  `--quality` adds the long studies (equal-residual
          budget scans, one frame per BASELINE workload against the reference's IC(0) on the oracle) to bench_full.json.
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench_blocks.bytes import HBM_PEAK_GBPS, W, _TILE, _RUPD, _SWEEP, ITER_BYTES, PCG_BYTES, set_as_stored, ONCE_PER_SOLVE_BYTES, PCG_CLASSES, KERNEL_OF_CLASS, MODE_NAME, TILE_MODES  # noqa: E402,F401
from bench_blocks.cpu import build_native_oracle, cpu_model, cpu_baseline_roofline_run, cpu_from_gpu_state, cpu_converged_baseline, oracle_from_sim  # noqa: E402,F401
from bench_blocks.pmc import pmc_live, traffic_of  # noqa: E402,F401
from bench_blocks.workload import load_workload, preroll_into_solves, balanced_partition, make_handle, pilot_partition, rank_balance  # noqa: E402,F401
from bench_blocks.timing import kernel_rows, time_frames, summarize  # noqa: E402,F401
from bench_blocks.studies import equal_residual_scan, equal_residual, PARITY_CASES, parity_vs_reference, quality_summary, converged_block, converged_deviation, strong_block  # noqa: E402,F401
from bench_blocks.line import LINE_LIMIT, _pick, _r, _short, ROOF_KEYS, ITER_KEYS, _block, compact_line, write_full  # noqa: E402,F401


set_as_stored(False)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=0, help="N of the NxN grid (default: configs[2] = 8192)")
    ap.add_argument("--workload", default="half_tank", choices=["dam_break", "half_tank", "waterfall"])
    ap.add_argument("--dot-mode", default="tree", choices=["tree", "sequential"])
    ap.add_argument("--precond", default="ic0_tile", choices=["ic0", "jacobi", "ic0_tile", "ic0_tile2", "ic0_tile_mg"],
                    help="ic0_tile = roofline mode (default), ic0 = parity mode (the reference's preconditioner), ic0_tile2 = roofline mode + coarse "
                         "correction (one GPU; fewer iterations to a given residual, docs/solver_two_level.md)")
    ap.add_argument("--tile-records", type=int, default=0)
    ap.add_argument("--max-iterations", type=int, default=100, help="PCG iteration cap per solve (the reference's: 100, main.c:735); lift it together with --tol 1e-6 "
                                                                    "to time frames whose solves converge (e.g. --precond ic0_tile_mg)")
    ap.add_argument("--tol", type=float, default=None, help="PCG tolerance (default: 0 for half_tank = the roofline run, else the reference's 1e-6)")
    ap.add_argument("--max-preroll", type=int, default=400)
    ap.add_argument("--partition", default="auto", choices=["auto", "even"],
                    help="row slabs of a strong-scaling run: auto = band ranges that balance the fluid (found by a pilot pass with even slabs), even = equal rows")
    ap.add_argument("--preroll", default="auto", choices=["auto", "solves"],
                    help="auto: the half tank is advanced into its saturated phase (8 substeps per frame) before the timed frames; solves: only to the first solve")
    ap.add_argument("--slab", default="rows", choices=["rows", "local", "exact", "replicas"],
                    help="N>1: rows = TRUE ROW SLABS for every stage (default; SURVEY 8e: each rank holds and steps only its rows and the "
                         "markers in them, ghost rows / marker migration / dt all-reduce between neighbours); local / exact = round 1's layout "
                         "(only the pressure solve is sharded, the cheap stages run replicated on the whole grid) with slab-local or exact IC(0) "
                         "coupling; replicas = independent copies.  The roofline mode has no preconditioner coupling between slabs at all.")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="N>1 exchange transport: the library's own RCCL communicator (C, no host code between kernels) "
                         "or the torch.distributed callbacks of euler_amd/slab.py")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = N x (N*gpus) grid, one tank per row slab (default; the driver's scaling run); strong = the "
                         "N x N grid of --size split into row slabs (BASELINE configs[3]: --size 16384 --scaling strong)")
    ap.add_argument("--p2p", action="store_true",
                    help="N>1: route the per-iteration PCG exchanges (scalar all-reduces, ghost rows of s) over the peer-to-peer IPC "
                         "mailboxes of csrc/comm_p2p.hip instead of RCCL (the default, as the north star names it); falls back to RCCL on "
                         "every rank if the mailboxes cannot be set up")
    ap.add_argument("--no-p2p", action="store_true", help=argparse.SUPPRESS)   # (round 1's spelling of the default)
    ap.add_argument("--grid-y-mult", type=int, default=0,
                    help="diagnostics: run the N x (N*M) grid of an M-GPU weak-scaling job on the GPUs given")
    ap.add_argument("--force-slab", action="store_true",
                    help="N=1 diagnostics: run the communicator code path with one rank (every exchange still goes through RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-size", type=int, default=2048, help="N of the N x N half tank the cpu_baseline leg times (tests use a small one)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the hipEvent per-kernel timing (used under rocprofv3)")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class (diagnostics)")
    ap.add_argument("--no-secondary", action="store_true", help="headline case only (no parity-mode / 1024^2 / 16384^2 / time-to-solution blocks)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 PMC passes (roofline.traffic is then null)")
    ap.add_argument("--no-16384", action="store_true", help="skip the 16384^2 projection block")
    ap.add_argument("--quality", action="store_true", help="also run the long quality studies (equal-residual budget scans, one frame per BASELINE workload against the "
                                                           "reference's IC(0) on the oracle); their results go to bench_full.json")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong_16384_dam_break block (BASELINE configs[3]; N = 1: its denominator)")
    ap.add_argument("--strong-size", type=int, default=16384, help="N of the strong block's N x N dam break (tests use a small one)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # this process runs under rocprofv3 --pmc
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # asked for N GPUs but started as a plain process: start the N ranks the way the driver does (a child
        # torch.distributed.run - nothing here has touched the GPU yet) and hand its one JSON line through
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        sys.stdout.write(r.stdout)
        sys.stdout.flush()
        sys.exit(r.returncode)

    N = args.size or 8192
    tol = args.tol if args.tol is not None else (0.0 if args.workload == "half_tank" else None)
    single = args.gpus == 1 and not args.force_slab and "RANK" not in os.environ
    child_common = ["--size", str(N), "--workload", args.workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-kernel-timing",
                    "--no-secondary", "--no-pmc", "--pmc-child", "--preroll", "solves", "--dot-mode", args.dot_mode, "--tile-records", str(args.tile_records)]
    if args.tol is not None:
        child_common += ["--tol", repr(args.tol)]

    # ---- live PMC passes FIRST (child processes; this process has not touched the GPU yet and holds no HBM)
    traffic = traffic_note = None
    traffic_exact = traffic_exact_note = None
    traffic_mg = traffic_mg_note = None
    t_start, t_pmc = time.perf_counter(), 0.0
    if single and not args.no_pmc and not args.pmc_child:
        t0 = time.perf_counter()
        traffic, traffic_note = pmc_live(child_common + ["--precond", args.precond])
        if not args.no_secondary and args.precond != "ic0":
            traffic_exact, traffic_exact_note = pmc_live(child_common + ["--precond", "ic0"])
        if not args.no_secondary and args.precond == "ic0_tile":      # the converged block's kernels (multilevel mode, every solve to 1e-6)
            mg_child = [a for a in child_common]
            if "--tol" in mg_child:
                del mg_child[mg_child.index("--tol"):mg_child.index("--tol") + 2]
            traffic_mg, traffic_mg_note = pmc_live(mg_child + ["--precond", "ic0_tile_mg", "--tol", "1e-6", "--max-iterations", "20000"])
        t_pmc = time.perf_counter() - t0
        print("bench: PMC passes took %.0f s (%s)" % (t_pmc, traffic_note), file=sys.stderr)

    # stdout carries exactly ONE line, the JSON: whatever native libraries print on fd 1 while the job runs
    # (RCCL writes its version banner there when a communicator is created) is diverted to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)
    from euler_amd.dist import Group, whole_job_rate
    grp = Group(force=args.force_slab)   # one process per GPU; "nccl" (= RCCL) when N > 1
    rank, local_rank, world = grp.rank, grp.local_rank, grp.world
    if world > 1:
        args.gpus = world
    import torch

    import euler_amd as ea
    from euler_amd import scenarios

    dot_mode = ea.DOT_TREE if args.dot_mode == "tree" else ea.DOT_SEQUENTIAL
    set_as_stored(args.dot_mode != "tree" or bool(args.p2p), 2 if args.p2p else 8)      # which launches this run makes (k_pcg.hip tile_recompute, p_steps; the handles keep euler_set_option's defaults)
    PC = {"ic0": ea.PRECOND_IC0, "jacobi": ea.PRECOND_JACOBI, "ic0_tile": ea.PRECOND_IC0_TILE, "ic0_tile2": ea.PRECOND_IC0_TILE2, "ic0_tile_mg": ea.PRECOND_IC0_TILE_MG}
    tile_w = args.tile_records or 16
    # N > 1 (weak scaling): the grid grows to N x (N * gpus) rows; the pressure solve is split into one
    # slab of N rows per GPU, the cheap stages run replicated (DESIGN.md "Multi-GPU")
    sharded = (world > 1 or args.force_slab) and args.slab != "replicas"
    GX, GY = N, N * (args.grid_y_mult if args.grid_y_mult > 0 else (world if sharded and args.scaling == "weak" else 1))
    if world > 1:
        torch.cuda.set_device(local_rank)
    rows = sharded and world > 1 and args.slab == "rows"      # true row slabs: this process holds its rows only
    tiles = GY // N
    saturate = args.workload == "half_tank" and args.preroll == "auto"
    comm_note = []

    ctx = dict(args=args, ea=ea, scenarios=scenarios, grp=grp, rank=rank, world=world, local_rank=local_rank, dot_mode=dot_mode, PC=PC,
               sharded=sharded, comm_note=comm_note, tol=tol)
    slab_arg, partition, preroll = ((rank, world) if rows else None), None, None
    if rows and args.scaling == "strong" and args.partition == "auto":
        partition, preroll = pilot_partition(ctx, GX, GY, args.workload, tiles, saturate)
        slab_arg = (rank, world, partition[rank][0], partition[rank][1])
    sim, comm, p2p_on, hbm_per_rank = make_handle(ctx, GX, GY, args.workload, tiles, slab_arg, args.precond, tol)
    if preroll is None:
        preroll = preroll_into_solves(sim, args.max_preroll, saturate)
    else:
        for _ in range(preroll):
            sim.step()
    if args.pmc_child:
        sim.copy_bandwidth(1 << 30, 2)     # the calibration launches of pmc_live(): a known 2^30 bytes read and written each

    big = GX * GY >= 4096 * 4096
    t = time_frames(sim, ea, grp, args, args.precond, args.steps, 0, args.warmup, big)
    if comm is not None and comm.error:
        raise RuntimeError(comm.error)
    cells = GX * GY
    job_rate = (cells * args.steps / t["elapsed"]) if sharded else whole_job_rate(float(cells), args.steps, t["elapsed"], grp)
    balance = rank_balance(ctx, sim, partition) if rows else None
    comm_calls = None
    if comm is not None:
        try:      # how many operations of each kind the communicator carried so far (preroll + warm-up + timed frames): with the fused
            # exchange a PCG iteration costs two `exchange` calls and nothing else
            comm_calls = dict(comm.counts)
        except Exception:
            comm_calls = None
    # the communicator as the job saw it: how many ranks the transport really connected, and the one unknown of the scaling model - the latency L of an
    # exchange point of the distributed PCG iteration, measured over THIS communicator (collective: every rank calls it; HIP-event time per call on rank 0's stream):
    # G1 = the slab's edge rows of z to the two neighbours + {max |r|, dot(z, r)} of every rank to every rank, G2 = the dot partial of every rank to every rank
    comm_info = None
    if comm is not None:
        comm_info = {"ranks": int(getattr(comm, "world", world)), "transport": "rccl" if args.comm == "rccl" else "torch.distributed", "rccl_version": getattr(comm, "version", None),
                     "p2p_mailboxes": bool(p2p_on)}
        if os.environ.get("EULER_SHARE_GPU"):      # (euler_amd/dist.py: more local ranks than devices, or a test that asked for it)
            comm_info["ranks_share_devices"] = True
        try:
            comm_info["exchange_us"] = {"g1_edge_rows_and_pair": round(sim.exchange_latency(100, GX, 2), 2), "g2_scalar": round(sim.exchange_latency(100, 0, 1), 2)}
            l = sum(comm_info["exchange_us"].values())
            comm_info["exchange_us"]["per_iteration"] = round(l, 2)
        except Exception as e:
            comm_info["exchange_us"] = {"error": repr(e)}
        if args.precond == "ic0_tile_mg":      # the multilevel cycle on row slabs: split by rows (a third exchange point) or replicated - and by which rule
            try:
                comm_info["mg_split_active"] = int(sim.get_option(ea.OPT_MG_SPLIT_ACTIVE))      # the gather level of the last solve, 0 = the cycle ran replicated
                comm_info["mg_split_rule"] = ("by size (EULER_OPT_MG_SPLIT_LEVEL = 0): split wherever a level of <= 16384 nodes lies above level 0 and every rank's zones reach "
                                              "into the next rank only; it trades the replicated level 0 and its all-gather for one more exchange point (exchange_us is the measured L)")
            except Exception as e:
                comm_info["mg_split_active"] = {"error": repr(e)}
    head = copy_gbps = device = quality = converged = exact = None
    tile_w_run = tile_w
    timings = {"pmc_passes": round(t_pmc, 1)}
    clock = [t_start + t_pmc]

    def lap(name):
        now = time.perf_counter()
        timings[name] = round(now - clock[0], 1)
        clock[0] = now
    lap("setup_preroll_headline_frames")
    solver_tol = 1e-6 if tol is None else tol
    if rank == 0:
        rank_cells, share = None, 1.0
        if rows:
            r0, r1 = sim.slab_rows()
            rank_cells = GX * (r1 - r0)
        elif sharded and world > 1:
            rank_cells, share = GX * GY // world, 1.0 / world
        head = summarize(t, GX, GY, args.precond, tile_w, traffic, traffic_note, args.steps, fused_search=True, rank_cells=rank_cells,
                         rank_fluid_share=share)
        try:      # the ceiling a plain device-to-device copy reaches on this very GPU (read + write), next to the 8 TB/s spec peak
            copy_gbps = round(sim.copy_bandwidth(1 << 30, 10), 1)
        except Exception:
            copy_gbps = None
        if head["roofline"]:
            head["roofline"]["measured_copy_GBps"] = copy_gbps
        device = sim.device_name()
    secondary = {}
    extras = single and not args.no_secondary and not args.pmc_child
    big_head = GX * GY >= 4096 * 4096
    if extras and args.precond == "ic0_tile":
        # the SAME handle goes on (no second preroll): the parity mode on the same frames, what 100 iterations are worth per mode, then the converged frames
        try:      # (1) the same workload in the parity mode: the reference's own IC(0), bit-identical iterates
            k2 = max(1, min(4, args.steps // 2))
            sim.set_precond(ea.PRECOND_IC0, args.tile_records)
            sim.step()
            t2 = time_frames(sim, ea, grp, args, "ic0", k2, 0, 0, big_head)
            exact = summarize(t2, GX, GY, "ic0", tile_w, traffic_exact, traffic_exact_note, k2)
            exact["workload"] = "%dx%d %s (the headline workload, the frames behind the headline's), %d frames" % (GX, GY, args.workload, k2)
            exact["steps"] = k2
        except Exception as e:
            exact = {"error": repr(e)}
        secondary["exact_ic0"] = exact
        sim.set_precond(ea.PRECOND_IC0_TILE, args.tile_records)
        lap("exact_ic0")
        try:      # (2) what the budget of 100 iterations is worth in every mode, against the converged solution of the same system
            quality = quality_summary(sim, ea, args.tile_records, solver_tol)
            if args.quality:
                quality["equal_residual"] = equal_residual(sim, ea, grp, args, GX, GY, tile_w, solver_tol)
        except Exception as e:
            quality = {"error": repr(e)}
        lap("quality")
        try:      # (3) every solve converged to the reference's tolerance
            converged = converged_block(sim, ea, grp, args, GX, GY, tile_w, max(2, min(6, args.steps // 3)), solver_tol, traffic_mg, traffic_mg_note)
        except Exception as e:
            converged = {"error": repr(e)}
        lap("converged")
    sim.close()
    del sim
    # BASELINE configs[3], the strong-scaling unit: N > 1 (the driver's scaling run) measures it beside the weak line, N = 1 its denominator
    strong = None
    want_strong = not args.no_strong and not args.pmc_child and not args.force_slab and args.scaling == "weak" and N < args.strong_size and \
        ((rows and world > 1) or (single and not args.no_secondary))
    if want_strong:
        try:
            strong = strong_block(ctx, args.strong_size, 2, tile_w)
        except Exception as e:      # (collective: a failure here is every rank's)
            strong = {"error": repr(e)}
        lap("strong_block")
    if rank != 0:
        grp.close()
        return

    cpu_obj = None
    libs = None
    if extras:
        libs = None if args.no_cpu_baseline else build_native_oracle()
        # (4) time to SOLVE one system to the reference's tolerance, every mode (2048^2 half tank, first projection)
        try:
            tts = {}
            for pc in ("ic0", "ic0_tile", "ic0_tile2", "ic0_tile_mg"):
                s3 = ea.Simulation(2048, 2048, device=local_rank, dot_mode=dot_mode, precond=PC[pc], tile_records=args.tile_records,
                                   max_iterations=20000, pcg_poll_interval=8).load_half_tank()
                s3.step()                      # untimed: allocations, first launches
                s3.load_half_tank()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s3.step()
                torch.cuda.synchronize()
                st = s3.stats()
                tts[pc] = {"ms": round(1e3 * (time.perf_counter() - t0), 2), "pcg_iterations": int(st.last_pcg_iterations),
                           "residual": float(st.last_residual), "substeps": int(st.last_substeps)}
                s3.close()
                del s3
            tts["workload"] = "2048x2048 half tank from rest, one frame, tol 1e-6 (the reference's), iteration cap lifted to 20000"
            secondary["time_to_solution"] = tts
        except Exception as e:
            secondary["time_to_solution"] = {"error": repr(e)}
        lap("time_to_solution")
        # (5) BASELINE configs[1]: 1024^2 dam break in the expensive phase, parity mode, checked in-run against the oracle
        try:
            s4 = ea.Simulation(1024, 1024, device=local_rank, dot_mode=dot_mode, precond=ea.PRECOND_IC0)
            load_workload(s4, scenarios, "dam_break", 1)
            pre4 = preroll_into_solves(s4, args.max_preroll)
            cpu1 = cpu_from_gpu_state(s4, ea, libs) if libs else None
            parity = None
            done = 0
            if cpu1 and "_oracle_after" in cpu1:
                import numpy as np
                ou, ov, ofl, k = cpu1.pop("_oracle_after")
                for _ in range(k):
                    s4.step()
                done = k
                gu, gv, gfl = s4.get(ea.F_U), s4.get(ea.F_V), s4.get(ea.F_COUNT) > 0
                parity = {"frames": k, "max_abs_du": float(np.abs(gu - ou).max()), "max_abs_dv": float(np.abs(gv - ov).max()),
                          "fluid_cells_differing": int((gfl != ofl).sum()),
                          "vs": "oracle (strict IEEE build) from the same state, EULER_DOT_TREE on the GPU"}
            t4 = time_frames(s4, ea, grp, args, "ic0", 4, done, 1, False)
            c1 = summarize(t4, 1024, 1024, "ic0", tile_w, None, "no PMC pass inside this block (profiles/ holds one)", 4)
            c1.update({"workload": "1024x1024 dam break (block layout upscaled), preroll %d frames into the expensive phase" % pre4, "steps": 4,
                       "parity_in_run": parity,
                       "cpu_same_state": {k: v for k, v in (cpu1 or {}).items() if not k.startswith("_")} or None})
            # the roofline mode at this size: the resident solver (one persistent launch per solve, vectors in registers), its float variant (configs[1] "fp32": a labelled
            # secondary, narrower than the reference's double PCG) and the multi-kernel form, all from the state the parity-mode frames left
            state = {f: s4.get(f) for f in (ea.F_SOLID, ea.F_SOURCE, ea.F_SINK, ea.F_COUNT, ea.F_PREV_COUNT, ea.F_U, ea.F_V, ea.F_UTMP, ea.F_VTMP, ea.F_PRECON)}
            state_m, state_st = s4.get(ea.F_MARKERS), s4.stats()
            s4.close()
            del s4
            variants = {}
            for name, kw in (("resident_f64", {}), ("resident_f32", dict(pcg_precision=ea.PCG_F32)), ("multi_kernel_f64", dict(resident=ea.RESIDENT_OFF))):
                sv = ea.Simulation(1024, 1024, device=local_rank, dot_mode=dot_mode, precond=ea.PRECOND_IC0_TILE, tile_records=args.tile_records, **kw)
                for f, a in state.items():
                    sv.set(f, a)
                sv.set_markers(state_m)
                sv.set_rng(state_st.rng_state, state_st.source_exhausted)
                tv = time_frames(sv, ea, grp, args, "ic0_tile", 8, 0, 1, False)
                bv = summarize(tv, 1024, 1024, "ic0_tile", tile_w, None, None, 8)
                variants[name] = {"value": bv["value"], "ms_per_step": bv["ms_per_step"], "substeps": bv["substeps"], "pcg_iterations": bv["pcg_iterations"],
                                  "us_per_iteration": (bv["pcg_iteration"] or {}).get("us_per_iteration"), "resident_info": list(sv.resident_info())}
                sv.close()
                del sv
            c1["roofline_mode"] = variants
            c1["roofline_mode_value"] = variants["resident_f64"]["value"]
            c1["roofline_mode_us_per_iteration"] = variants["resident_f64"]["us_per_iteration"]
            c1["f32_value"] = variants["resident_f32"]["value"]
            c1["f32_us_per_iteration"] = variants["resident_f32"]["us_per_iteration"]
            c1["multi_kernel_us_per_iteration"] = variants["multi_kernel_f64"]["us_per_iteration"]
            secondary["configs1_1024_dam_break"] = c1
        except Exception as e:
            secondary["configs1_1024_dam_break"] = {"error": repr(e)}
        lap("configs1")
        # (5b, --quality) the roofline mode's fields against the reference's preconditioner, one state per BASELINE workload
        if libs and args.quality:
            try:
                secondary["parity_vs_reference_ic0"] = parity_vs_reference(ea, scenarios, libs["strict"], local_rank, dot_mode, args.tile_records)
            except Exception as e:
                secondary["parity_vs_reference_ic0"] = {"error": repr(e)}
            lap("parity_vs_reference")
        # (6) the converged frame against the reference's algorithm run to convergence
        if libs and isinstance(converged, dict) and "error" not in converged:
            try:
                converged["deviation_vs_reference_converged"] = converged_deviation(ea, scenarios, libs["strict"], local_rank, dot_mode, args.tile_records)
            except Exception as e:
                converged["deviation_vs_reference_converged"] = {"error": repr(e)}
            lap("converged_deviation")
        # (7) the north star's target size: the pressure projection at 16384^2 (half tank, tol = 0: 100 iterations per substep), with its own PMC pass
        if not args.no_16384 and N < 16384:
            try:
                t0 = time.perf_counter()
                s5 = ea.Simulation(16384, 16384, device=local_rank, dot_mode=dot_mode, precond=PC[args.precond], tile_records=args.tile_records, tol=0.0)
                s5.load_half_tank()
                setup_s = time.perf_counter() - t0
                preroll_into_solves(s5, 4)
                t5 = time_frames(s5, ea, grp, args, args.precond, 1, 0, 0, True)
                st5 = t5
                s5.close()
                del s5
                traffic16 = note16 = None
                if not args.no_pmc:      # (after the handle is gone: the child processes need the HBM)
                    child16 = list(child_common)
                    child16[child16.index("--size") + 1] = "16384"
                    traffic16, note16 = pmc_live(child16 + ["--precond", args.precond, "--max-preroll", "4"])
                p16 = summarize(st5, 16384, 16384, args.precond, tile_w, traffic16, note16, 1)
                p16["workload"] = "16384x16384 half tank, tol 0, 100 iterations per substep, 1 frame"
                p16["steps"] = 1
                p16["setup_seconds"] = round(setup_s, 1)
                secondary["projection_16384"] = p16
            except Exception as e:
                secondary["projection_16384"] = {"error": repr(e)}
            lap("projection_16384")
    if extras or (world > 1 and not args.no_cpu_baseline):
        if not single:
            libs = build_native_oracle()
        # (8) the CPU path beside it (rank 0): single thread, bounded sample
        if libs:
            try:
                NS = args.cpu_sample_size
                cpu = cpu_baseline_roofline_run(libs, tol if tol is not None else 1e-6, NS)
                ref = cpu["reference_flags"]
                per_cell_substep = ref["seconds"] / (NS * NS * max(ref["substeps"], 1))
                cpu_obj = {"value": round(ref["value"], 1), "unit": "cells*steps/s", "cores": 1, "kind": "port",
                           # like for like with the GPU line's cells_substeps_per_s: a CPU sample frame from rest is ONE substep, a GPU headline step is eight
                           "cells_substeps_per_s": round(ref["value"] * max(ref["substeps"], 1), 1), "substeps_per_step": int(ref["substeps"]),
                           "sample": "1 frame = %d substep(s) (the GPU headline step: 8 substeps - compare cells_substeps_per_s), %d PCG iterations, of the %dx%d half tank - the headline workload at 1/%d of its "
                                     "cells, same fluid fraction, same tol / iteration budget; oracle/euler_oracle.c built -O3 -ffast-math "
                                     "-march=native (the reference's CMake flags), single thread like the reference" % (ref["substeps"], ref["pcg_iterations"], NS, NS, max(1, (8192 // NS) ** 2)),
                           "seconds": ref["seconds"], "strict_ieee_value": round(cpu["strict"]["value"], 1), "cpu_model": cpu_model(),
                           "extrapolated_seconds_per_substep": {"8192x8192": round(per_cell_substep * 8192 * 8192, 1),
                                                                "16384x16384": round(per_cell_substep * 16384 * 16384, 1),
                                                                "note": "EXTRAPOLATED from the %d^2 sample at constant time per cell and substep (100 iterations each)" % NS},
                           "configs0_100x40_block_100_steps": cpu.get("_native"),
                           "reference_main_c_100x40_block_100_steps": cpu.get("_reference"),
                           "configs1_1024_dam_break_same_state": (secondary.get("configs1_1024_dam_break") or {}).get("cpu_same_state"),
                           "host_cores_available": os.cpu_count()}
            except Exception as e:
                cpu_obj = {"error": repr(e)}
            if single and isinstance(converged, dict) and "error" not in converged:
                try:
                    converged["cpu_baseline_equal_tolerance"] = cpu_converged_baseline(libs)
                    if isinstance(cpu_obj, dict):
                        cpu_obj["equal_tolerance"] = converged["cpu_baseline_equal_tolerance"]
                except Exception as e:
                    converged["cpu_baseline_equal_tolerance"] = {"error": repr(e)}
            lap("cpu_baseline")

    transports = (("peer-to-peer mailboxes (PCG scalars, ghost rows of s) + " if p2p_on else "")
                  + ("RCCL over xGMI, called from the C library on the kernels' stream" if args.comm == "rccl" else "torch.distributed callbacks"))
    pc_name = "tile-local IC(0): no coupling between slabs" if args.precond in TILE_MODES else ("slab-local" if rows else args.slab) + " IC(0) coupling"
    parallelism = "1 GPU" if args.gpus == 1 and not sharded else (
        "%d independent replicas" % args.gpus if not sharded else
        ("%d row slabs of %s rows, every stage decomposed (%.2f GB of HBM per rank; ghost rows, marker migration, dt all-reduce, distributed PCG with %s; exchanges by %s); grid %dx%d"
         % (args.gpus, ("fluid-balanced numbers of" if partition else str(GY // max(world, 1))), hbm_per_rank / 1e9, pc_name, transports, GX, GY)) if rows else
        "%d row slabs of %d rows: distributed PCG (%s, exchanges by %s), replicated marker/advection stages; grid %dx%d"
        % (args.gpus, GY // max(world, 1), pc_name, transports, GX, GY))
    parallelism_short = "1 GPU" if args.gpus == 1 and not sharded else (
        "%d independent replicas" % args.gpus if not sharded else
        "%d row slabs%s; %s; exchanges: %s%s" % (args.gpus, (" (%s rows each), EVERY stage decomposed" % ("fluid-balanced" if partition else str(GY // max(world, 1)))) if rows else
                                               " of the pressure solve only", pc_name, "peer-to-peer mailboxes + " if p2p_on else "", "RCCL" if args.comm == "rccl" else "torch.distributed callbacks"))
    full = {
        "metric": "cells*steps/sec of sim_step() (frames incl. the PCG pressure projection) + pressure-solve HBM GB/s vs roofline",
        "value": job_rate,
        "unit": "cells*steps/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t["elapsed"] / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f64",      # the pressure solve computes in double on float fields: the reference's mix (main.c:64-67,577-578,716); the f32 variant is a labelled secondary
        "dtype_note": "f64 PCG vectors, f32 velocity / marker fields (the reference's mix, main.c:64-67,577-578,716)",
        "data": "synthetic",
        "config": {"workload": "%dx%d %s%s, %s" % (GX, GY, args.workload,
                                                   " (BASELINE configs[2], pressure-solve roofline run: tol 0, exactly 100 PCG iterations per substep%s)"
                                                   % ("; timed in the saturated phase: 8 CFL substeps per frame" if saturate else "")
                                                   if args.workload == "half_tank" and tol == 0.0 else "", args.precond),
                   "grid": [GX, GY], "preroll_frames": preroll, "precond": args.precond, "tile_records": tile_w if args.precond in TILE_MODES else None,
                   "dot_mode": args.dot_mode, "max_iterations": args.max_iterations, "tol": tol if tol is not None else 1e-6, "parallelism": parallelism_short, "parallelism_detail": parallelism},
        "mode": head["mode"],
        "substeps": head["substeps"], "pcg_iterations": head["pcg_iterations"], "cells_substeps_per_s": head["cells_substeps_per_s"],
        "markers": head["markers"], "fluid_cells": head["fluid_cells"], "hbm_bytes_this_rank": int(hbm_per_rank),
        "roofline": head["roofline"],
        "pcg_iteration": head["pcg_iteration"],
        "balance": balance,
        "comm_calls_rank0": comm_calls,
        "comm": comm_info,
        "kernels": head["kernels"],
        "stages": head.get("stages"),      # round 6: kernel time per substep of the stages AROUND the iterations (marker advection + binning, assembly, velocity update ...)
        "cpu_baseline": cpu_obj,
        # the same workload with EVERY solve run to the reference's tolerance 1e-6 (multilevel mode, cap lifted) - not the headline (whose work is fixed at 100 iterations per
        # substep by BASELINE configs[2]), the figure for "this grid, actually solved", with its own roofline object, the CPU at equal tolerance and the deviation from the
        # reference's algorithm run to convergence
        "converged": converged,
        "quality": quality,
        "strong_%d_dam_break" % args.strong_size: strong,
        "secondary": secondary or None,
        "device": device,
        "timings_s": timings,
    }
    full["full"] = write_full(full)
    emit(json.dumps(compact_line(full)))
    os.dup2(2, 1)
    grp.close()


if __name__ == "__main__":
    main()
