#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configuration.

metric   cells*steps/s of the whole sim_step() path (one step = one 0.1 s frame = up to 8 CFL
         substeps, each with a <=100-iteration PCG pressure projection), plus the achieved HBM
         rate of the dominant pressure-solve kernel against the MI355X roofline.
workload N=1: configs[1], the 1024x1024 dam break (block layout upscaled), synthetic.
         The reference's precision mix is kept: float fields, double PCG vectors.
         The dam first falls freely: for ~22 frames the divergence is exactly zero and the
         reference's `all_zero(r)` test (main.c:742) skips the solve, so a frame costs < 1 ms.  From
         then on float rounding of the growing velocities leaves a residual above the 1e-6 tolerance
         and EVERY substep runs the full 100 PCG iterations (8 substeps/frame at peak).  The bench
         "prerolls" untimed until a frame needs >= 100 PCG iterations, so the timed window always
         lies in the expensive phase, never in the free-fall phase.

One JSON line on stdout (rank 0).  Inputs are resident in HBM when the timed region starts.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# algorithmic bytes per grid cell per launch (w = 8 for double vectors, 1 mask byte)
W = 8
ALGO_BYTES = {
    "forward_solve": 3 * W + 1,    # read r, precon; write q
    "backward_solve": 3 * W + 1,   # read q, precon; write z     (dot(z,r) is a separate launch here)
    "apply_a": 2 * W + 1,          # read s; write z (+ in-register dot partial)
    "dot": 2 * W + 1,              # read z, r
    "update_pr": 6 * W + 1,        # read s, z, p, r; write p, r
    "update_search": 3 * W + 1,    # read z, s; write s
}
PCG_BYTES_PER_CELL_ITER = 18 * W + 5   # 149 B, the figure BASELINE.md prescribes for IC(0) PCG


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="N of the NxN grid (default: configs[1] = 1024)")
    ap.add_argument("--workload", default="dam_break", choices=["dam_break", "half_tank", "waterfall"])
    ap.add_argument("--dot-mode", default="tree", choices=["tree", "sequential"])
    ap.add_argument("--precond", default="ic0", choices=["ic0", "jacobi", "ic0_tile"])
    ap.add_argument("--tile-units", type=int, default=0)
    ap.add_argument("--max-preroll", type=int, default=400)
    ap.add_argument("--slab", default="local", choices=["local", "exact", "replicas"],
                    help="N>1: slab-local IC(0) (scales; tolerance-only), exact coupling (the 1-GPU iterates; sweeps "
                         "serialize across GPUs) or independent replicas")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="N>1 exchange transport: the library's own RCCL communicator (C, no host code between kernels) "
                         "or the torch.distributed callbacks of euler_amd/slab.py")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = N x (N*gpus) grid, one tank per row slab (default; the driver's scaling run); strong = the "
                         "N x N grid of --size split into row slabs (BASELINE configs[3]: --size 16384 --scaling strong)")
    ap.add_argument("--no-p2p", action="store_true",
                    help="N>1: keep the per-iteration exchanges (3 scalar all-reduces, ghost rows) on the communicator instead of "
                         "the peer-to-peer mailboxes of csrc/comm_p2p.hip")
    ap.add_argument("--grid-y-mult", type=int, default=0,
                    help="diagnostics: run the N x (N*M) grid of an M-GPU weak-scaling job on the GPUs given (e.g. on one GPU: the "
                         "single-GPU time of the 8-GPU job's grid)")
    ap.add_argument("--force-slab", action="store_true",
                    help="N=1 diagnostics: run the communicator code path with one rank (every exchange still goes through RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the hipEvent per-kernel timing (used under rocprofv3)")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class (diagnostics)")
    return ap.parse_args()


def build_native_oracle():
    """cpu_baseline leg only: compile the oracle for THIS host (reference flags -O3 -ffast-math
    -march=native, CMakeLists.txt:11,18, and strict IEEE) into a temp dir."""
    src = os.path.join(ROOT, "oracle", "euler_oracle.c")
    out = {}
    d = tempfile.mkdtemp(prefix="euler_oracle_")
    for name, flags in (("strict", ["-O3", "-ffp-contract=off"]), ("reference_flags", ["-O3", "-ffast-math", "-march=native"])):
        so = os.path.join(d, "liboracle_%s.so" % name)
        subprocess.check_call(["gcc", "-std=gnu99", "-fPIC", "-shared"] + flags + ["-o", so, src, "-lm"])
        out[name] = so
    return out


def cpu_baseline(sim, ea, steps_budget_s=12.0):
    """Time the CPU oracle (kind 'port': the from-scratch restatement proven bit-identical to the
    compiled reference at 100x40) on this host, single thread like the reference, starting from the
    SAME state the GPU timing starts from.  Bounded sample: one frame (<= 8 substeps)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    libs = build_native_oracle()
    res = {}
    snap = {n: sim.get(f) for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"),
                                        (ea.F_SINK, "sink"), (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"),
                                        (ea.F_PRECON, "precon"), (ea.F_MARKERS, "markers"))}
    st = sim.stats()
    for name, so in libs.items():
        o = oracle_lib.Oracle(sim.X, sim.Y, lib_path=so)
        for n in ("u", "v", "solid", "source", "sink", "count", "prev_count", "precon"):
            getattr(o, n)[...] = snap[n]
        o.set_markers(snap["markers"])
        o.c.rng_state = st.rng_state
        o.c.source_exhausted = st.source_exhausted
        t0 = time.perf_counter()
        nsteps = 0
        while True:
            o.step()
            nsteps += 1
            if time.perf_counter() - t0 > steps_budget_s or nsteps >= 3:
                break
        dt = time.perf_counter() - t0
        res[name] = dict(value=sim.X * sim.Y * nsteps / dt, seconds=dt, steps=nsteps,
                         substeps=int(o.c.total_substeps), pcg_iterations=int(o.c.total_pcg_iterations))
        if name == "strict":
            res["_oracle_after"] = (o.u.copy(), o.v.copy(), (o.count > 0).copy(), nsteps)
        o.close()
    # BASELINE configs[0]: the reference's own grid and scenario (block layout, 100 x 40, 100 frames), the oracle only
    try:
        from euler_amd import scenarios as _sc
        o = oracle_lib.Oracle(100, 40, lib_path=libs["reference_flags"]).load_text(_sc.dam_break())
        t0 = time.perf_counter()
        for _ in range(100):
            o.step()
        dt = time.perf_counter() - t0
        res["_native"] = dict(value=4000 * 100 / dt, seconds=round(dt, 3), steps=100, substeps=int(o.c.total_substeps),
                              pcg_iterations=int(o.c.total_pcg_iterations))
        o.close()
    except Exception as e:      # never let the extra figure break the bench line
        res["_native"] = {"error": str(e)}
    return res


PMC_KERNEL = {"forward_solve": "k_sweep_skew<1>", "backward_solve": "k_sweep_skew<2>", "precon_factor": "k_sweep_skew<0>",
              "apply_a": "k_apply_a", "dot": "k_dot_partial", "update_pr": "k_update_pr", "update_search": "k_update_search<false>"}


def pmc_traffic(size, workload, kernel_class):
    """HBM bytes per launch of a kernel class from the committed PMC passes of the same workload
    (profiles/pmc_traffic_<size>_<workload>.json, written by tools/summarize_profile.py from
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this very command); None if there is none."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic_%d_%s.json" % (size, workload))
    try:
        name = PMC_KERNEL.get(kernel_class, "")
        kernels = json.load(open(p))["kernels"]
        # template arguments added since a PMC file was written: k_sweep_skew<OP, XG>, k_dot_partial<EDGES>, k_search_apply<SLAB>
        k = kernels.get(name) or kernels.get(name.replace(">", ", false>")) or kernels.get(name + "<false>")
        return int(k["hbm_bytes_per_launch"]) if k else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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

    N = args.size
    dot_mode = ea.DOT_TREE if args.dot_mode == "tree" else ea.DOT_SEQUENTIAL
    precond = {"ic0": ea.PRECOND_IC0, "jacobi": ea.PRECOND_JACOBI, "ic0_tile": ea.PRECOND_IC0_TILE}[args.precond]
    # N > 1 (weak scaling): the grid grows to N x (N * gpus) rows; the pressure solve is split into one
    # slab of N rows per GPU, the cheap stages run replicated (DESIGN.md "Multi-GPU")
    sharded = (world > 1 or args.force_slab) and args.slab != "replicas"
    GX, GY = N, N * (args.grid_y_mult if args.grid_y_mult > 0 else (world if sharded and args.scaling == "weak" else 1))
    if world > 1:
        torch.cuda.set_device(local_rank)
    sim = ea.Simulation(GX, GY, device=local_rank, dot_mode=dot_mode, precond=precond, tile_units=args.tile_units)
    comm = None
    p2p_on = False
    if sharded:
        from euler_amd.slab import SLAB_EXACT, SLAB_LOCAL, RcclComm, TorchComm, attach_p2p
        coupling = SLAB_EXACT if args.slab == "exact" else SLAB_LOCAL
        if args.comm == "rccl":
            from euler_amd.slab import RcclUnavailable
            try:
                comm = RcclComm(sim, coupling)
            except RcclUnavailable as e:          # raised on every rank alike: the job goes on over torch.distributed, and says so
                if rank == 0:
                    print("bench: %s; exchanges fall back to torch.distributed callbacks" % e, file=sys.stderr)
                args.comm = "torch"
        if args.comm == "torch":
            comm = TorchComm(sim, coupling)
        # the latency-bound exchanges of every PCG iteration go peer to peer (xGMI); if the mailboxes cannot be set up
        # on every rank the job stays on the communicator, and the JSON line says which one ran
        p2p_on = (not args.no_p2p) and attach_p2p(sim)
        if rank == 0 and not args.no_p2p and not p2p_on:
            print("bench: peer-to-peer mailboxes unavailable (%s); exchanges stay on %s" % (sim._p2p_error, args.comm), file=sys.stderr)
    # weak scaling: the N x (N*M) grid holds M copies of the single-GPU picture on top of each other (closed tanks), so
    # that every row slab does the work of the N = 1 job: same free-fall phase, same substeps, same iteration counts
    tiles = GY // N
    if args.workload == "dam_break":
        sim.load_text(scenarios.stacked(scenarios.dam_break(), tiles), upscale=True)
    elif args.workload == "waterfall":
        sim.load_text(scenarios.stacked(scenarios.waterfall(), tiles), upscale=True)
    else:
        sim.load_half_tank()

    # untimed preroll into the expensive phase: the first frame whose solves run the full
    # iteration budget (>= 100 PCG iterations in the frame); see the module docstring
    preroll = 0
    while preroll < args.max_preroll:
        sim.step()
        preroll += 1
        if sim.stats().last_pcg_iterations >= 100:
            break

    # CPU baseline from the same state (rank 0, N=1 only)
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and not args.force_slab:
        cpu = cpu_baseline(sim, ea)

    # in-run parity note: advance the GPU by the same number of frames the strict-IEEE oracle ran
    parity = None
    warm_done = 0
    if cpu and "_oracle_after" in cpu:
        import numpy as np
        ou, ov, ofl, k = cpu.pop("_oracle_after")
        for _ in range(k):
            sim.step()
        warm_done = k
        gu, gv, gfl = sim.get(ea.F_U), sim.get(ea.F_V), sim.get(ea.F_COUNT) > 0
        parity = {"frames": k, "max_abs_du": float(np.abs(gu - ou).max()), "max_abs_dv": float(np.abs(gv - ov).max()),
                  "fluid_cells_differing": int((gfl != ofl).sum()), "vs": "oracle (strict IEEE build) from the same state"}
    for _ in range(max(args.warmup - warm_done, 0)):
        sim.step()

    dominant = "backward_solve" if precond != ea.PRECOND_JACOBI else "update_pr"
    # Inside the timed region only the DOMINANT kernel is bracketed by HIP events (on the kernel's own
    # stream): an event pair around every launch of all six PCG kernels costs ~20 % throughput at
    # 1024^2 (measured), around the dominant one alone ~2 %.  The other classes are timed in a second,
    # untimed pass of the same number of steps right after it (same phase: every substep runs the
    # full iteration budget).
    timed_classes = [dominant] if not args.no_kernel_timing else []
    sim.profile_reset()
    sim.profile_enable(timed_classes)
    st0 = sim.stats()

    # timed region: barrier + device sync on both sides, MAX over ranks (euler_amd/dist.py)
    elapsed = grp.timed(sim.step, args.steps)

    st1 = sim.stats()
    prof = sim.profile() if timed_classes else {}
    sim.profile_enable([])
    iters_pass2 = 0
    if not args.no_kernel_timing:
        sim.profile_reset()
        sim.profile_enable(ea.profile_class_names() if args.profile_all else [k for k in ALGO_BYTES if k != dominant])
        for _ in range(args.steps):
            sim.step()
        iters_pass2 = sim.stats().total_pcg_iterations - st1.total_pcg_iterations
        prof2 = sim.profile()
        sim.profile_enable([])
        for k, v in prof2.items():
            prof.setdefault(k, v)
    substeps = st1.total_substeps - st0.total_substeps
    iters = st1.total_pcg_iterations - st0.total_pcg_iterations
    cells = GX * GY
    cells_launch = cells // world if sharded else cells      # cells one kernel launch covers on one GPU

    if comm is not None and comm.error:
        raise RuntimeError(comm.error)
    # sharded: ONE job of GX*GY cells; replicas / single GPU: one job of N*N cells per rank
    job_rate = (GX * GY * args.steps / elapsed) if sharded else whole_job_rate(float(GX * GY), args.steps, elapsed, grp)
    if rank != 0:
        grp.close()
        return

    roof = None
    kern = {}
    try:      # the ceiling a plain device-to-device copy reaches on this very GPU (read + write), next to the 8 TB/s spec peak
        copy_gbps = round(sim.copy_bandwidth(1 << 30, 10), 1)
    except Exception:
        copy_gbps = None
    for name, (ms, launches) in prof.items():
        entry = {"ms_total": round(ms, 3), "launches": int(launches), "avg_us": round(1e3 * ms / launches, 2)}
        if name in ALGO_BYTES:
            entry["algo_GBps"] = round(ALGO_BYTES[name] * cells_launch / (ms / launches * 1e-3) / 1e9, 1)
        kern[name] = entry
    if dominant in prof:
        ms, launches = prof[dominant]
        achieved = ALGO_BYTES[dominant] * cells_launch / (ms / launches * 1e-3) / 1e9
        traffic = pmc_traffic(N, args.workload, dominant) if not sharded and GY == N else None
        roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "measured_copy_GBps": copy_gbps,
                "traffic": traffic,
                # the HBM bytes rocprofv3 counted for this kernel (committed PMC passes of this very command) over the launch time
                # measured here: what the memory system really delivered, next to the algorithmic figure above
                "achieved_traffic": round(traffic / (ms / launches * 1e-3) / 1e9, 2) if traffic else None,
                "frac_traffic": round(traffic / (ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic else None,
                "algorithmic_bytes_per_launch": ALGO_BYTES[dominant] * cells_launch,
                "note": "algorithmic bytes count ALL X*Y cells of a launch (SURVEY 8d, like the reference's dense loops); the sweeps skip "
                        "fluid-free blocks, so on sparse scenes the bytes really moved (traffic) are fewer and achieved can exceed what HBM delivered",
                "fluid_fraction": round(float(st1.fluid_cells) / cells, 4),
                "avg_launch_us": round(1e3 * ms / launches, 2), "launches": int(launches)}
    # whole PCG iteration: per-launch averages summed over the six kernel classes (dominant: timed region;
    # the others: second pass), against the 149 B per cell and iteration BASELINE.md prescribes
    # total time of each class / PCG iterations of the pass it was timed in (dominant: the timed region; the others:
    # the second pass) - a class may have fewer launches than iterations (the s = z copy at the start of a solve)
    per_iter_ms = sum(prof[k][0] / (iters if k == dominant else iters_pass2) for k in ALGO_BYTES
                      if k in prof and (iters if k == dominant else iters_pass2))
    pcg_ms = per_iter_ms * iters
    # single GPU: update_search (K5) runs fused into the next iteration's apply_a (K1) and has no launches of its own
    fused_k5 = not sharded and "apply_a" in prof
    if fused_k5:
        kern["apply_a"]["note"] = "update_search fused in: 42 algorithmic B/cell (2w+1 + 3w+1)"
        kern["apply_a"]["algo_GBps"] = round((ALGO_BYTES["apply_a"] + ALGO_BYTES["update_search"]) * cells_launch
                                              / (prof["apply_a"][0] / prof["apply_a"][1] * 1e-3) / 1e9, 1)
    complete = all(k in prof for k in ALGO_BYTES if k != "update_search" or not fused_k5)
    pcg_gbps = PCG_BYTES_PER_CELL_ITER * cells / (per_iter_ms * 1e-3) / 1e9 if per_iter_ms and complete else None

    cpu_obj = None
    if cpu:
        ref = cpu["reference_flags"]
        cpu_obj = {"value": round(ref["value"], 1), "unit": "cells*steps/s", "cores": 1, "kind": "port",
                   "sample": "%d frame(s) (%d substeps, %d PCG iterations) of the same %dx%d %s state the GPU timing starts from; "
                             "oracle/euler_oracle.c built -O3 -ffast-math -march=native (the reference's CMake flags), single thread"
                             % (ref["steps"], ref["substeps"], ref["pcg_iterations"], N, N, args.workload),
                   "strict_ieee_value": round(cpu["strict"]["value"], 1), "cpu_model": cpu_model(),
                   "configs0_100x40_block_100_steps": cpu.get("_native"),
                   "host_cores_available": os.cpu_count()}

    out = {
        "metric": "cells*steps/sec of sim_step() (dam-break frames incl. PCG pressure projection)",
        "value": job_rate,
        "unit": "cells*steps/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f32 fields, f64 PCG (the reference's mix)",
        "data": "synthetic",
        "config": {"workload": "%dx%d %s, %s%s" % (GX, GY, args.workload, "block layout upscaled" if args.workload == "dam_break" else "synthetic",
                                                   "" if tiles == 1 or args.workload == "half_tank" else ", %d tanks stacked (one per row slab)" % tiles),
                   "grid": [GX, GY], "preroll_frames": preroll, "precond": args.precond, "dot_mode": args.dot_mode,
                   "max_iterations": 100, "tol": 1e-6,
                   "parallelism": "1 GPU" if args.gpus == 1 and not sharded else (
                       "%d independent replicas" % args.gpus if not sharded else
                       "%d row slabs of %d rows: distributed PCG (%s IC(0) coupling, exchanges by %s), replicated marker/advection stages; grid %dx%d"
                       % (args.gpus, GY // max(world, 1), args.slab,
                          ("peer-to-peer mailboxes (scalars, ghost rows) + " if p2p_on else "")
                          + ("RCCL from the C library" if args.comm == "rccl" else "torch.distributed callbacks"), GX, GY))},
        "substeps": int(substeps), "pcg_iterations": int(iters),
        "cells_substeps_per_s": cells * substeps / elapsed,
        "markers": int(st1.n_markers), "fluid_cells": int(st1.fluid_cells),
        "roofline": roof,
        "pcg_aggregate": {"algorithmic_GBps": round(pcg_gbps, 1) if pcg_gbps else None,
                          "bytes_per_cell_iteration": PCG_BYTES_PER_CELL_ITER, "kernel_us_per_iteration": round(1e3 * per_iter_ms, 2),
                          "note": "dominant kernel timed inside the timed region, the other five in a second pass of the same length"},
        "kernels": kern,
        "cpu_baseline": cpu_obj,
        "parity_in_run": parity,
        "device": sim.device_name(),
    }
    emit(json.dumps(out))
    os.dup2(2, 1)
    grp.close()


if __name__ == "__main__":
    main()
