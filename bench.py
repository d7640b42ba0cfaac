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
        # multilevel: + the V-cycle: 48 doubles of partial sums per 16x64 tile written and read (0.75 B/cell) and the node grids - a node per 256 cells, per node nine stencil
        # entries read on the way down and again on the way up + right-hand side / result (~190 B per node of level 0, a third more for the levels above: ~1.0 B/cell)
        "ic0_tile_mg": {"apply_a": apply_, "precond_tile": _TILE, "coarse_cycle": 1.75},
        "jacobi": {"apply_a": apply_, "update_pr": _RUPD, "jacobi": 2 * W + 1, "dot": 2 * W + 1},
    })
    PCG_BYTES.clear()
    PCG_BYTES.update({m: sum(c.values()) for m, c in ITER_BYTES.items()})


set_as_stored(False)
ONCE_PER_SOLVE_BYTES = {"update_pr": 3 * W + 1}      # k_finish_p in the tile modes: the last one or two p += alpha s (read s, p; write p)
PCG_CLASSES = ["forward_solve", "backward_solve", "apply_a", "dot", "update_pr", "update_search", "precond_tile", "coarse_cycle", "jacobi", "resident_pcg"]
KERNEL_OF_CLASS = {"forward_solve": "k_sweep_skew<1", "backward_solve": "k_sweep_skew<2", "precon_factor": "k_sweep_skew<0",
                   "apply_a": "k_search_apply", "dot": "k_dot_partial", "update_pr": "k_update_pr", "precond_tile": "k_precond_tile"}
MODE_NAME = {"ic0": "parity mode: the reference's IC(0), bit-identical iterates",
             "ic0_tile": "roofline mode: tile-local IC(0) (64x%d-cell blocks), NOT the reference's iterates (tolerance parity where PCG converges)",
             "ic0_tile2": "two-level mode: tile-local IC(0) (64x%d-cell blocks) + a coarse correction (<= 256 aggregates, dense inverse), NOT the reference's iterates",
             "ic0_tile_mg": "multilevel mode: tile-local IC(0) (64x%d-cell blocks) + one V-cycle over node grids of 16, 32, ... cells spacing (bilinear interpolation, nine-point Galerkin stencils, dense top level), NOT the reference's iterates",
             "jacobi": "Jacobi stand-in, NOT the reference's iterates"}
TILE_MODES = ("ic0_tile", "ic0_tile2", "ic0_tile_mg")


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
                         "correction (one GPU; fewer iterations to a given residual, DESIGN.md 5c)")
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


# ------------------------------------------------------------------------------------------------ CPU baseline
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_roofline_run(libs, tol, N=2048):
    """The oracle ('port': the from-scratch restatement proven bit-identical to the compiled reference at 100x40), single
    thread like the reference, on a BOUNDED sample of the headline workload: the half-filled tank at 2048^2 (1/16 of the
    8192^2 grid, same fluid fraction, same tol = 0 / 100 iterations per substep), one frame; plus configs[0]."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    res = {}
    for name, so in libs.items():
        o = oracle_lib.Oracle(N, N, lib_path=so).load_half_tank()
        o.c.tol = tol
        t0 = time.perf_counter()
        o.step()
        dt = time.perf_counter() - t0
        res[name] = dict(value=N * N / dt, seconds=round(dt, 3), steps=1, substeps=int(o.c.total_substeps),
                         pcg_iterations=int(o.c.total_pcg_iterations))
        o.close()
    try:      # BASELINE configs[0]: the reference's own grid and scenario (block layout, 100 x 40, 100 frames)
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
    try:      # ... and the COMPILED REFERENCE itself (oracle/_ref: the unmodified main.c, built -O3 -ffp-contract=off in the build container) on the only grid it has
        if oracle_lib.have_ref():
            from golden_util import load as gload, scenario_text
            path = os.path.join(tempfile.mkdtemp(prefix="euler_ref_"), "block.txt")
            with open(path, "w") as f:
                f.write(scenario_text(gload("block_frames.npz")))
            r = oracle_lib.Reference().init(path)
            t0 = time.perf_counter()
            for _ in range(100):
                r.step()
            dt = time.perf_counter() - t0
            res["_reference"] = dict(value=4000 * 100 / dt, seconds=round(dt, 3), steps=100, kind="reference",
                                     what="oracle/_ref/libeuler_ref.so: the unmodified reference main.c (gcc -O3 -ffp-contract=off), scenarios/block.txt on its compile-time 100x40 grid, 100 sim_step calls, single thread")
    except Exception as e:
        res["_reference"] = {"error": str(e)}
    return res


def cpu_from_gpu_state(sim, ea, libs, budget_s=10.0, max_steps=2):
    """configs[1] block: the oracle from the SAME state the GPU timing starts from (single thread); also returns the strict
    build's state after its frames for the in-run parity note."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
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
            if time.perf_counter() - t0 > budget_s or nsteps >= max_steps:
                break
        dt = time.perf_counter() - t0
        res[name] = dict(value=sim.X * sim.Y * nsteps / dt, seconds=round(dt, 3), steps=nsteps,
                         substeps=int(o.c.total_substeps), pcg_iterations=int(o.c.total_pcg_iterations))
        if name == "strict":
            res["_oracle_after"] = (o.u.copy(), o.v.copy(), (o.count > 0).copy(), nsteps)
        o.close()
    return res


# ------------------------------------------------------------------------------------------------ live PMC passes
def pmc_live(child_args, timeout_s=420):
    """HBM bytes per launch of every kernel of THIS workload from two rocprofv3 PMC passes run as child processes of this
    run (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; MI355X_MICROARCH.md 'rocprofv3 PMC slots').  The guide's
    corrections are not assumed but CALIBRATED in the same pass: the child also runs the library's copy probe, whose
    launches move a known number of bytes (k_copy16: 2^30 read + 2^30 written), and the KiB the counters report for it give
    the factor for reads (the guide: x2 for wide coalesced reads on gfx950) and for writes.  -> (dict kernel -> bytes, note)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ):
        return None, "this run is itself being profiled: no nested PMC passes"
    out = {}
    base = tempfile.mkdtemp(prefix="euler_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__)] + child_args
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or "")[-300:])
            con = sqlite3.connect(dbs[0])
            tables = [t[0] for t in con.execute("select name from sqlite_master where type in ('table','view')")]
            if "counters_collection" not in tables:
                return None, "no counters_collection view in %s" % os.path.basename(dbs[0])
            rows = con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name",
                               (counter,)).fetchall()
            out[counter] = {k.split("(")[0].replace("void ", ""): (n, v) for k, n, v in rows}
        cal = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            probe = [v for k, v in out[counter].items() if k.startswith("k_copy16")]
            if not probe or probe[0][1] <= 0:
                return None, "copy probe missing from the %s pass" % counter
            cal[counter] = float(1 << 30) / (probe[0][1] * 1024.0)      # true bytes per reported byte
        traffic = {}
        for k, (n, v) in out["FETCH_SIZE"].items():
            w = out["WRITE_SIZE"].get(k, (0, 0.0))[1]
            traffic[k] = {"launches": int(n), "read_bytes": v * 1024.0 * cal["FETCH_SIZE"], "write_bytes": w * 1024.0 * cal["WRITE_SIZE"]}
            traffic[k]["bytes"] = traffic[k]["read_bytes"] + traffic[k]["write_bytes"]
        note = ("live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (separate child processes), KiB -> bytes, "
                "calibrated on the copy probe of the same pass: reads x%.3f (guide: x2 on gfx950), writes x%.3f"
                % (cal["FETCH_SIZE"], cal["WRITE_SIZE"]))
        return traffic, note
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 pass timed out after %d s" % timeout_s
    except Exception as e:      # the bench line must survive a profiler problem
        return None, "PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(base, ignore_errors=True)


def traffic_of(traffic, cls, precond=None):
    """bytes per launch of a kernel class from a pmc_live() table (kernel names carry template arguments)."""
    if not traffic:
        return None
    key = KERNEL_OF_CLASS.get(cls)
    if cls == "update_pr" and precond is not None and precond not in TILE_MODES:
        key = "k_precond_tile"      # (r -= alpha A s, max |r| = the first half of the tile pass, r_only)
    if not key:
        return None
    hits = [(k, v) for k, v in traffic.items() if k.startswith(key)]
    if cls == "apply_a":      # iterations >= 1 run k_search_apply; the first of a solve k_apply_a (few launches)
        hits = hits or [(k, v) for k, v in traffic.items() if k.startswith("k_apply_a")]
    if not hits:
        return None
    # (several instantiations of one class - k_search_apply with and without the two-iteration p update - alternate: the class's
    # bytes per launch are their launch-weighted mean, like its average launch time)
    n = sum(v["launches"] for _, v in hits)
    return sum(v["bytes"] * v["launches"] for _, v in hits) / max(n, 1)


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
                       slab=slab_arg, max_iterations=args.max_iterations, pcg_poll_interval=8 if args.max_iterations <= 100 else 32)
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
            conv = {"mode": MODE_NAME["ic0_tile_mg"] % tile_w, "value": size * size / el, "unit": "cells*steps/s", "ms_per_step": 1e3 * el, "steps": 1, "tol": 1e-6,
                    "substeps": int(st1.total_substeps - st0.total_substeps), "pcg_iterations": int(st1.total_pcg_iterations - st0.total_pcg_iterations),
                    "last_residual": float(st1.last_residual)}
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
                                   "markers", "last_residual", "roofline", "pcg_iteration")}
        out.update({"workload": "%dx%d dam break (block layout upscaled; BASELINE configs[3]), %d timed frames after %d preroll frames "
                                "(into the phase where every substep runs PCG to the iteration cap)" % (size, size, steps, preroll),
                    "n_gpus": world if multi else 1, "scaling": "strong", "balance": balance, "hbm_bytes_this_rank": int(hbm),
                    "setup_and_preroll_seconds": round(setup_s, 1), "converged_frames_multilevel": conv,
                    "note": ("rank 0's kernels cover its slab; `value` is the whole job" if multi else
                             "one GPU: the denominator of the strong-scaling curve (run `bench.py --gpus N` for the N-GPU points of the same scenario)")})
    sim.close()
    del sim
    return out


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


def time_frames(sim, ea, grp, args, precond, steps, warmup_done, warmup, big):
    """the timed region + per-kernel HIP-event timing.  Large grids: every PCG class is bracketed inside the timed region
    (an event pair costs microseconds, the kernels hundreds); small grids (launches of ~10 us): only the dominant class,
    the others in a second pass of the same length."""
    for _ in range(max(warmup - warmup_done, 0)):
        sim.step()
    dominant = {"ic0": "backward_solve", "ic0_tile": "apply_a", "ic0_tile2": "apply_a", "ic0_tile_mg": "apply_a", "jacobi": "update_pr"}[precond]
    classes_all = ea.profile_class_names() if args.profile_all else PCG_CLASSES
    if not big and precond == "ic0_tile" and sim.resident_info()[0]:
        dominant = "resident_pcg"
    timed = [] if args.no_kernel_timing else (classes_all if big else [dominant])
    sim.profile_reset()
    sim.profile_enable(timed)
    st0 = sim.stats()
    elapsed = grp.timed(sim.step, steps)
    st1 = sim.stats()
    prof = sim.profile() if timed else {}
    sim.profile_enable([])
    iters = st1.total_pcg_iterations - st0.total_pcg_iterations
    iters2 = iters
    if not args.no_kernel_timing and not big:
        sim.profile_reset()
        sim.profile_enable([k for k in classes_all if k != dominant])
        for _ in range(steps):
            sim.step()
        iters2 = sim.stats().total_pcg_iterations - st1.total_pcg_iterations
        for k, v in sim.profile().items():
            prof.setdefault(k, v)
        sim.profile_enable([])
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
                substeps=st1.total_substeps - st0.total_substeps, resident=resident)


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
                "achieved_is": "algorithmic bytes per cell x fluid cells of one launch / average launch time (HIP events in the timed region)",
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
    return {"mode": MODE_NAME[precond] % tile_w if precond in TILE_MODES else MODE_NAME[precond],
            "value": cells_job * steps / t["elapsed"], "unit": "cells*steps/s", "ms_per_step": 1e3 * t["elapsed"] / steps,
            "substeps": int(t["substeps"]), "pcg_iterations": int(t["iters"]),
            "cells_substeps_per_s": cells_job * t["substeps"] / t["elapsed"],
            "fluid_cells": fluid, "markers": int(t["st1"].n_markers), "last_residual": float(t["st1"].last_residual),
            "roofline": roof, "pcg_iteration": agg, "kernels": rows}


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


def oracle_from_sim(sim, ea, so, tile_records=0):
    """an oracle (test infrastructure, checker only) holding exactly the state of a GPU handle"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle(sim.X, sim.Y, lib_path=so)
    o.c.tile_records = tile_records
    for f, n in ((ea.F_U, "u"), (ea.F_V, "v"), (ea.F_UTMP, "utmp"), (ea.F_VTMP, "vtmp"), (ea.F_SOLID, "solid"), (ea.F_SOURCE, "source"),
                 (ea.F_SINK, "sink"), (ea.F_COUNT, "count"), (ea.F_PREV_COUNT, "prev_count"), (ea.F_PRECON, "precon")):
        getattr(o, n)[...] = sim.get(f)
    o.set_markers(sim.get(ea.F_MARKERS))
    st = sim.stats()
    o.c.rng_state = st.rng_state
    o.c.source_exhausted = st.source_exhausted
    return o


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
                               "roofline", "pcg_iteration", "kernels")}
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
    sim = ea.Simulation(n, n, device=device, dot_mode=dot_mode, precond=ea.PRECOND_IC0_TILE_MG, tile_records=tile_records, max_iterations=20000, pcg_poll_interval=32)
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


def cpu_converged_baseline(libs, n=1024):
    """cpu_baseline at EQUAL TOLERANCE: the oracle with the reference's IC(0), single thread, tol 1e-6, cap lifted, one frame of the n x n half tank from rest"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    o = oracle_lib.Oracle(n, n, lib_path=libs["reference_flags"]).load_half_tank()
    o.c.max_iterations = 20000
    t0 = time.perf_counter()
    o.step()
    dt = time.perf_counter() - t0
    out = {"value": round(n * n / dt, 1), "unit": "cells*steps/s", "cores": 1, "kind": "port", "seconds": round(dt, 2),
           "substeps": int(o.c.total_substeps), "pcg_iterations": int(o.c.total_pcg_iterations), "last_residual": float(o.c.last_residual),
           "sample": "1 frame of the %dx%d half tank from rest, the reference's IC(0) run to tol 1e-6 (cap lifted), -O3 -ffast-math -march=native, single thread; "
                     "its iteration count grows with N (445 / 880 / 1726 at 512 / 1024 / 2048), so the rate at 8192 is ~8x lower" % (n, n)}
    o.close()
    return out


# ------------------------------------------------------------------------------------------------ the ONE line
LINE_LIMIT = 8000      # bytes; the driver keeps a bounded tail of stdout (round 3's 25.7 KB line came back unparsed)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(x, nd=4):
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x)) if abs(x) >= 1 else round(x, nd + 2)
    return x


def _short(x):
    """numbers to 6 significant digits, recursively"""
    if isinstance(x, dict):
        return {k: _short(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v) for v in x]
    if isinstance(x, float):
        return float("%.6g" % x)
    return x


ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_cell",
             "algorithmic_bytes_per_launch", "avg_launch_us", "launches", "measured_copy_GBps")
ITER_KEYS = ("us_per_iteration", "bytes_per_cell_iteration", "classes", "launches_per_iteration", "GBps_active", "frac_active", "GBps_traffic", "frac_traffic")


def _block(b, extra=()):
    """the compact form of a summarize() block"""
    if not isinstance(b, dict):
        return None
    if "error" in b:
        return {"error": str(b["error"])[:160]}
    out = _pick(b, ("value", "ms_per_step", "steps", "substeps", "pcg_iterations", "fluid_cells", "tol", "iterations_per_solve", "n_gpus") + tuple(extra))
    if isinstance(b.get("roofline"), dict):
        out["roofline"] = _pick(b["roofline"], ("kernel", "frac", "achieved", "avg_launch_us", "algorithmic_bytes_per_cell", "traffic", "traffic_over_algorithmic"))
    if isinstance(b.get("pcg_iteration"), dict):
        out["pcg_iteration"] = _pick(b["pcg_iteration"], ("us_per_iteration", "bytes_per_cell_iteration", "frac_active", "frac_traffic"))
    return out


def compact_line(full, limit=LINE_LIMIT):
    """The driver-facing line: the contract's keys, `roofline`, `cpu_baseline`, `pcg_iteration`, `kernels`, the converged block and one-number summaries of the
    secondary blocks - under `limit` bytes whatever the full object holds (which goes to bench_full.json).  Pure function of `full` (tests/test_bench_line.py)."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    line["vs_baseline"] = full.get("vs_baseline")
    cfg = dict(full.get("config") or {})
    cfg.pop("parallelism_detail", None)
    for k in ("workload", "parallelism"):
        if isinstance(cfg.get(k), str) and len(cfg[k]) > 200:
            cfg[k] = cfg[k][:197] + "..."
    line["config"] = cfg
    line.update(_pick(full, ("substeps", "pcg_iterations", "cells_substeps_per_s", "fluid_cells", "markers")))
    roof = full.get("roofline")
    line["roofline"] = _pick(roof, ROOF_KEYS) if isinstance(roof, dict) else None
    if isinstance(roof, dict) and "traffic" not in line["roofline"]:
        line["roofline"]["traffic"] = None
    line["pcg_iteration"] = _pick(full.get("pcg_iteration"), ITER_KEYS) or None
    line["kernels"] = {k: _pick(v, ("avg_us", "launches", "bytes_per_cell", "GBps_active", "GBps_traffic"))
                       for k, v in (full.get("kernels") or {}).items() if isinstance(v, dict) and v.get("bytes_per_cell") is not None}
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "seconds", "strict_ieee_value", "cpu_model", "host_cores_available", "error"))
        if isinstance(c.get("sample"), str) and len(c["sample"]) > 240:
            c["sample"] = c["sample"][:237] + "..."
        if isinstance(cpu.get("equal_tolerance"), dict):
            c["equal_tolerance"] = _pick(cpu["equal_tolerance"], ("value", "seconds", "pcg_iterations", "substeps", "error"))
        if isinstance(cpu.get("configs0_100x40_block_100_steps"), dict):
            c["configs0_value"] = cpu["configs0_100x40_block_100_steps"].get("value")
        if isinstance(cpu.get("reference_main_c_100x40_block_100_steps"), dict) and "value" in cpu["reference_main_c_100x40_block_100_steps"]:
            c["reference_main_c_100x40_value"] = cpu["reference_main_c_100x40_block_100_steps"]["value"]      # (the compiled reference itself: kind "reference")
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    conv = full.get("converged")
    if isinstance(conv, dict):
        c = _block(conv, ("last_residual", "cells_substeps_per_s"))
        if isinstance(conv.get("roofline"), dict):
            c["roofline"] = _pick(conv["roofline"], ROOF_KEYS)
        if isinstance(conv.get("pcg_iteration"), dict):
            c["pcg_iteration"] = _pick(conv["pcg_iteration"], ITER_KEYS)
        if isinstance(conv.get("deviation_vs_reference_converged"), dict):
            c["deviation_vs_reference_converged"] = _pick(conv["deviation_vs_reference_converged"],
                                                          ("state", "max_abs_du", "max_abs_dv", "max_abs_velocity", "dp_over_max_p", "fluid_cells_differing", "pcg_iterations", "error"))
        if isinstance(conv.get("cpu_baseline_equal_tolerance"), dict):
            c["cpu_baseline_equal_tolerance"] = _pick(conv["cpu_baseline_equal_tolerance"], ("value", "unit", "cores", "kind", "seconds", "pcg_iterations", "error"))
        line["converged"] = c
    summary = {}
    q = full.get("quality")
    if isinstance(q, dict) and isinstance(q.get("modes"), dict):
        summary["quality_100_iterations"] = {"pressure_error_vs_converged": {m: v.get("pressure_error") for m, v in q["modes"].items()},
                                             "solve_ms": {m: v.get("ms") for m, v in q["modes"].items()},
                                             "converged_iterations": (q.get("converged") or {}).get("iterations")}
    elif isinstance(q, dict) and "error" in q:
        summary["quality_100_iterations"] = {"error": str(q["error"])[:160]}
    sec = full.get("secondary") or {}
    if isinstance(sec.get("exact_ic0"), dict):
        summary["exact_ic0"] = _block(sec["exact_ic0"])
    if isinstance(sec.get("projection_16384"), dict):
        summary["projection_16384"] = _block(sec["projection_16384"])
    c1 = sec.get("configs1_1024_dam_break")
    if isinstance(c1, dict):
        b = _block(c1, ("roofline_mode_value", "roofline_mode_us_per_iteration", "f32_value", "f32_us_per_iteration", "multi_kernel_us_per_iteration"))
        if isinstance(c1.get("parity_in_run"), dict):
            b["parity_in_run"] = _pick(c1["parity_in_run"], ("frames", "max_abs_du", "max_abs_dv", "fluid_cells_differing"))
        summary["configs1_1024_dam_break"] = b
    tts = sec.get("time_to_solution")
    if isinstance(tts, dict):
        summary["time_to_solution_2048_ms"] = {k: v.get("ms") for k, v in tts.items() if isinstance(v, dict) and "ms" in v} or _pick(tts, ("error",))
    for k, v in full.items():
        if k.startswith("strong_") and isinstance(v, dict):
            b = _block(v, ("scaling", "setup_and_preroll_seconds"))
            if isinstance(v.get("converged_frames_multilevel"), dict):
                b["converged"] = _pick(v["converged_frames_multilevel"], ("value", "ms_per_step", "substeps", "pcg_iterations", "error"))
            if isinstance(v.get("balance"), dict):
                b["balance_max_over_mean"] = v["balance"].get("max_over_mean")
            summary[k] = b
    line["summary"] = summary
    line.update(_pick(full, ("balance", "comm_calls_rank0", "comm", "device", "full", "timings_s", "dtype_note")))
    if isinstance(line.get("balance"), dict):
        line["balance"] = _pick(line["balance"], ("partition", "max_over_mean"))
    line = _short(line)
    for k in ("value", "ms_per_step"):      # (the contract's two numbers at full precision: value x ms_per_step is checkable)
        if k in full:
            line[k] = full[k]
    # the size guard: drop the least important parts until the line fits
    for drop in (("timings_s",), ("kernels",), ("summary", "time_to_solution_2048_ms"), ("summary", "configs1_1024_dam_break"), ("summary", "exact_ic0"),
                 ("summary", "projection_16384"), ("summary",), ("comm_calls_rank0",), ("balance",)):
        if len(json.dumps(line)) < limit:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) or {}
        d.pop(drop[-1], None)
    return line


def write_full(full):
    """the full object beside the line: bench_full.json at the repo root and (the GPU box's scratch that travels back) under gpurun_out/"""
    where = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "bench_full.json"), "w") as f:
                json.dump(full, f)
                f.write("\n")
            where = where or os.path.join(os.path.relpath(d, ROOT), "bench_full.json").replace("./", "")
        except OSError:
            pass
    return where


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
        try:
            comm_info["exchange_us"] = {"g1_edge_rows_and_pair": round(sim.exchange_latency(100, GX, 2), 2), "g2_scalar": round(sim.exchange_latency(100, 0, 1), 2)}
            l = sum(comm_info["exchange_us"].values())
            comm_info["exchange_us"]["per_iteration"] = round(l, 2)
        except Exception as e:
            comm_info["exchange_us"] = {"error": repr(e)}
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
                                   max_iterations=20000, pcg_poll_interval=32).load_half_tank()
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
                           "sample": "1 frame (%d substep(s), %d PCG iterations) of the %dx%d half tank - the headline workload at 1/%d of its "
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
