"""bench_blocks.line - the ONE stdout line: a pure function of the full result object that stays below the driver's limit (compact_line), and bench_full.json.

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
    st = full.get("stages")
    if isinstance(st, dict) and isinstance(st.get("ms_per_substep"), dict):      # the stages around the iterations: the six largest classes and the sums
        top = dict(list(st["ms_per_substep"].items())[:6])
        line["stages"] = {"ms_per_substep": top, "non_pcg_ms_per_substep": st.get("non_pcg_ms_per_substep"), "pcg_ms_per_substep": st.get("pcg_ms_per_substep")}
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = _pick(cpu, ("value", "unit", "cores", "kind", "cells_substeps_per_s", "substeps_per_step", "sample", "seconds", "strict_ieee_value", "cpu_model", "host_cores_available", "error"))
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
            # (no PMC figures here: the counters' per-launch averages of a converging run include the launches queued behind convergence, which move no data - VERDICT r5)
            c["roofline"] = _pick(conv["roofline"], tuple(k for k in ROOF_KEYS if "traffic" not in k))
        if isinstance(conv.get("pcg_iteration"), dict):
            c["pcg_iteration"] = _pick(conv["pcg_iteration"], ITER_KEYS)
        if isinstance(conv.get("stages"), dict):
            c["stages"] = _pick(conv["stages"], ("non_pcg_ms_per_substep", "pcg_ms_per_substep"))
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
                cs = v["converged_frames_multilevel"].get("stages")
                if isinstance(cs, dict):      # how the converged frame's kernel time divides: iterations / the stages around them
                    b["converged"]["stages"] = _pick(cs, ("pcg_ms_per_substep", "non_pcg_ms_per_substep", "non_pcg_share_of_kernel_time", "non_pcg_share_of_frame"))
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
