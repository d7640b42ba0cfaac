"""bench.py's ONE stdout line stays readable by the driver: round 3's line had grown to 25.7 KB and came back unparsed (BENCH_r03.json: parsed null).
compact_line() is a pure function of the full result object; the full object goes to bench_full.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_full(noise=0):
    kern = {"apply_a": {"ms_total": 1015.063, "launches": 3200, "avg_us": 317.21, "per_iteration": True, "bytes_per_cell": 45.0, "GBps_active": 4746.6,
                        "traffic_bytes_per_launch": 1708143240, "GBps_traffic": 5384.9},
            "precond_tile": {"ms_total": 696.341, "launches": 3232, "avg_us": 215.45, "per_iteration": True, "bytes_per_cell": 33, "GBps_active": 5124.8},
            "update_search": {"ms_total": 7.295, "launches": 32, "avg_us": 227.98, "per_iteration": False}}
    roof = {"bound": "hbm", "kernel": "apply_a", "achieved": 4746.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.5933, "traffic": 1708143240, "frac_traffic": 0.6731,
            "traffic_over_algorithmic": 1.134, "frac_dense": 1.19, "achieved_is": "x" * 120, "algorithmic_bytes_per_cell": 45.0,
            "algorithmic_bytes_per_launch": 1505712825, "avg_launch_us": 317.21, "launches": 3200, "fluid_fraction": 0.4986, "traffic_source": "y" * 300, "note": "z" * 200,
            "measured_copy_GBps": 5301.2}
    it = {"us_per_iteration": 539.57, "bytes_per_cell_iteration": 78.0, "classes": {"apply_a": 45.0, "precond_tile": 33}, "complete": True, "launches_per_iteration": 2,
          "GBps_active": 4836.8, "frac_active": 0.6046, "GBps_traffic": 5189.2, "frac_traffic": 0.6487, "frac_dense": 1.2127}
    blk = {"mode": "m" * 150, "value": 1.2e8, "unit": "cells*steps/s", "ms_per_step": 560.1, "steps": 4, "substeps": 32, "pcg_iterations": 3200, "cells_substeps_per_s": 9.6e8,
           "fluid_cells": 33460285, "markers": 134086686, "last_residual": 3000.5, "roofline": roof, "pcg_iteration": it, "kernels": kern, "workload": "w" * 200}
    full = {"metric": "cells*steps/sec of sim_step() + pressure-solve HBM GB/s vs roofline", "value": 136094011.123456789, "unit": "cells*steps/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 493.1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 PCG vectors, f32 fields", "data": "synthetic",
            "config": {"workload": "8192x8192 half_tank " + "c" * 400, "grid": [8192, 8192], "preroll_frames": 10, "precond": "ic0_tile", "tile_records": 16, "dot_mode": "tree",
                       "max_iterations": 100, "tol": 0.0, "parallelism": "1 GPU"},
            "mode": "roofline mode", "substeps": 160, "pcg_iterations": 16000, "cells_substeps_per_s": 1.08e9, "markers": 134086686, "fluid_cells": 33460285,
            "hbm_bytes_this_rank": 12345678901, "roofline": roof, "pcg_iteration": it, "balance": None, "comm_calls_rank0": None, "kernels": kern,
            "cpu_baseline": {"value": 687276.0, "unit": "cells*steps/s", "cores": 1, "kind": "port", "sample": "s" * 400, "seconds": 6.1, "strict_ieee_value": 758689.2,
                             "cpu_model": "AMD EPYC 9575F 64-Core Processor", "extrapolated_seconds_per_substep": {"8192x8192": 97.6, "note": "n" * 200},
                             "configs0_100x40_block_100_steps": {"value": 812895.0, "seconds": 0.49}, "host_cores_available": 256,
                             "equal_tolerance": {"value": 80000.0, "seconds": 13.1, "pcg_iterations": 880, "substeps": 1, "sample": "q" * 300}},
            "converged": dict(blk, tol=1e-6, iterations_per_solve=185.0, deviation_vs_reference_converged={"state": "1024x1024 dam break", "max_abs_du": 1e-5, "max_abs_dv": 2e-5,
                                                                                                           "max_abs_velocity": 796.0, "dp_over_max_p": 1e-9, "fluid_cells_differing": 0,
                                                                                                           "pcg_iterations": [800, 2900], "vs": "v" * 200},
                              cpu_baseline_equal_tolerance={"value": 80000.0, "unit": "cells*steps/s", "cores": 1, "kind": "port", "seconds": 13.1, "pcg_iterations": 880}),
            "quality": {"modes": {m: {"ms": 55.2, "iterations": 100, "residual": 6351.7, "pressure_error": 0.98} for m in ("ic0", "ic0_tile", "ic0_tile2", "ic0_tile_mg")},
                        "converged": {"iterations": 193}, "equal_residual": {"residual_scan": [[i, 1234.5678 * i] for i in range(noise)]}},
            "strong_16384_dam_break": dict(blk, n_gpus=1, scaling="strong", setup_and_preroll_seconds=8.3, balance=None,
                                           converged_frames_multilevel={"value": 1.1e9, "ms_per_step": 240.6, "substeps": 4, "pcg_iterations": 100}),
            "secondary": {"exact_ic0": blk, "projection_16384": blk, "configs1_1024_dam_break": dict(blk, parity_in_run={"frames": 2, "max_abs_du": 1e-13, "max_abs_dv": 0.0,
                                                                                                                       "fluid_cells_differing": 0, "vs": "o" * 100},
                                                                                                     roofline_mode_value=5.7e7, roofline_mode_us_per_iteration=31.58),
                          "time_to_solution": {m: {"ms": 12.5, "pcg_iterations": 118} for m in ("ic0", "ic0_tile", "ic0_tile2", "ic0_tile_mg")},
                          "parity_vs_reference_ic0": [{"residual_scan": [[i, 0.123456789 * i] for i in range(noise)]} for _ in range(4)]},
            "device": "AMD Instinct MI355X", "timings_s": {"a": 1.0, "b": 2.0}, "full": "bench_full.json"}
    # round 6: kernel time by class per substep (the stages around the iterations are half of the converged frame)
    full["stages"] = {"ms_per_substep": {"apply_a": 3.17, "precond_tile": 2.15, "advect_bin": 1.04, "build_system": 0.48, "velocity_update": 0.66, "transpose": 0.28,
                                         "narrow_counts": 0.05, "dt": 0.01}, "non_pcg_ms_per_substep": 4.0, "pcg_ms_per_substep": 45.9}
    full["converged"]["stages"] = dict(full["stages"])
    full["strong_16384_dam_break"]["converged_frames_multilevel"]["stages"] = {"pcg_ms_per_substep": 18.2, "non_pcg_ms_per_substep": 12.1, "non_pcg_share_of_kernel_time": 0.399, "non_pcg_share_of_frame": 0.387,
                                                                               "ms_per_substep": {"advect_bin": 4.1}}
    return full


def test_compact_line_is_small_and_carries_the_contract():
    for noise in (0, 50, 2000):
        full = canned_full(noise)
        line = json.dumps(bench.compact_line(full))
        assert len(line) < 8192, len(line)
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in d, k
        assert d["config"]["workload"].startswith("8192x8192 half_tank")
        assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
        assert d["roofline"]["unit"] == "GB/s" and d["roofline"]["traffic"] == 1708143240
        assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
        assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["sample"]
        assert d["converged"]["roofline"]["frac"] > 0 and d["converged"]["cpu_baseline_equal_tolerance"]["value"] > 0
        assert d["converged"]["deviation_vs_reference_converged"]["fluid_cells_differing"] == 0
        assert d["pcg_iteration"]["bytes_per_cell_iteration"] == 78.0
        assert d["summary"]["quality_100_iterations"]["pressure_error_vs_converged"]["ic0"] == 0.98
        assert d["stages"]["non_pcg_ms_per_substep"] == 4.0 and len(d["stages"]["ms_per_substep"]) == 6 and "dt" not in d["stages"]["ms_per_substep"]
        assert d["converged"]["stages"]["non_pcg_ms_per_substep"] == 4.0
        assert d["summary"]["strong_16384_dam_break"]["converged"]["stages"]["non_pcg_share_of_kernel_time"] == 0.399
        assert d["summary"]["strong_16384_dam_break"]["converged"]["stages"]["non_pcg_share_of_frame"] == 0.387


def test_compact_line_of_an_eight_gpu_job_carries_the_communicator():
    """N = 8: the line of a row-slab job also holds what the transport connected (`comm.ranks`) and the measured latency of the two exchange
    points of a distributed PCG iteration (`comm.exchange_us`: the one unknown of the scaling model), the fluid balance and the call counters - and still fits."""
    full = canned_full(50)
    full.update({"n_gpus": 8, "scaling": "weak",
                 "balance": {"partition": [[0, 128], [128, 256], [256, 384], [384, 512], [512, 640], [640, 768], [768, 896], [896, 1024]],
                             "fluid_cells_per_rank": [33460285] * 8, "max_over_mean": 1.0, "note": "b" * 300},
                 "comm_calls_rank0": {"allreduce": 1621, "halo": 3362, "chain": 0, "allgather": 480, "exchange": 32320},
                 "comm": {"ranks": 8, "transport": "rccl", "rccl_version": 22703, "p2p_mailboxes": False,
                          "exchange_us": {"g1_edge_rows_and_pair": 14.82, "g2_scalar": 9.31, "per_iteration": 24.13}}})
    full["config"]["parallelism"] = "8 row slabs (8192 rows each), EVERY stage decomposed; tile-local IC(0): no coupling between slabs; exchanges: RCCL"
    full["config"]["parallelism_detail"] = "p" * 600
    full["strong_16384_dam_break"] = dict(full["strong_16384_dam_break"], n_gpus=8, balance={"partition": [[i, i + 32] for i in range(0, 256, 32)], "max_over_mean": 1.08, "x": "y" * 200})
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 8192, len(line)
    d = json.loads(line)
    assert d["n_gpus"] == 8 and d["comm"]["ranks"] == 8 and d["comm"]["transport"] == "rccl"
    assert d["comm"]["exchange_us"]["per_iteration"] == 24.13
    assert d["comm_calls_rank0"]["exchange"] == 32320 and d["balance"]["max_over_mean"] == 1.0
    assert d["summary"]["strong_16384_dam_break"]["balance_max_over_mean"] == 1.08


def test_compact_line_survives_missing_and_failed_blocks():
    full = canned_full()
    full["converged"] = {"error": "RuntimeError('boom')"}
    full["secondary"] = {"exact_ic0": {"error": "x" * 1000}}
    full["cpu_baseline"] = None
    full["quality"] = {"error": "nope"}
    full["roofline"]["traffic"] = None
    d = json.loads(json.dumps(bench.compact_line(full)))
    assert d["cpu_baseline"] is None and d["roofline"]["traffic"] is None
    assert "error" in d["converged"] and len(json.dumps(d)) < 8192


def test_the_guard_drops_parts_rather_than_overflow():
    full = canned_full()
    full["kernels"] = {"class_%d" % i: {"avg_us": 1.0, "launches": 2, "bytes_per_cell": 3.0, "GBps_active": 4.0} for i in range(400)}
    d = bench.compact_line(full)
    assert len(json.dumps(d)) < 8192 and "kernels" not in d and d["roofline"]["frac"] > 0


def test_iteration_bytes_are_the_sum_of_the_launches():
    # the byte accounting follows the kernels (round 3 credited the parity mode 145 B for launches that declared 120); since k_search_apply no longer
    # stores A s' the launches declare 8 B less - except in the configurations that keep the stored form (sequential dots, mailboxes)
    assert bench.PCG_BYTES["ic0"] == 25 + 25 + 34 + 25 == 109      # (p += alpha s on every EIGHTH pass: 1.125 w instead of 1.5 w)
    assert bench.PCG_BYTES["ic0_tile"] == 34 + 33 == 67
    for mode, classes in bench.ITER_BYTES.items():
        assert abs(bench.PCG_BYTES[mode] - sum(classes.values())) < 1e-12
    try:
        bench.set_as_stored(True, 2)
        assert bench.PCG_BYTES["ic0"] == 120 and bench.PCG_BYTES["ic0_tile"] == 78
        bench.set_as_stored(False, 2)
        assert bench.PCG_BYTES["ic0_tile"] == 70
    finally:
        bench.set_as_stored(False)


def test_the_pmc_passes_run_bench_py_itself(monkeypatch):
    """The live PMC passes are child processes of the bench run: `rocprofv3 --pmc ... -- python bench.py <args> --pmc-child`.  The script they name must be bench.py (after the
    split into bench_blocks/ it was the module's own file for a while, and `roofline.traffic` came back null)."""
    import bench_blocks.pmc as pmc
    seen = []

    class Done:
        returncode, stderr, stdout = 1, "stop here", ""

    monkeypatch.setattr(pmc.shutil, "which", lambda name: "/opt/rocm/bin/rocprofv3")
    monkeypatch.setattr(pmc.subprocess, "run", lambda cmd, **kw: (seen.append(cmd), Done())[1])
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_TOOL"))] + ["LD_PRELOAD"]:
        monkeypatch.delenv(k, raising=False)
    traffic, note = pmc.pmc_live(["--size", "64", "--pmc-child"])
    assert traffic is None and "failed" in note
    cmd = seen[0]
    i = cmd.index("--")
    assert cmd[:3] == ["/opt/rocm/bin/rocprofv3", "--pmc", "FETCH_SIZE"] and not any(a in cmd for a in ("--sys-trace", "-s", "--hip-trace", "--runtime-trace"))
    assert os.path.basename(cmd[i + 2]) == "bench.py" and os.path.samefile(cmd[i + 2], os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    assert cmd[i + 3:] == ["--size", "64", "--pmc-child"]
