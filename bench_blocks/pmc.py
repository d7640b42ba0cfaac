"""bench_blocks.pmc - HBM traffic from rocprofv3's PMC counters: live passes of this very workload in child processes, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes.

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
from bench_blocks.bytes import KERNEL_OF_CLASS, TILE_MODES  # noqa: F401


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
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py")] + child_args
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
