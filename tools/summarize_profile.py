#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel-trace + two PMC passes, SQLite `*_results.db` as written by
rocprofv3 7.2) into a markdown summary for profiles/.

usage: summarize_profile.py gpurun_out/prof_<size> > profiles/rNN_<what>.md
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate passes; on gfx950 FETCH_SIZE counts wide coalesced reads at 1/2 -> x2."""
import glob
import json
import os
import sqlite3
import statistics
import sys

out = sys.argv[1]
cells = None


def db(pattern):
    f = glob.glob(os.path.join(out, pattern))
    return sqlite3.connect(f[0]) if f else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("eu_resident::", "")
    return name.split("(")[0].replace("void ", "")[:64]


for name in ("bench_events.json", "bench_trace.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            bench = json.loads(lines[-1])
            cells = bench["config"]["grid"][0] * bench["config"]["grid"][1]
            break

print("# rocprofv3 summary: %s\n" % os.path.basename(out.rstrip("/")))
print("command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --size N --workload W --precond P --steps K --warmup 1 --no-secondary "
      "--no-pmc --no-kernel-timing` (+ one `--pmc FETCH_SIZE` and one `--pmc WRITE_SIZE` pass), see tools/profile_run.sh\n")
con = db("trace/*_results.db")
if con:
    print("## kernel trace (all launches, including the few idle early-exit launches after convergence)\n")
    print("| kernel | calls | total ms | avg us | median us | % |")
    print("|---|---|---|---|---|---|")
    rows = con.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc limit 22").fetchall()
    total = con.execute("select sum(duration) from kernels").fetchone()[0]
    for name, n, tot, avg in rows:
        d = [r[0] for r in con.execute("select duration from kernels where name=? order by duration", (name,))]
        print("| %s | %d | %.3f | %.2f | %.2f | %.2f |" % (short(name), n, tot / 1e6, avg / 1e3, statistics.median(d) / 1e3, 100.0 * tot / total))
    print("\nregisters / LDS of the sweep kernels:")
    for name, v, s, l in con.execute("select distinct name, vgpr_count, sgpr_count, lds_size from kernels where name like '%sweep%'"):
        print("- %s: %d VGPR, %d SGPR, %d B LDS" % (short(name), v, s, l))

fetch, write = db("pmc_fetch/*_results.db"), db("pmc_write/*_results.db")
if fetch and write:
    print("\n## HBM traffic per launch from PMC (separate passes)\n")
    q = "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name"
    f = {short(k): (n, v) for k, n, v in fetch.execute(q, ("FETCH_SIZE",))}
    w = {short(k): (n, v) for k, n, v in write.execute(q, ("WRITE_SIZE",))}
    print("| kernel | launches | FETCH_SIZE KiB (raw) | x2 corrected MB | WRITE_SIZE KiB | write MB | traffic MB/launch |")
    print("|---|---|---|---|---|---|---|")
    for k in sorted(f, key=lambda k: -f[k][0] * f[k][1])[:14]:
        fr = f[k][1]
        wr = w.get(k, (0, 0.0))[1]
        print("| %s | %d | %.0f | %.2f | %.0f | %.2f | %.2f |" % (k, f[k][0], fr, 2 * fr * 1024 / 1e6, wr, wr * 1024 / 1e6,
                                                              (2 * fr + wr) * 1024 / 1e6))
    print("\n(averages include idle early-exit launches, which move no data; active-launch traffic is slightly higher)")
    # machine-readable copy for bench.py's roofline.traffic (bytes per launch, corrected as above)
    traffic = {k: {"launches": f[k][0], "hbm_bytes_per_launch": round((2 * f[k][1] + w.get(k, (0, 0.0))[1]) * 1024)} for k in f}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 on gfx950, KiB -> bytes",
               "workload": os.path.basename(out.rstrip("/")).replace("prof_", ""), "kernels": traffic},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)

for name in ("bench_trace.json", "bench_events.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if not lines:
            continue
        d = json.loads(lines[-1])
        print("\n## %s (%s)\n" % (name, "under rocprofv3" if "trace" in name else "un-profiled, HIP-event kernel timing"))
        print("value %.4g %s, ms_per_step %.2f, substeps %d, pcg_iterations %d" % (
            d["value"], d["unit"], d["ms_per_step"], d["substeps"], d["pcg_iterations"]))
        if d.get("roofline"):
            print("\nroofline: `%s`" % json.dumps(d["roofline"]))
        if d.get("kernels"):
            print()
            for k, v in d["kernels"].items():
                print("- %s: %s" % (k, v))
