#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gap; mkdir -p gpurun_out/prof_gap
python3 tools/r04/gap_probe.py ${1:-1024} ${2:-6} 2>&1 | grep -v "^[WE]2026" | tail -2
rocprofv3 --kernel-trace -d gpurun_out/prof_gap -o g -- python3 tools/r04/gap_probe.py ${1:-1024} ${2:-6} > gpurun_out/gap_probe.txt 2>&1
grep -v "^[WE]2026" gpurun_out/gap_probe.txt | tail -2
python3 - ${2:-6} <<'P'
import glob, sqlite3, sys, collections
frames = int(sys.argv[1])
con = sqlite3.connect(glob.glob("gpurun_out/prof_gap/*_results.db")[0])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# the timed frames are the tail: take the last `frames` frames = from the (frames*?)-th k_maxsq from the end... use the last 40% of the k_dt launches' region
names = [r[0].split("(")[0].replace("void ", "") for r in rows]
dts = [i for i, n in enumerate(names) if n.startswith("k_maxsq")]
# substeps of the timed region: count back substeps until ~frames*4.. just take the last 20 substeps
first = dts[-21]
last = dts[-1]
seg = rows[first:last]
segn = names[first:last]
wall = seg[-1][2] - seg[0][1]
busy = sum(e - s for _, s, e in seg)
print("20 substeps: wall %.1f us per substep, kernels %.1f us per substep (%d launches per substep), gaps %.1f us per substep" % (wall / 20e3, busy / 20e3, len(seg) / 20, (wall - busy) / 20e3))
gap_after = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.defaultdict(float)
for i in range(len(seg) - 1):
    g = seg[i + 1][1] - seg[i][2]
    gap_after[segn[i] + " -> " + segn[i + 1]] += max(g, 0); cnt[segn[i] + " -> " + segn[i + 1]] += 1
for i in range(len(seg)):
    dur[segn[i]] += seg[i][2] - seg[i][1]
print("-- largest gaps (us per substep)")
for k, v in sorted(gap_after.items(), key=lambda x: -x[1])[:24]:
    print("%-90s %8.1f  (%d per substep, %.1f us each)" % (k[:90], v / 20e3, cnt[k] / 20, v / cnt[k] / 1e3))
print("-- kernel time (us per substep)")
for k, v in sorted(dur.items(), key=lambda x: -x[1])[:40]:
    print("%-60s %8.1f" % (k[:60], v / 20e3))
P
rm -rf gpurun_out/prof_gap
