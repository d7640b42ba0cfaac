#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_frame; mkdir -p gpurun_out/prof_frame
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_frame -o fp -- python3 tools/r04/frame_probe.py ${1:-16384} ${2:-4} > gpurun_out/frame_probe.txt 2>&1
grep -v "^[WE]2026" gpurun_out/frame_probe.txt | tail -3
python3 - <<'P'
import glob, sqlite3
con = sqlite3.connect(glob.glob("gpurun_out/prof_frame/*_results.db")[0])
rows = con.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc limit 36").fetchall()
total = con.execute("select sum(duration) from kernels").fetchone()[0]
print("total kernel ms %.1f" % (total / 1e6))
for name, n, tot, avg in rows:
    print("%-64s %7d %10.3f ms %9.2f us %5.1f%%" % (name.split("(")[0].replace("void ", "")[:64], n, tot / 1e6, avg / 1e3, 100.0 * tot / total))
P
rm -rf gpurun_out/prof_frame
