#!/bin/bash
cd "$(dirname "$0")/../../.."
export TMPDIR=/tmp
for g in 0 1; do
  if [ $g = 1 ]; then export EULER_BUILD_GATHER=1; fi
  OUT=gpurun_out/prof_bs_$g; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --size 8192 --steps 2 --warmup 1 --no-secondary --no-pmc --no-kernel-timing --no-strong --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.log
  python3 - <<P
import sqlite3, glob
db = glob.glob("$OUT/trace/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
for n, c, a in rows:
    if any(w in n for w in ("build_system", "cell_system", "bin_markers")): print("gather" if $g else "lds", n[:60], c, round(a, 1), "us")
P
done
