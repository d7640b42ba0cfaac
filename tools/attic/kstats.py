#!/usr/bin/env python3
"""Average duration per kernel out of a rocprofv3 (7.2) results database.  usage: kstats.py DIR [name filter ...]"""
import glob, os, sqlite3, sys
dbs = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)
if not dbs:
    print("no *_results.db under", sys.argv[1]); sys.exit(0)
con = sqlite3.connect(dbs[0])
flt = sys.argv[2:]
for name, n, tot, avg, mn, mx in con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"):
    short = name.split("(")[0].replace("void ", "")
    if flt and not any(f in short for f in flt):
        continue
    print("   %-58s calls %6d  avg %9.2f us  total %9.2f ms  (min %.1f, max %.1f us)" % (short[:58], n, avg / 1e3, tot / 1e6, mn / 1e3, mx / 1e3))
