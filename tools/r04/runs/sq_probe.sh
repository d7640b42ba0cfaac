#!/bin/bash
# SQ counters of the two PCG kernels (8192^2 headline workload): are they VALU-issue bound?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_sq; mkdir -p gpurun_out/prof_sq
ARGS="bench.py --steps 1 --warmup 0 --no-secondary --no-pmc --no-cpu-baseline --no-kernel-timing --preroll solves"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace -d gpurun_out/prof_sq/a -o a -- python3 $ARGS > /dev/null 2> gpurun_out/prof_sq/a.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof_sq/b -o b -- python3 $ARGS > /dev/null 2> gpurun_out/prof_sq/b.log
python3 - <<'P'
import glob, sqlite3
for tag in "ab":
    f = glob.glob("gpurun_out/prof_sq/%s/*_results.db" % tag)
    if not f:
        print(tag, "no db"); continue
    con = sqlite3.connect(f[0])
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_search_apply%' or kernel_name like '%k_precond_tile%' group by kernel_name, counter_name").fetchall()
    for k, c, n, v in rows:
        print("%-48s %-22s %6d %16.1f" % (k.split("(")[0].replace("void ", "")[:48], c, n, v))
    d = con.execute("select name, count(*), avg(duration) from kernels where name like '%k_search_apply%' or name like '%k_precond_tile%' group by name").fetchall()
    for k, n, v in d:
        print("%-48s duration us %.1f (%d)" % (k.split("(")[0].replace("void ", "")[:48], v / 1e3, n))
P
tail -2 gpurun_out/prof_sq/a.log
rm -rf gpurun_out/prof_sq/a gpurun_out/prof_sq/b
