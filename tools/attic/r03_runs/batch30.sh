#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tile_precond.py -m gpu -q -x -s -k "multilevel" 2>&1 | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
python - <<'P'
import sys, time
sys.path.insert(0, ".")
import euler_amd as ea, os
for n in (1024, 2048, 4096):
    for fuse in (1, 0):
        # (the switch is read once per process: children)
        import subprocess
        code = "import sys,time; sys.path.insert(0,'.'); import euler_amd as ea\ns=ea.Simulation(%d,%d,dot_mode=ea.DOT_TREE,precond=ea.PRECOND_IC0_TILE_MG,max_iterations=20000,pcg_poll_interval=32).load_half_tank(); s.step(); s.load_half_tank(); t0=time.perf_counter(); s.step(); st=s.stats(); print(%d, 'fuse' if %d else 'nofuse', round(1e3*(time.perf_counter()-t0),2), 'ms', st.last_pcg_iterations, 'its', st.last_substeps, 'substeps')" % (n, n, n, fuse)
        env = dict(os.environ)
        if not fuse: env["EULER_MG_NO_FUSE"] = "1"
        print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
P
timeout 600 python bench.py --no-pmc --no-strong --no-cpu-baseline --no-secondary --precond ic0_tile_mg > $O/bench_mg_main.json 2> $O/bench_mg_main.err; tail -1 $O/bench_mg_main.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_mg_main.json"))
print(d["value"], d["ms_per_step"], json.dumps(d.get("pcg_iteration")))
P
