#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
# N = 1: default line incl. the strong denominator with its converged-frames block
( time timeout 1500 python bench.py --no-pmc --no-cpu-baseline > $O/bench_strong1.json 2> $O/bench_strong1.err ) 2>&1 | tail -3
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_strong1.json"))
b=d["strong_16384_dam_break"]
print({k:b[k] for k in ("value","ms_per_step","substeps","pcg_iterations")}, b["converged_frames_multilevel"])
P
export EULER_DIST_BACKEND=gloo EULER_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --strong-size 4096 --no-cpu-baseline ) > $O/bench_rows_2ranks_strongmg.json 2> $O/bench_rows_2ranks_strongmg.err
tail -3 $O/bench_rows_2ranks_strongmg.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_rows_2ranks_strongmg.json') if l.startswith('{')][-1])
k=[x for x in d if x.startswith("strong_")][0]
b=d[k]
print(k, {q:b[q] for q in ("value","ms_per_step","substeps","pcg_iterations")} if "error" not in b else b, b.get("converged_frames_multilevel"))
P
