#!/bin/bash
# development aid: a long run of N row-slab ranks SHARING one GPU (gloo) beside the single-GPU run, summarised
cd "$(dirname "$0")/.."
export EULER_DIST_BACKEND=gloo EULER_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-3}; X=${2:-256}; Y=${3:-384}; W=${4:-golden:waterfall}; F=${5:-200}; P=${6:-2}
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29571 tests/slab_rows_worker.py $X $Y $W $F $P > gpurun_out/soak_slab.json 2> gpurun_out/soak_slab.err
echo rc=$?
python3 - <<'P'
import json
d = json.loads(open("gpurun_out/soak_slab.json").read().strip().splitlines()[-1])
fr = d["frames"]
exact = [f["count_differ"] == 0 and f["markers_at_keys"] for f in fr]
first_bad = exact.index(False) if False in exact else None
print("ranks", d["world"], "frames", len(fr), "exact frames", sum(exact), "first inexact", first_bad)
print("keys permutation in every frame:", all(f["keys_are_a_permutation"] for f in fr), " markers in own rows:", all(f["markers_in_rows"] for f in fr),
      " rng equal:", all(f["rng"][0] for f in fr))
print("markers (single GPU, slabs) at the end:", fr[-1]["n_markers"][:2], " substeps last frame:", fr[-1]["substeps"], " iterations:", fr[-1]["iters"])
print("max du / dv / dp over the exact frames:", max([f["du"] for f, e in zip(fr, exact) if e] or [0]), max([f["dv"] for f, e in zip(fr, exact) if e] or [0]), max([f["dp"] for f, e in zip(fr, exact) if e] or [0]))
capped = [i for i, f in enumerate(fr) if f["iters"][0] >= 100]
print("first frame whose last solve hit the iteration cap:", capped[0] if capped else None)
P
tail -3 gpurun_out/soak_slab.err
