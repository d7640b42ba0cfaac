#!/bin/bash
# First contact with a multi-GPU node (nothing of the 1 -> 8 curve has been measured yet).  Launches ranks with torch.distributed.run from a shell that
# never touches the GPU (never re-exec a GPU process).  Order: (1) the exchange latency L over the library's RCCL communicator - DESIGN 7a's only unknown - with
# min(device count, 8) ranks; (2) the driver's scaling line at that N (weak line + strong_16384_dam_break block, compact: bench.py prints < 8 KB); (3) N = 1 for the denominator.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0} MASTER_ADDR=127.0.0.1
N=$(python3 -c 'import torch; print(min(torch.cuda.device_count(), 8))')      # (device_count does not initialise the GPU on this image)
OUT=${1:-gpurun_out/node_first_contact}
mkdir -p "$OUT"
echo "node_first_contact: $N GPU(s)"
if [ "$N" -lt 2 ]; then
  echo "one GPU: the exchange probe runs with a single rank (RCCL refuses two ranks on one device); the scaling runs are skipped"
fi
PORT=29720
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT tools/exchange_latency.py 16384 \
  > "$OUT/exchange_latency.json" 2> "$OUT/exchange_latency.log" || echo "exchange probe failed (see $OUT/exchange_latency.log)"
tail -1 "$OUT/exchange_latency.json"
if [ "$N" -ge 2 ]; then
  for G in $N 1; do
    PORT=$((PORT + 1))
    if [ "$G" -gt 1 ]; then
      timeout 1800 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$G" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$G" --steps 10 --warmup 3 \
        > "$OUT/bench_gpus$G.json" 2> "$OUT/bench_gpus$G.log"
    else
      timeout 1800 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-pmc > "$OUT/bench_gpus$G.json" 2> "$OUT/bench_gpus$G.log"
    fi
    cp bench_full.json "$OUT/bench_full_gpus$G.json" 2>/dev/null
    python3 - "$OUT/bench_gpus$G.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d.get("summary", {})
st = next((v for k, v in s.items() if k.startswith("strong_")), {})
print("n_gpus %d: weak value %.3g cells*steps/s (%.1f ms/step, roofline.frac %s); strong block: %s cells*steps/s, %s ms/step, converged %s"
      % (d["n_gpus"], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), st.get("value"), st.get("ms_per_step"), (st.get("converged") or {}).get("value")))
P
  done
fi
