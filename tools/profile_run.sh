#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace summary of bench.py, then two separate
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950: MI355X_MICROARCH.md §rocprofv3).
# Outputs land in gpurun_out/prof_*; tools/summarize_profile.py condenses them for profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
SIZE=${1:-8192}
STEPS=${2:-2}
WORKLOAD=${3:-half_tank}
PRECOND=${4:-ic0_tile}
EXTRA=${5:-}
OUT=gpurun_out/prof_${SIZE}_${WORKLOAD}_${PRECOND}
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --size $SIZE --workload $WORKLOAD --precond $PRECOND --steps $STEPS --warmup 1 --no-secondary --no-pmc --no-kernel-timing $EXTRA"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- python3 $ARGS > /dev/null 2> "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- python3 $ARGS > /dev/null 2> "$OUT/write.log"
python3 bench.py --size $SIZE --workload $WORKLOAD --precond $PRECOND --steps $STEPS --warmup 1 --no-secondary --no-pmc $EXTRA > "$OUT/bench_events.json" 2>> "$OUT/trace.log"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2>> "$OUT/trace.log"
# keep only small artefacts for the merge back
find "$OUT" -type f -size +4M -delete
tail -5 "$OUT/trace.log"
cat "$OUT/summary.md" | head -60
