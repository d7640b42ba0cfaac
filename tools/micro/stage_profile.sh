#!/bin/bash
# development aid: kernel-trace + PMC of tools/micro/stage_bench.py (the marker / velocity stages of one substep, repeated)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/stage_bench; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 tools/micro/stage_bench.py "$@" > $OUT/out.txt 2> $OUT/t.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python3 tools/micro/stage_bench.py "$@" > /dev/null 2> $OUT/f.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python3 tools/micro/stage_bench.py "$@" > /dev/null 2> $OUT/w.log
cat $OUT/out.txt
python3 tools/summarize_profile.py $OUT 2>/dev/null | grep -E "k_advect_markers_a|k_bin_markers|k_advect_velocity|k_extrapolate|k_zero|k_rotate|k_narrow|k_sel|k_compact"
