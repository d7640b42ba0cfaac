# round 6: the stages of a substep with and without the tile map (EU_NO_TILE_MAP=1: rounds 1-5's full passes), per kernel, under rocprofv3
export TMPDIR=/tmp
cd /root/repo
for cfg in "8192 half_tank" "16384 dam_break"; do
  for v in MAP NOMAP; do
    if [ $v = NOMAP ]; then export EU_NO_TILE_MAP=1; else unset EU_NO_TILE_MAP; fi
    d=gpurun_out/tm_${v}_${cfg%% *}
    rm -rf $d; mkdir -p $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 tools/micro/stage_bench.py $cfg > $d/out.txt 2>&1
    echo "== $cfg $v"; python3 - <<PY
import csv,glob
f=glob.glob('$d/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if any(k in n for k in ('advect', 'narrow', 'transpose', 'zero_bounds', 'extrapolate', 'build_system', 'velocity_update')): print(n[:44], r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
    find $d -type f -size +1M -delete
  done
done
