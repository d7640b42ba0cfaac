# round 6: the multilevel mode's per-solve set-up kernels under rocprofv3 (16384^2 dam break, every solve to 1e-6)
export TMPDIR=/tmp
cd /root/repo
d=gpurun_out/mg_setup; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --size ${1:-16384} --workload dam_break --precond ic0_tile_mg --tol 1e-6 --max-iterations 20000 --preroll solves --steps 1 --warmup 1 --no-secondary --no-pmc --no-kernel-timing --no-cpu-baseline > $d/out.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$d/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if any(k in n for k in ('k_mg_coarsen', 'k_mg_assemble0', 'k_mg_convert0', 'k_mg_inner0', 'k_mg_wd', 'fillBuffer', 'k_factor_tile', 'k_coarse')): print(n[:44], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), round(float(r['TotalDurationNs']) / 1e6, 2))
PY
find $d -type f -size +1M -delete
