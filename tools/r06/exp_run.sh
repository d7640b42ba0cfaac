export TMPDIR=/tmp
cd /root/repo
for v in BASE; do
  if [ $v = BASE ]; then unset EULER_HIP_LIB; else export EULER_HIP_LIB=/root/repo/tools/micro/lib_ablate/libeuler_hip_$v.so; fi
  rm -rf gpurun_out/exp_$v; mkdir -p gpurun_out/exp_$v
  rocprofv3 --kernel-trace --stats -d gpurun_out/exp_$v -o t -- python3 tools/micro/stage_bench.py 8192 > gpurun_out/exp_$v/out.txt 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/exp_$v/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'advect' in n or 'bin' in n or 'transpose' in n: print(n[:40], r['Calls'], r['AverageNs'])
PY
  find gpurun_out/exp_$v -type f -size +1M -delete
done
