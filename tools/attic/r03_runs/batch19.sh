#!/bin/bash
# the driver's N = 4 command with all ranks on ONE GPU (gloo; exchanges by the torch callbacks incl. the fused exchange): functional record
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
export EULER_DIST_BACKEND=gloo EULER_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 2 --warmup 1 ) > $O/bench_rows_4ranks_1gpu.json 2> $O/bench_rows_4ranks_1gpu.err
tail -6 $O/bench_rows_4ranks_1gpu.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_rows_4ranks_1gpu.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['scaling'], d['config']['grid'], d['balance'])
st=d.get('strong_16384_dam_break')
print('strong', {k:st[k] for k in ('value','ms_per_step','n_gpus','balance','hbm_bytes_this_rank','setup_and_preroll_seconds')} if st and 'error' not in st else st)
print('cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None, 'hbm', d['hbm_bytes_this_rank'])
P
