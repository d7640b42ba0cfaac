#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
export EULER_DIST_BACKEND=gloo EULER_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 ) > $O/bench_rows_2ranks_1gpu.json 2> $O/bench_rows_2ranks_1gpu.err
tail -4 $O/bench_rows_2ranks_1gpu.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_rows_2ranks_1gpu.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['scaling'], d['config']['grid'], d['balance'])
s=d['strong_16384_dam_break']
print('strong', {k:s[k] for k in ('value','ms_per_step','n_gpus','hbm_bytes_this_rank','setup_and_preroll_seconds')} if s and 'error' not in s else s)
print('converged', s.get('converged_frames_multilevel') if s else None)
print('cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None, d.get('comm_calls_rank0'))
P
