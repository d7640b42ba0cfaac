#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time python -m pytest tests/test_slab.py tests/test_slab_host.py tests/test_slab_rows.py tests/test_snapshot.py tests/test_sweep_isa.py -m gpu -q -x ) 2>&1 | tail -12 > $O/gputest_tail.txt; cat $O/gputest_tail.txt
# the driver's N = 2 command with both ranks on ONE GPU (gloo; RCCL refuses two ranks on a device and the run falls back to the torch callbacks): functional record of the default N > 1 path at full size
export EULER_DIST_BACKEND=gloo EULER_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 ) > $O/bench_rows_2ranks_1gpu.json 2> $O/bench_rows_2ranks_1gpu.err
tail -5 $O/bench_rows_2ranks_1gpu.err
python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_rows_2ranks_1gpu.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['scaling'], d['config']['grid'], d['balance'])
print('strong', {k:d['strong_16384_dam_break'][k] for k in ('value','ms_per_step','n_gpus','balance','hbm_bytes_this_rank','setup_and_preroll_seconds')} if d.get('strong_16384_dam_break') and 'error' not in d['strong_16384_dam_break'] else d.get('strong_16384_dam_break'))
print('cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None, 'hbm', d['hbm_bytes_this_rank'])
print(d['config']['parallelism'])
P
