#!/bin/bash
cd "$(dirname "$0")/../../.."
O=gpurun_out/r03
mkdir -p $O
( time python -m pytest tests/test_slab.py tests/test_slab_host.py tests/test_slab_rows.py tests/test_snapshot.py tests/test_dist_gloo.py -m gpu -q -x ) 2>&1 | tail -8
# 2 ranks on ONE GPU (gloo, torch callbacks), the strong-scaling unit at reduced size in the multilevel mode with solves to tolerance: functional record
export EULER_DIST_BACKEND=gloo EULER_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for NP in 2 4; do
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node $NP --master-addr 127.0.0.1 --master-port 2962$NP bench.py --gpus $NP --steps 2 --warmup 1 --scaling strong --size 4096 --workload dam_break --precond ic0_tile_mg --tol 1e-6 --max-iterations 4000 --no-strong --no-cpu-baseline ) > $O/bench_mg_rows_${NP}ranks_1gpu.json 2> $O/bench_mg_rows_${NP}ranks_1gpu.err
tail -3 $O/bench_mg_rows_${NP}ranks_1gpu.err
python - <<P
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_mg_rows_${NP}ranks_1gpu.json') if l.startswith('{')][-1])
print($NP, 'ranks', d['value'], d['ms_per_step'], d['scaling'], d['config']['grid'], d.get('balance'), d.get('substeps'), d.get('pcg_iterations'), d.get('comm_calls_rank0'))
P
done
timeout 600 python bench.py --steps 2 --warmup 1 --size 4096 --workload dam_break --precond ic0_tile_mg --tol 1e-6 --max-iterations 4000 --no-strong --no-cpu-baseline --no-secondary --no-pmc > $O/bench_mg_4096_1gpu.json 2>/dev/null
python - <<P
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_mg_4096_1gpu.json') if l.startswith('{')][-1])
print('1 gpu', d['value'], d['ms_per_step'], d.get('substeps'), d.get('pcg_iterations'))
P
