#!/bin/bash
cd "$(dirname "$0")/../../.."
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "3 512 1024 waterfall 80" "4 768 768 dam_break 120"; do
set -- $spec
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 --master-port 2966$1 tests/slab_rows_worker.py $2 $3 $4 $5 4 maxit=4000 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
fr=d['frames']
print('$spec', 'ranks', d['world'], 'frames', len(fr), 'max du %.2e dv %.2e dp/pmax %.2e' % (max(f['du'] for f in fr), max(f['dv'] for f in fr), max(f['dp']/max(f['pmax'],1e-30) for f in fr)),
      'count_differ max', max(f['count_differ'] for f in fr), 'iters 1gpu/slabs', sum(f['iters'][0] for f in fr), sum(f['iters'][1] for f in fr),
      'all markers at keys', all(f['markers_at_keys'] for f in fr), 'n_markers', fr[-1]['n_markers'][:2], 'max residual', max(max(f['residual']) for f in fr))
"
done
