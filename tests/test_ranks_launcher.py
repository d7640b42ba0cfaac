"""tests/ranks.py (the rank launcher of the multi-process tests): two gloo ranks on the CPU - rank 0's stdout comes back, the ranks see the environment
torch.distributed.run would give them, and a rank that fails takes the others down instead of leaving them in a collective."""
import os
import sys
import time

import ranks

WORKER = '''
import os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
r, n = dist.get_rank(), dist.get_world_size()
assert r == int(os.environ["RANK"]) == int(os.environ["LOCAL_RANK"]) and n == int(os.environ["WORLD_SIZE"]) == int(os.environ["LOCAL_WORLD_SIZE"])
t = torch.tensor([float(r + 1)])
dist.all_reduce(t)
if r == 0:
    print('{"sum": %g, "arg": "%s", "omp": "%s"}' % (t.item(), sys.argv[1], os.environ.get("OMP_NUM_THREADS")))
if len(sys.argv) > 2 and r == 1:
    raise SystemExit("rank 1 gives up: " + sys.argv[2])
dist.barrier()
if len(sys.argv) > 2:
    dist.barrier(); dist.barrier()      # (rank 0 would wait here for ever)
'''


def test_rank_launcher(tmp_path):
    script = os.path.join(str(tmp_path), "w.py")
    open(script, "w").write(WORKER)
    env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
    rc, out, err = ranks.launch(3, script, ["x"], 29631, env=env, timeout=120)
    assert rc == 0, err
    assert '"sum": 6' in out and '"arg": "x"' in out and '"omp": "1"' in out, out
    t0 = time.time()
    rc, out, err = ranks.launch(2, script, ["x", "do not tile"], 29632, env=env, timeout=120)
    assert rc != 0 and "do not tile" in err, (rc, err)
    assert time.time() - t0 < 60      # the surviving rank was ended, not waited for
