"""The N>1 path of bench.py on CPU: two processes, gloo backend.  Checks the timing protocol
(barrier brackets, MAX over ranks) and the whole-job aggregation of euler_amd.dist."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from euler_amd.dist import Group, whole_job_rate
    g = Group(backend="gloo")
    assert g.world == 2
    delay = 0.05 if g.rank == 0 else 0.15          # rank 1 is the slow one
    t = g.timed(lambda: time.sleep(delay), steps=4)
    rate = whole_job_rate(1000.0, 4, t, g)
    mx = g.reduce_max(g.rank)
    if g.rank == 0:
        print(json.dumps({"t": t, "rate": rate, "max_rank": mx}))
    g.close()
""") % ROOT


def test_two_rank_gloo_timing_and_aggregation(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert 0.6 <= d["t"] < 2.0            # 4 x 0.15 s: the slow rank sets the time
    assert abs(d["rate"] - 2 * 4 * 1000.0 / d["t"]) < 1e-6 * d["rate"]
    assert d["max_rank"] == 1.0
