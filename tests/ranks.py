"""Test infrastructure: start N rank processes of a worker script directly (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, what the workers'
torch.distributed.init_process_group("gloo") reads) instead of through `python -m torch.distributed.run`: the launcher imports torch, starts an elastic agent and a
store before the first rank exists - about two seconds per test case, a third of the row-slab tests' time.  (bench.py's N > 1 contract test keeps the driver's own
launcher command.)  Rank 0's stdout is what the tests read; a rank that fails takes the others with it (they would wait in a collective for ever)."""
import os
import subprocess
import sys
import tempfile
import time


def launch(nproc, script, args, port, env=None, timeout=900):
    """-> (returncode, stdout of rank 0, stderr tails of all ranks)"""
    base = dict(os.environ if env is None else env, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                HSA_ENABLE_IPC_MODE_LEGACY="0")
    base.setdefault("OMP_NUM_THREADS", "1")      # (what the launcher sets for nproc > 1)
    procs, files = [], []
    for r in range(nproc):
        out = tempfile.TemporaryFile(mode="w+")
        err = tempfile.TemporaryFile(mode="w+")
        files.append((out, err))
        procs.append(subprocess.Popen([sys.executable, script] + [str(a) for a in args], stdout=out, stderr=err,
                                      env=dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", ROLE_RANK=str(r))))
    t0 = time.time()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = next((c for c in codes if c), 0)
            break
        bad = next((c for c in codes if c not in (None, 0)), None)
        if bad is not None or time.time() - t0 > timeout:
            time.sleep(1.0 if bad is not None else 0.0)      # (let the others report what they saw)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            rc = bad if bad is not None else -9
            break
        time.sleep(0.02)
    texts = []
    for out, err in files:
        out.seek(0); err.seek(0)
        texts.append((out.read(), err.read()))
        out.close(); err.close()
    return rc, texts[0][0], "\n".join("[rank %d] %s" % (r, t[1][-1500:]) for r, t in enumerate(texts))
