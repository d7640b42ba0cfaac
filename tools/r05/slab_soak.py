import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ranks
import euler_amd as ea
rc, out, err = ranks.launch(3, os.path.join(ROOT, "tests", "slab_rows_worker.py"), [1000, 1100, "waterfall", 400, ea.PRECOND_IC0_TILE_MG, "split=2", "bands=0-5,5-11,11-18", "maxit=4000"], 29777, timeout=1500)
if rc: print(err[-2000:]); sys.exit(1)
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
fr = d["frames"]
print("split_active", d["split_active"], "frames", len(fr))
for i in (0, 50, 100, 200, 300, 399):
    f = fr[i]
    print(i, {k: f[k] for k in ("iters", "substeps", "dp", "pmax", "du", "dv", "count_differ", "residual")})
print("max iters ratio", max(f["iters"][1] / max(f["iters"][0], 1) for f in fr if f["iters"][0] > 0))
print("worst du", max(f["du"] for f in fr), "worst count_differ", max(f["count_differ"] for f in fr))
