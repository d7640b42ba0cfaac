#!/usr/bin/env python3
"""Regenerate tests/golden/trajectories.json: the pinned oracle stepped through every trajectory of tests/trajectories.py, per frame the SHA-1 of
each array the GPU tests compare (+ counters).  Run in the build container (a minute of CPU):  python tests/golden/make_trajectories.py [name ...]
The oracle itself is pinned to the compiled reference by tests/test_oracle_vs_ref.py and the golden fixtures; tests/test_trajectories.py re-checks
the cheap trajectories of this file against a live oracle on every CPU run."""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import trajectories as T  # noqa: E402


def record(name):
    spec = T.SPECS[name]
    o = T.make_oracle(spec)
    rec = {"frames": []}
    if spec.get("init"):
        rec["init"] = T.snapshot(o, spec)
    n = spec.get("substeps") or spec["frames"]
    every = spec.get("every", 1)
    for f in range(n):
        t0 = time.perf_counter()
        T.advance(o, spec)
        full = (f + 1) % every == 0 or f == n - 1
        snap = T.snapshot(o, spec if full else dict(spec, fields=()))
        rec["frames"].append(snap)
        if spec.get("big"):
            print("  %s %d: %d substeps, %d iterations, %d markers, %.1f s" % (name, f, snap["last_substeps"], snap["last_pcg_iterations"], snap["n_markers"], time.perf_counter() - t0), flush=True)
    if spec.get("render"):
        rec["render"] = hashlib.sha1(o.render(spec["X"], spec["Y"])).hexdigest()[:20]
    o.close()
    return rec


def stamp():
    """MANIFEST.json <- the SHA-1 of oracle/euler_oracle.c the records stand for (tests/test_trajectories.py checks it on every CPU run): call this after the
    big records have been replayed against the current source (EULER_REPLAY_BIG=1) or regenerated"""
    root = os.path.dirname(os.path.dirname(HERE))
    with open(os.path.join(root, "oracle", "euler_oracle.c"), "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    path = os.path.join(HERE, "MANIFEST.json")
    with open(path) as f:
        man = json.load(f)
    man["oracle_source"] = {"file": "oracle/euler_oracle.c", "sha1": sha, "big_records_replayed": sorted(n for n, s in T.SPECS.items() if s.get("big")),
                            "how": "EULER_REPLAY_BIG=1 python -m pytest tests/test_trajectories.py -k big, then make_trajectories.py --stamp"}
    with open(path, "w") as f:
        json.dump(man, f, indent=1)
        f.write("\n")
    print("stamped", sha)


def main():
    if sys.argv[1:] == ["--stamp"]:
        return stamp()
    names = sys.argv[1:] or sorted(T.SPECS)
    try:
        with open(T.PATH) as f:
            out = json.load(f)
    except OSError:
        out = {}
    for n in names:
        t0 = time.perf_counter()
        out[n] = record(n)
        print("%-34s %3d frames  %.1f s" % (n, len(out[n]["frames"]), time.perf_counter() - t0), flush=True)
        if os.environ.get("EULER_TRAJ_OUT"):      # (several big recordings side by side: each writes its own file, merged afterwards)
            with open(os.environ["EULER_TRAJ_OUT"] + n + ".json", "w") as f:
                json.dump({n: out[n]}, f, indent=0, sort_keys=True)
    if os.environ.get("EULER_TRAJ_OUT"):
        return
    with open(T.PATH, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
