"""Worker for tests/test_slab_rows.py: run with torch.distributed.run, N processes (gloo), all sharing cuda:0.
Every rank holds ONE row slab of the job (euler_config.slab_nranks: rows of its bands only, markers inside them with their
global keys) and, as the thing to match, a plain single-GPU run of the same scenario in the same process.  Rank 0 prints one
JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import euler_amd as ea
from euler_amd import scenarios
from euler_amd.slab import SLAB_LOCAL, RcclComm, TorchComm, attach_p2p


def load(sim, workload):
    if workload == "half_tank":
        sim.load_half_tank()
    else:
        sim.load_text(getattr(scenarios, workload)(), upscale=True)
    return sim


def main():
    X, Y, workload, frames, precond = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    p2p = "p2p" in sys.argv[6:]
    rccl = "rccl" in sys.argv[6:]       # one rank per GPU, the library's own RCCL communicator (1 rank on a 1-GPU box)
    if rccl:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        local = 0
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    ref = load(ea.Simulation(X, Y, device=local, dot_mode=ea.DOT_TREE, precond=precond), workload)       # single GPU, the whole grid
    sim = ea.Simulation(X, Y, device=local, dot_mode=ea.DOT_TREE, precond=precond, slab=(rank, world))   # one slab
    comm = RcclComm(sim, SLAB_LOCAL) if rccl else TorchComm(sim, SLAB_LOCAL)
    out = {"world": world, "frames": []}
    if p2p:
        out["p2p_ok"] = attach_p2p(sim)
    load(sim, workload)
    lo, hi = sim.slab_rows()
    out["rows"] = [lo, hi]
    free0, total = torch.cuda.mem_get_info()
    for f in range(frames):
        ref.step()
        sim.step()
        if comm.error:
            raise RuntimeError(comm.error)
        sr, ss = ref.stats(), sim.stats()
        d = {}
        for name, fld in (("u", ea.F_U), ("v", ea.F_V), ("p", ea.F_PRESSURE)):
            a, b = sim.get(fld), ref.get(fld)[lo:hi]
            d["d" + name] = float(np.abs(a - b).max())
        d["pmax"] = float(np.abs(ref.get(ea.F_PRESSURE)).max())
        d["count_differ"] = int((sim.get(ea.F_COUNT) != ref.get(ea.F_COUNT)[lo:hi]).sum())
        d["prev_count_differ"] = int((sim.get(ea.F_PREV_COUNT) != ref.get(ea.F_PREV_COUNT)[lo:hi]).sum())
        # markers: the local ones sit, bit for bit, where their keys say in the single-GPU array, inside the own rows
        m, k, rm = sim.get(ea.F_MARKERS), sim.get(ea.F_MARKER_KEYS), ref.get(ea.F_MARKERS)
        ok = len(m) == len(k) and (len(k) == 0 or int(k.max()) < len(rm))
        d["markers_at_keys"] = bool(ok and np.array_equal(m.view(np.uint32), rm[k].view(np.uint32)))
        d["markers_in_rows"] = bool(len(m) == 0 or (np.floor(m[:, 1]).min() >= lo and np.floor(m[:, 1]).max() < hi))
        # ... and all ranks together hold every key exactly once
        keys_all = [None] * world
        dist.all_gather_object(keys_all, k.tolist())
        flat = np.sort(np.concatenate([np.asarray(x, np.int64) for x in keys_all])) if sum(len(x) for x in keys_all) else np.zeros(0, np.int64)
        d["keys_are_a_permutation"] = bool(len(flat) == len(rm) and np.array_equal(flat, np.arange(len(rm))))
        d["keys_cover_own_count"] = bool(len(flat) == int(ss.n_markers) and np.array_equal(flat, np.arange(len(flat))))
        d["n_markers"] = [int(sr.n_markers), int(ss.n_markers), len(m)]
        d["iters"] = [sr.last_pcg_iterations, ss.last_pcg_iterations]
        d["substeps"] = [sr.last_substeps, ss.last_substeps]
        d["rng"] = [int(sr.rng_state) == int(ss.rng_state), sr.source_exhausted == ss.source_exhausted]
        d["dt_events"] = [int(sr.marker_dt_events), int(ss.marker_dt_events)]
        # worst over the ranks
        agg = [None] * world
        dist.all_gather_object(agg, d)
        w = dict(agg[0])
        for o in agg[1:]:
            for key in ("du", "dv", "dp", "count_differ", "prev_count_differ"):
                w[key] = max(w[key], o[key])
            for key in ("markers_at_keys", "markers_in_rows", "keys_are_a_permutation", "keys_cover_own_count"):
                w[key] = w[key] and o[key]
            w["rng"] = [w["rng"][0] and o["rng"][0], w["rng"][1] and o["rng"][1]]
            w.setdefault("local_markers", [agg[0]["n_markers"][2]]).append(o["n_markers"][2])
        out["frames"].append(w)
    free1, _ = torch.cuda.mem_get_info()
    out["calls"] = comm.counts
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
