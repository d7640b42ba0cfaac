"""Worker for the slab tests: run with torch.distributed.run, N processes (gloo), all on cuda:0.
Each rank runs the distributed solve and, in the same process, a plain single-GPU run of the same
scenario as the reference; rank 0 prints one JSON line."""
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
from euler_amd.slab import SLAB_EXACT, SLAB_LOCAL, TorchComm, attach_p2p, p2p_counts


def build(X, Y, workload):
    sim = ea.Simulation(X, Y, dot_mode=ea.DOT_TREE)
    if workload == "half_tank":
        sim.load_half_tank()
    elif workload.startswith("stacked_dam_break_"):      # bench.py's weak-scaling workload: one closed tank per row slab
        sim.load_text(scenarios.stacked(scenarios.dam_break(), int(workload.rsplit("_", 1)[1])), upscale=True)
    else:
        sim.load_text(getattr(scenarios, workload)(), upscale=True)
    return sim


def main():
    X, Y, workload, frames, coupling = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    p2p = len(sys.argv) > 6 and sys.argv[6] == "p2p"
    fusion = "fusion" in sys.argv[6:]      # opt-in (euler_set_option EULER_OPT_SLAB_FUSION, on every rank's handle): read the neighbouring slabs' edge rows where they live
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ref = build(X, Y, workload)          # single-GPU run, the thing to match
    sim = build(X, Y, workload)
    comm = TorchComm(sim, coupling)
    out = {"world": world, "bands": [comm.band_lo, comm.band_hi, comm.nbands], "frames": []}
    if fusion:
        sim.set_option(ea.OPT_SLAB_FUSION, 1)
    if p2p:   # scalar all-reduces and ghost rows over IPC mailboxes (here: several processes on one GPU)
        out["p2p_ok"] = attach_p2p(sim)
        out["p2p_error"] = sim._p2p_error
    for f in range(frames):
        ref.step()
        sim.step()
        if comm.error:
            raise RuntimeError(comm.error)
        sr, ss = ref.stats(), sim.stats()
        du = float(np.abs(sim.get(ea.F_U) - ref.get(ea.F_U)).max())
        dv = float(np.abs(sim.get(ea.F_V) - ref.get(ea.F_V)).max())
        dp = float(np.abs(sim.get(ea.F_PRESSURE) - ref.get(ea.F_PRESSURE)).max())
        pmax = float(np.abs(ref.get(ea.F_PRESSURE)).max())
        cells = int(((sim.get(ea.F_COUNT) > 0) != (ref.get(ea.F_COUNT) > 0)).sum())
        ms_, mr_ = sim.get(ea.F_MARKERS), ref.get(ea.F_MARKERS)
        same_markers = bool(np.array_equal(ms_.view(np.uint32), mr_.view(np.uint32)))
        marker_diff = float(np.abs(ms_ - mr_).max()) if ms_.shape == mr_.shape and len(ms_) else (0.0 if ms_.shape == mr_.shape else 1e30)
        out["frames"].append({"du": du, "dv": dv, "dp": dp, "pmax": pmax, "cells_differing": cells, "markers_equal": same_markers, "marker_diff": marker_diff,
                              "iters": [sr.last_pcg_iterations, ss.last_pcg_iterations], "substeps": [sr.last_substeps, ss.last_substeps],
                              "residual": [sr.last_residual, ss.last_residual],
                              "finite": bool(np.isfinite(sim.get(ea.F_U)).all() and np.isfinite(sim.get(ea.F_V)).all())})
    out["calls"] = comm.counts
    if p2p:
        out["p2p_calls"] = p2p_counts(sim)
    # every rank must hold the same replicated state
    h = torch.tensor([float(np.abs(sim.get(ea.F_U)).sum()), float(sim.stats().n_markers)], dtype=torch.float64)
    lo, hi = h.clone(), h.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out["ranks_agree"] = bool(torch.equal(lo, hi))
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
