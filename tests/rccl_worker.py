"""Worker for the RCCL transport tests: run with torch.distributed.run, backend "nccl" (= RCCL).
A 1-GPU box can only offer world size 1 (RCCL refuses two ranks on one device), which still covers:
dlopen + ncclCommInitRank from the C library, the in-place collectives on the handle's own buffers
and stream, and the whole communicator code path of the PCG driver.  With more GPUs the same
script checks the distributed solve against a plain single-GPU run held by every rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import euler_amd as ea
from euler_amd import scenarios
from euler_amd.slab import RcclComm, TorchComm, attach_p2p, p2p_counts


def build(X, Y, workload, device):
    sim = ea.Simulation(X, Y, device=device, dot_mode=ea.DOT_TREE)
    if workload == "half_tank":
        sim.load_half_tank()
    else:
        sim.load_text(getattr(scenarios, workload)(), upscale=True)
    return sim


def torch_transport_selfcheck(rank, world):
    """TorchComm's operations on raw device pointers (zero-copy views), in place, over RCCL."""
    tc = object.__new__(TorchComm)
    tc.torch, tc.dist, tc.rank, tc.world, tc.stage, tc.error, tc._cache = torch, dist, rank, world, False, None, {}
    tc.counts = {"allreduce": 0, "halo": 0, "chain": 0, "allgather": 0}
    tc.bytes = {"rows": 0, "gathered": 0, "allreduce": 0, "halo_rows": 0, "allgather": 0}
    x = torch.full((4,), float(rank + 1), dtype=torch.float64, device="cuda")
    rc = tc._allreduce(None, x.data_ptr(), 4, 0)
    want_sum = world * (world + 1) / 2
    ok = rc == 0 and bool((x == want_sum).all())
    y = torch.full((2,), float(rank), dtype=torch.float64, device="cuda")
    rc = tc._allreduce(None, y.data_ptr(), 2, 1)
    ok = ok and rc == 0 and bool((y == world - 1).all())
    n = 1024
    g = torch.zeros(world * n, dtype=torch.uint8, device="cuda")
    g[rank * n:(rank + 1) * n] = rank + 1
    off = (ea.C.c_int64 * world)(*[r * n for r in range(world)])
    cnt = (ea.C.c_int64 * world)(*[n] * world)
    rc = tc._allgather(None, g.data_ptr(), off, cnt)
    ok = ok and rc == 0 and all(bool((g[r * n:(r + 1) * n] == r + 1).all()) for r in range(world))
    h = torch.arange(4 * 8, dtype=torch.float64, device="cuda") + 100 * rank
    p = h.data_ptr()
    rc = tc._halo(None, p, p + 64, p + 128, p + 192, 8)
    torch.cuda.synchronize()
    ok = ok and rc == 0
    if rank > 0:
        ok = ok and bool((h[16:24] == torch.arange(8, 16, device="cuda") + 100 * (rank - 1)).all())
    if rank + 1 < world:
        ok = ok and bool((h[24:32] == torch.arange(0, 8, device="cuda") + 100 * (rank + 1)).all())
    return ok, tc.error


def main():
    X, Y, workload, frames, coupling = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {"world": world}
    out["torch_transport_ok"], out["torch_transport_error"] = torch_transport_selfcheck(rank, world)
    ref = build(X, Y, workload, local)
    sim = build(X, Y, workload, local)
    comm = RcclComm(sim, coupling)
    p2p = len(sys.argv) > 6 and sys.argv[6] == "p2p"
    if p2p:   # what bench.py runs on a node: mailboxes for the per-iteration exchanges, RCCL for the bulk transfers
        out["p2p_ok"], out["p2p_error"] = attach_p2p(sim), sim._p2p_error
    out["rccl_version"] = comm.version
    out["bands"] = [comm.band_lo, comm.band_hi, comm.nbands]
    out["frames"] = []
    for f in range(frames):
        ref.step()
        sim.step()
        sr, ss = ref.stats(), sim.stats()
        pr = ref.get(ea.F_PRESSURE)
        out["frames"].append({
            "du": float(np.abs(sim.get(ea.F_U) - ref.get(ea.F_U)).max()), "dv": float(np.abs(sim.get(ea.F_V) - ref.get(ea.F_V)).max()),
            "dp": float(np.abs(sim.get(ea.F_PRESSURE) - pr).max()), "pmax": float(np.abs(pr).max()),
            "cells_differing": int(((sim.get(ea.F_COUNT) > 0) != (ref.get(ea.F_COUNT) > 0)).sum()),
            "markers_equal": bool(np.array_equal(sim.get(ea.F_MARKERS).view(np.uint32), ref.get(ea.F_MARKERS).view(np.uint32))),
            "marker_diff": float(np.abs(sim.get(ea.F_MARKERS) - ref.get(ea.F_MARKERS)).max()) if sim.get(ea.F_MARKERS).shape == ref.get(ea.F_MARKERS).shape else 1e30,
            "iters": [sr.last_pcg_iterations, ss.last_pcg_iterations], "substeps": [sr.last_substeps, ss.last_substeps],
            "residual": [sr.last_residual, ss.last_residual]})
    out["calls"] = comm.counts
    if p2p:
        out["p2p_calls"] = p2p_counts(sim)
    h = torch.tensor([float(np.abs(sim.get(ea.F_U)).sum()), float(sim.stats().n_markers)], dtype=torch.float64, device="cuda")
    lo, hi = h.clone(), h.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out["ranks_agree"] = bool(torch.equal(lo, hi))
    sim.close()
    ref.close()
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
