"""Row slabs, multilevel mode: what ONE rank of N does per PCG iteration with the cycle split by rows against the replicated cycle.

    python tools/r05/split_probe.py N X Y [split]        (starts N ranks on the visible GPU(s); split: EULER_OPT_MG_SPLIT_LEVEL, default 0 = by size, -1 = replicated)

N ranks share one GPU here, so wall time means nothing; what is read off: the bytes a rank receives per G1 exchange (the all-gather's whole buffer) and sends to a neighbour, the
number of exchange points per iteration, iterations per solve, and rank 0's kernel time per iteration by class (HIP events on its stream: the other ranks' kernels interleave,
so the SUM over a class is an upper bound of what the rank would take alone)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker():
    import torch
    import torch.distributed as dist
    import euler_amd as ea
    from euler_amd.slab import SLAB_LOCAL, TorchComm
    X, Y, split, substeps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    sim = ea.Simulation(X, Y, device=0, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE_MG, slab=(rank, world), max_iterations=4000)
    sim.set_option(ea.OPT_MG_SPLIT_LEVEL, split)
    comm = TorchComm(sim, SLAB_LOCAL)
    sim.load_half_tank()
    sim.substep(0.01)      # (the first solve builds the plan and the buffers)
    torch.cuda.synchronize()
    b0, c0 = dict(comm.bytes), dict(comm.counts)
    st0 = sim.stats()
    sim.profile_enable(ea.profile_class_names())
    sim.profile_reset()
    for _ in range(substeps):
        sim.substep(0.01)
    torch.cuda.synchronize()
    st1 = sim.stats()
    its = st1.total_pcg_iterations - st0.total_pcg_iterations
    prof = sim.profile()
    out = {"ranks": world, "grid": [X, Y], "split_level": sim.get_option(ea.OPT_MG_SPLIT_ACTIVE), "iterations": its, "substeps": substeps,
           "exchange_points_per_iteration": round((comm.counts["exchange"] - c0["exchange"] + comm.counts["allreduce"] - c0["allreduce"]) / max(its, 1), 2),
           "bytes_gathered_per_iteration": round((comm.bytes["gathered"] - b0["gathered"]) / max(its, 1)),
           "bytes_to_neighbours_per_iteration": round((comm.bytes["rows"] - b0["rows"]) / max(its, 1)),
           "per_solve_bytes": {k: round((comm.bytes[k] - b0[k]) / substeps) for k in ("allreduce", "halo_rows", "allgather")},
           "rank0_kernel_us_per_iteration": {k: round(v[0] * 1e3 / max(its, 1), 1) for k, v in prof.items() if v[0] > 0}}
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if sys.argv[1] == "worker":
        worker()
    else:
        import ranks
        n, X, Y = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        split = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        substeps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
        rc, out, err = ranks.launch(n, os.path.abspath(__file__), ["worker", X, Y, split, substeps], 29733, timeout=1500)
        print(out.strip() if rc == 0 else err[-3000:])
        sys.exit(rc)
