#!/usr/bin/env python3
"""First contact with a node, step 1: the latency L of one exchange point of the distributed PCG (DESIGN 7a's only unknown), over the library's own RCCL communicator,
one rank per GPU: launched by tools/node_first_contact.sh through torch.distributed.run.  Rank 0 prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import euler_amd as ea
from euler_amd.slab import SLAB_LOCAL, RcclComm


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    rows = max(64 * world, 512)      # a thin grid: the probe needs a communicator, not a workload
    sim = ea.Simulation(N, rows, device=local, dot_mode=ea.DOT_TREE, precond=ea.PRECOND_IC0_TILE, slab=(rank, world))
    comm = RcclComm(sim, SLAB_LOCAL)
    out = {"world": world, "rccl_version": comm.version, "row_doubles": N, "us_per_exchange": {}}
    for name, rd, ns in (("scalars_only (G2: alpha partials)", 0, 1), ("pair (max |r|, dot)", 0, 2), ("edge rows of z + pair (G1)", N, 2),
                         ("G1 + level-0 rows of the multilevel mode (cells / 256 / world doubles)", N, 2 + N * N // 256 // world)):
        out["us_per_exchange"][name] = round(sim.exchange_latency(200, rd, ns), 2)
    l1 = out["us_per_exchange"]["edge rows of z + pair (G1)"] + out["us_per_exchange"]["scalars_only (G2: alpha partials)"]
    # a slab's iteration at 16384^2 / world costs t_1 / world of kernel time + the two exchanges
    t1 = 2000.0
    out["model_16384"] = {"t1_us_one_gpu_iteration": t1, "exchanges_us_per_iteration": round(l1, 2),
                          "strong_scaling_speedup_estimate": round(t1 / (t1 / world + l1), 2), "note": "kernel time / world + G1 + G2; no overlap assumed"}
    if rank == 0:
        print(json.dumps(out))
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
