"""Process-group plumbing for bench.py: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" on CPU for tests).  Nothing here touches simulation data: in round 1 the N>1
path runs one independent replica per rank; this module only provides the timing
protocol the bench contract prescribes (barrier + device sync on both sides, MAX over ranks) and
the whole-job aggregation."""
import os
import time


class Group:
    def __init__(self, backend=None, force=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        if self.world > 1 or force:   # force: a one-rank process group (transport self-test on a 1-GPU box)
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if force and self.world == 1:
                for k, v in (("MASTER_PORT", "29577"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                    os.environ.setdefault(k, v)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend is None:
                backend = os.environ.get("EULER_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
            if backend == "nccl" and torch.cuda.device_count() < local_world and not os.environ.get("EULER_DIST_BACKEND"):
                # more ranks than devices on this node (a 1-GPU box asked for --gpus 4): the ranks share the devices over gloo - a functional run, not a scaling measurement
                import sys
                if self.rank == 0:
                    print("bench: %d local ranks on %d device(s): the ranks share them (gloo); not a scaling measurement" % (local_world, torch.cuda.device_count()), file=sys.stderr)
                backend = "gloo"
                os.environ["EULER_SHARE_GPU"] = "1"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                if os.environ.get("EULER_SHARE_GPU"):       # tests: several gloo ranks on one GPU
                    self.local_rank = 0
                dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def sync_device(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def reduce_max(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def reduce_sum(self, x):
        if not self.dist:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, fn, steps):
        """Run fn() `steps` times between barrier+sync brackets; returns the MAX wall time over ranks."""
        self.sync_device()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.sync_device()
        self.barrier()
        return self.reduce_max(time.perf_counter() - t0)

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
            self.dist = None


def whole_job_rate(units_per_rank_per_step, steps, elapsed_max, group):
    """Aggregate throughput of the whole job: units all ranks processed / the slowest rank's time."""
    total_units = group.reduce_sum(units_per_rank_per_step * steps)
    return total_units / elapsed_max
