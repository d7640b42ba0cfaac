"""Row-slab distribution of the pressure solve over torch.distributed (DESIGN.md "Multi-GPU").

The C library drives the PCG loop and calls four exchange operations (include/euler.h
`euler_comm_ops`); this module implements them: backend "nccl" (= RCCL over xGMI on an MI355X
node) works on the device buffers in place; backend "gloo" (tests, incl. several ranks sharing
one GPU) stages through host memory.  Device pointers of the handle are wrapped zero-copy as torch
tensors through `__cuda_array_interface__`; the library is switched onto torch's current stream
so that collectives and kernels are ordered by the stream.
"""
import ctypes as C
import traceback

SLAB_LOCAL, SLAB_EXACT = 0, 1

_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32)
_HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32)
_CHAIN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)
_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32)


class CommOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("nranks", C.c_int32),
                ("allreduce", _ALLREDUCE), ("halo", _HALO), ("chain", _CHAIN), ("allgather", _ALLGATHER), ("exchange", _EXCHANGE)]


class _DevMem:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def slab_bands(nbands, rank, nranks):
    """Bands [lo, hi) of a rank: the same even split the C library uses (euler_set_comm)."""
    return nbands * rank // nranks, nbands * (rank + 1) // nranks


class TorchComm:
    """Attach a torch.distributed process group to one Simulation handle."""

    def __init__(self, sim, coupling=SLAB_EXACT):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.sim = torch, dist, sim
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.stage = dist.get_backend() != "nccl"      # gloo: go through host memory
        self.error = None
        self._cache = {}
        self.counts = {"allreduce": 0, "halo": 0, "chain": 0, "allgather": 0, "exchange": 0}
        # what this rank sent to neighbours / received of all-gathers (the whole buffer) / all-reduced; halo_* and allgather: the stand-alone operations (per solve: the
        # multilevel mode's operators), rows / gathered: the fused exchange points of the PCG iterations
        self.bytes = {"rows": 0, "gathered": 0, "allreduce": 0, "halo_rows": 0, "allgather": 0}
        # EULER_TORCH_NO_EXCHANGE: leave the fused operation out (the library then issues halo + allgather: the fallback path)
        import os
        fused = _EXCHANGE(self._exchange) if not os.environ.get("EULER_TORCH_NO_EXCHANGE") else _EXCHANGE()
        self._cb = (_ALLREDUCE(self._allreduce), _HALO(self._halo), _CHAIN(self._chain), _ALLGATHER(self._allgather), fused)
        self.ops = CommOps(None, self.rank, self.world, *self._cb)
        L = sim.L
        rc = L.euler_set_stream(sim.h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc == 0:
            rc = L.euler_set_comm(sim.h, C.cast(C.pointer(self.ops), C.c_void_p), coupling)
        if rc:
            raise RuntimeError("euler_set_comm failed: %s" % L.euler_last_error().decode())
        lo, hi, nb = C.c_int32(), C.c_int32(), C.c_int32()
        L.euler_slab_info(sim.h, C.byref(lo), C.byref(hi), C.byref(nb))
        self.band_lo, self.band_hi, self.nbands = lo.value, hi.value, nb.value
        assert getattr(sim, "slab", None) and len(sim.slab) == 4 or (self.band_lo, self.band_hi) == slab_bands(self.nbands, self.rank, self.world)
        sim._comm = self       # keep the callbacks alive as long as the handle

    # -- helpers
    def _t(self, ptr, nbytes, dtype=None):
        key = (ptr, nbytes, dtype)
        t = self._cache.get(key)
        if t is None:                  # the handle's buffers never move: wrap each one once
            t = self.torch.as_tensor(_DevMem(ptr, nbytes), device="cuda")
            if dtype is not None:
                t = t.view(dtype)
            self._cache[key] = t
        return t

    def _guard(self, name, fn):
        try:
            self.counts[name] += 1
            fn()
            return 0
        except Exception:          # never let an exception cross the C boundary
            self.error = traceback.format_exc()
            return -1

    # -- the four operations
    def _allreduce(self, ctx, ptr, count, is_max):
        def run():
            self.bytes["allreduce"] += 8 * count
            t = self._t(ptr, 8 * count, self.torch.float64)
            op = self.dist.ReduceOp.MAX if is_max else self.dist.ReduceOp.SUM
            if self.stage:
                c = t.cpu()
                self.dist.all_reduce(c, op=op)
                t.copy_(c)
            else:
                self.dist.all_reduce(t, op=op)
        return self._guard("allreduce", run)

    def _halo(self, ctx, send_lo, send_hi, recv_lo, recv_hi, count):
        def run():
            f64, n = self.torch.float64, 8 * count
            pairs = []   # (send tensor, recv tensor, peer)
            if self.rank > 0:
                pairs.append((self._t(send_lo, n, f64), self._t(recv_lo, n, f64), self.rank - 1))
            if self.rank + 1 < self.world:
                pairs.append((self._t(send_hi, n, f64), self._t(recv_hi, n, f64), self.rank + 1))
            self.bytes["halo_rows"] += n * len(pairs)
            if self.stage:
                host = [(s.cpu(), self.torch.empty(count, dtype=f64), r, p) for s, r, p in pairs]
                reqs = []
                for s, tmp, r, p in host:
                    reqs.append(self.dist.isend(s, p))
                    reqs.append(self.dist.irecv(tmp, p))
                for q in reqs:
                    q.wait()
                for s, tmp, r, p in host:
                    r.copy_(tmp)
            elif pairs:
                ops = []
                for s, r, p in pairs:
                    ops.append(self.dist.P2POp(self.dist.isend, s, p))
                    ops.append(self.dist.P2POp(self.dist.irecv, r, p))
                for q in self.dist.batch_isend_irecv(ops):
                    q.wait()
        return self._guard("halo", run)

    def _exchange(self, ctx, send_lo, send_hi, recv_lo, recv_hi, count, small, nsmall):
        """euler_comm_ops.exchange: neighbour rows + an all-gather of nsmall doubles per rank (in place), as one operation."""
        def run():
            f64 = self.torch.float64
            pairs = []
            if count > 0:
                n = 8 * count
                if self.rank > 0:
                    pairs.append((self._t(send_lo, n, f64), self._t(recv_lo, n, f64), self.rank - 1))
                if self.rank + 1 < self.world:
                    pairs.append((self._t(send_hi, n, f64), self._t(recv_hi, n, f64), self.rank + 1))
            sm = self._t(small, 8 * nsmall * self.world, f64) if nsmall > 0 else None
            self.bytes["rows"] += 8 * count * len(pairs)
            self.bytes["gathered"] += 8 * nsmall * self.world
            if self.stage:
                host = [(s.cpu(), self.torch.empty(count, dtype=f64), r, p) for s, r, p in pairs]
                reqs = []
                for s, tmp, r, p in host:
                    reqs.append(self.dist.isend(s, p))
                    reqs.append(self.dist.irecv(tmp, p))
                if sm is not None:
                    mine = sm[self.rank * nsmall:(self.rank + 1) * nsmall].cpu()
                    every = [self.torch.empty(nsmall, dtype=f64) for _ in range(self.world)]
                    self.dist.all_gather(every, mine)
                for q in reqs:
                    q.wait()
                for s, tmp, r, p in host:
                    r.copy_(tmp)
                if sm is not None:
                    sm.copy_(self.torch.cat(every))
            else:
                ops = []
                for s, r, p in pairs:
                    ops.append(self.dist.P2POp(self.dist.isend, s, p))
                    ops.append(self.dist.P2POp(self.dist.irecv, r, p))
                reqs = self.dist.batch_isend_irecv(ops) if ops else []
                if sm is not None:
                    self.dist.all_gather_into_tensor(sm, sm[self.rank * nsmall:(self.rank + 1) * nsmall].clone())
                for q in reqs:
                    q.wait()
        return self._guard("exchange", run)

    def _chain(self, ctx, ptr, nbytes, src, dst):
        def run():
            t = self._t(ptr, nbytes)
            if self.rank == src:
                self.dist.send(t.cpu() if self.stage else t, dst)
            elif self.rank == dst:
                if self.stage:
                    tmp = self.torch.empty(nbytes, dtype=self.torch.uint8)
                    self.dist.recv(tmp, src)
                    t.copy_(tmp)
                else:
                    self.dist.recv(t, src)
        return self._guard("chain", run)

    def _allgather(self, ctx, base, off, cnt):
        def run():
            self.bytes["allgather"] += sum(cnt[r] for r in range(self.world))
            for r in range(self.world):
                if cnt[r] == 0:           # (a rank may have nothing to contribute: the same on every rank)
                    continue
                t = self._t(base + off[r], cnt[r])
                if self.stage:
                    c = t.cpu() if r == self.rank else self.torch.empty(cnt[r], dtype=self.torch.uint8)
                    self.dist.broadcast(c, src=r)
                    if r != self.rank:
                        t.copy_(c)
                else:
                    self.dist.broadcast(t, src=r)
        return self._guard("allgather", run)


class RcclUnavailable(RuntimeError):
    pass


class RcclComm:
    """The library's own RCCL communicator (csrc/comm_rccl.hip): no Python between the kernels of a
    PCG iteration.  torch.distributed is used once, to hand rank 0's ncclUniqueId to every rank.
    Construction is collective and fails collectively: if RCCL cannot be bound on rank 0, or
    ncclCommInitRank fails on any rank, every rank raises RcclUnavailable (and the handle is left without a
    communicator), so that the caller can fall back on all ranks alike."""

    ID_BYTES = 128

    def __init__(self, sim, coupling=SLAB_EXACT):
        import torch
        import torch.distributed as dist
        self.sim = sim
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.error = None
        L = sim.L
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        buf = (C.c_ubyte * self.ID_BYTES)()
        status, why = 0, ""
        if self.rank == 0:
            status = L.euler_rccl_unique_id(buf, self.ID_BYTES)
            why = L.euler_last_error().decode() if status else ""
        t = torch.tensor(list(bytes(buf)) + [1 if status else 0], dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        t = t.cpu()
        if int(t[-1]):
            raise RcclUnavailable("euler_rccl_unique_id failed on rank 0: %s" % (why or "see rank 0"))
        ident = bytes(t[:-1].tolist())
        rc = L.euler_set_comm_rccl(sim.h, ident, len(ident), self.rank, self.world, coupling)
        why = L.euler_last_error().decode() if rc else ""
        ok = torch.tensor([0 if rc else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            L.euler_set_comm(sim.h, None, 0)      # drop whatever was installed
            raise RcclUnavailable("euler_set_comm_rccl failed: %s" % (why or "on another rank"))
        lo, hi, nb = C.c_int32(), C.c_int32(), C.c_int32()
        L.euler_slab_info(sim.h, C.byref(lo), C.byref(hi), C.byref(nb))
        self.band_lo, self.band_hi, self.nbands = lo.value, hi.value, nb.value
        assert getattr(sim, "slab", None) and len(sim.slab) == 4 or (self.band_lo, self.band_hi) == slab_bands(self.nbands, self.rank, self.world)
        self.version = L.euler_rccl_version()
        sim._comm = self

    @property
    def counts(self):
        out = (C.c_uint64 * 5)()
        self.sim.L.euler_comm_calls(self.sim.h, out)
        return dict(zip(("allreduce", "halo", "chain", "allgather", "exchange"), (int(v) for v in out)))


P2P_HANDLE_BYTES = 256     # EULER_P2P_HANDLE_BYTES: mailbox + the z / s / s2 arrays


def attach_p2p(sim):
    """Route the per-iteration exchanges (three scalar all-reduces, the ghost rows) over peer-to-peer
    mailboxes (csrc/comm_p2p.hip) on top of the communicator already installed on `sim`.
    torch.distributed only carries the IPC handles (256 bytes per rank), once.  Returns True when the mailboxes are in
    use on EVERY rank; on any failure all ranks stay on the installed communicator (the reason is in
    sim._p2p_error)."""
    import torch
    import torch.distributed as dist
    L = sim.L
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = (C.c_ubyte * P2P_HANDLE_BYTES)()
    rc = L.euler_p2p_export(sim.h, buf, P2P_HANDLE_BYTES)
    sim._p2p_error = None if rc == 0 else L.euler_last_error().decode()
    mine = torch.tensor(list(bytes(buf)) + [0 if rc == 0 else 1], dtype=torch.uint8, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    every = [t.cpu() for t in every]
    if any(int(t[-1]) for t in every):          # some rank could not export: nobody connects
        return False
    handles = b"".join(bytes(t[:-1].tolist()) for t in every)
    rc = L.euler_p2p_connect(sim.h, handles, world)
    if rc:
        sim._p2p_error = L.euler_last_error().decode()
    # connect is collective (its self-test is an all-reduce over the mailboxes): a rank that failed before it
    # leaves the others to time out, so agree on the outcome and fall back together
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0 and rc == 0:
        sim._p2p_error = "another rank could not connect its mailboxes"
        L.euler_p2p_disconnect(sim.h)
    return int(ok.item()) == 1


def p2p_counts(sim):
    out = (C.c_uint64 * 2)()
    sim.L.euler_p2p_calls(sim.h, out)
    return {"allreduce": int(out[0]), "halo": int(out[1])}
