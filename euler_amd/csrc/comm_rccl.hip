// comm_rccl.hip — the library's own communicator for the row-slab pressure solve: the four exchange
// operations of euler_comm_ops (include/euler.h) issued straight to RCCL on the handle's HIP stream.
//
// No host round trip and no interpreter between two kernels of a PCG iteration: an exchange is one
// RCCL enqueue (~ the cost of a kernel launch) ordered by the stream.  On an MI355X node RCCL moves
// these messages over xGMI; all of them are latency-bound (8 B scalars, one grid row of 8 X bytes).
//
// RCCL is bound at run time (dlopen): a single-GPU program never loads it, and inside a process
// that already carries an RCCL (PyTorch bundles one under the same soname) that copy is reused, so
// that there is exactly one RCCL and one HIP runtime per process.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: every call goes through the table below

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "euler_dev.h"

namespace {

struct RcclApi {
  void* dl;
  decltype(&ncclGetVersion) GetVersion;
  decltype(&ncclGetUniqueId) GetUniqueId;
  decltype(&ncclCommInitRank) CommInitRank;
  decltype(&ncclCommDestroy) CommDestroy;
  decltype(&ncclGetErrorString) GetErrorString;
  decltype(&ncclAllReduce) AllReduce;
  decltype(&ncclBroadcast) Broadcast;
  decltype(&ncclAllGather) AllGather;
  decltype(&ncclSend) Send;
  decltype(&ncclRecv) Recv;
  decltype(&ncclGroupStart) GroupStart;
  decltype(&ncclGroupEnd) GroupEnd;
};
RcclApi g_api;
int g_api_state = 0;   // 0 = not tried, 1 = bound, -1 = unavailable

template <typename F>
bool bind(void* dl, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(dl, name));
  if (!*out) eu_set_error("RCCL: symbol %s not found (%s)", name, dlerror());
  return *out != nullptr;
}

int load_rccl() {
  if (g_api_state) return g_api_state > 0 ? EULER_OK : EULER_ECOMM;
  g_api_state = -1;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* dl = nullptr;
  for (const char* n : names)                       // a copy the process already carries wins
    if ((dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!dl)
    for (const char* n : names)
      if ((dl = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!dl) { eu_set_error("RCCL: librccl.so.1 cannot be loaded (%s)", dlerror()); return EULER_ECOMM; }
  g_api.dl = dl;
  if (!bind(dl, "ncclGetVersion", &g_api.GetVersion) || !bind(dl, "ncclGetUniqueId", &g_api.GetUniqueId) ||
      !bind(dl, "ncclCommInitRank", &g_api.CommInitRank) || !bind(dl, "ncclCommDestroy", &g_api.CommDestroy) ||
      !bind(dl, "ncclGetErrorString", &g_api.GetErrorString) || !bind(dl, "ncclAllReduce", &g_api.AllReduce) ||
      !bind(dl, "ncclBroadcast", &g_api.Broadcast) || !bind(dl, "ncclAllGather", &g_api.AllGather) ||
      !bind(dl, "ncclSend", &g_api.Send) || !bind(dl, "ncclRecv", &g_api.Recv) ||
      !bind(dl, "ncclGroupStart", &g_api.GroupStart) || !bind(dl, "ncclGroupEnd", &g_api.GroupEnd))
    return EULER_ECOMM;
  g_api_state = 1;
  return EULER_OK;
}

struct RcclComm {
  ncclComm_t comm;
  euler_sim* S;
  int rank, n;
  uint64_t calls[5];   // allreduce, halo, chain, allgather, exchange
  int small_by_allgather;   // the all-to-all part of an exchange as ncclAllGather instead of sends / receives: 0 = when the slot exceeds 64 doubles (default), 1 = always (EULER_RCCL_SMALL=allgather), -1 = never
};

#define NCHK(call)                                                                              \
  do {                                                                                          \
    ncclResult_t _r = (call);                                                                   \
    if (_r != ncclSuccess) { eu_set_error("RCCL: %s -> %s", #call, g_api.GetErrorString(_r)); return -1; } \
  } while (0)
// inside ncclGroupStart ... ncclGroupEnd: remember the first failure and keep going, so that the group is always closed
// (an open group would swallow every later call of this process)
#define GCHK(call)                                                                              \
  do {                                                                                          \
    ncclResult_t _r = (call);                                                                   \
    if (_r != ncclSuccess && !group_failed) { eu_set_error("RCCL: %s -> %s", #call, g_api.GetErrorString(_r)); group_failed = true; } \
  } while (0)

int op_allreduce(void* ctx, void* dev, int32_t count, int32_t is_max) {
  RcclComm* c = static_cast<RcclComm*>(ctx);
  c->calls[0]++;
  NCHK(g_api.AllReduce(dev, dev, (size_t)count, ncclDouble, is_max ? ncclMax : ncclSum, c->comm, c->S->stream));
  return 0;
}

// both neighbours in one group: the four transfers progress together, no ordering between ranks
int op_halo(void* ctx, void* send_lo, void* send_hi, void* recv_lo, void* recv_hi, int32_t count) {
  RcclComm* c = static_cast<RcclComm*>(ctx);
  c->calls[1]++;
  if (c->n < 2) return 0;
  hipStream_t st = c->S->stream;
  bool group_failed = false;
  NCHK(g_api.GroupStart());
  if (c->rank > 0) {
    GCHK(g_api.Send(send_lo, (size_t)count, ncclDouble, c->rank - 1, c->comm, st));
    GCHK(g_api.Recv(recv_lo, (size_t)count, ncclDouble, c->rank - 1, c->comm, st));
  }
  if (c->rank + 1 < c->n) {
    GCHK(g_api.Send(send_hi, (size_t)count, ncclDouble, c->rank + 1, c->comm, st));
    GCHK(g_api.Recv(recv_hi, (size_t)count, ncclDouble, c->rank + 1, c->comm, st));
  }
  GCHK(g_api.GroupEnd());
  return group_failed ? -1 : 0;
}

int op_chain(void* ctx, void* dev, int64_t nbytes, int32_t src, int32_t dst) {
  RcclComm* c = static_cast<RcclComm*>(ctx);
  c->calls[2]++;
  if (c->rank == src) NCHK(g_api.Send(dev, (size_t)nbytes, ncclUint8, dst, c->comm, c->S->stream));
  else if (c->rank == dst) NCHK(g_api.Recv(dev, (size_t)nbytes, ncclUint8, src, c->comm, c->S->stream));
  return 0;
}

// in place: rank r's chunk already sits at base + off[r].  Equal chunks -> one all-gather,
// otherwise (bands do not divide evenly) one broadcast per rank inside a group.
int op_allgather(void* ctx, void* base, const int64_t* off, const int64_t* cnt) {
  RcclComm* c = static_cast<RcclComm*>(ctx);
  c->calls[3]++;
  char* b = static_cast<char*>(base);
  hipStream_t st = c->S->stream;
  bool even = cnt[0] > 0;
  for (int r = 0; r < c->n; ++r) even = even && cnt[r] == cnt[0] && off[r] == off[0] + (int64_t)r * cnt[0];
  if (even) {
    NCHK(g_api.AllGather(b + off[c->rank], b + off[0], (size_t)cnt[0], ncclUint8, c->comm, st));
  } else {
    bool group_failed = false;
    NCHK(g_api.GroupStart());
    for (int r = 0; r < c->n; ++r)
      if (cnt[r] > 0) GCHK(g_api.Broadcast(b + off[r], b + off[r], (size_t)cnt[r], ncclUint8, r, c->comm, st));      // (a rank may have nothing to contribute)
    GCHK(g_api.GroupEnd());
    if (group_failed) return -1;
  }
  return 0;
}

// ONE group for a PCG iteration's exchange point (include/euler.h euler_comm_ops.exchange): the edge rows to / from the two
// neighbours and `nsmall` doubles of every rank to every rank, all as sends and receives that progress together - the
// per-iteration traffic of the distributed PCG is latency-bound (8 B ... 131 KB), so what counts is the number of
// serial RCCL operations, two per iteration this way.
int op_exchange(void* ctx, void* send_lo, void* send_hi, void* recv_lo, void* recv_hi, int32_t count, void* small, int32_t nsmall) {
  RcclComm* c = static_cast<RcclComm*>(ctx);
  c->calls[4]++;
  if (c->n < 2) return 0;
  hipStream_t st = c->S->stream;
  double* sm = static_cast<double*>(small);
  bool group_failed = false;
  NCHK(g_api.GroupStart());
  if (count > 0) {
    if (c->rank > 0) {
      GCHK(g_api.Send(send_lo, (size_t)count, ncclDouble, c->rank - 1, c->comm, st));
      GCHK(g_api.Recv(recv_lo, (size_t)count, ncclDouble, c->rank - 1, c->comm, st));
    }
    if (c->rank + 1 < c->n) {
      GCHK(g_api.Send(send_hi, (size_t)count, ncclDouble, c->rank + 1, c->comm, st));
      GCHK(g_api.Recv(recv_hi, (size_t)count, ncclDouble, c->rank + 1, c->comm, st));
    }
  }
  // the all-to-all part: a few doubles per rank ride in the same group as sends / receives (latency-bound: one serial RCCL operation); a large slot
  // (the multilevel mode's level-0 rows: cells / 256 doubles over all ranks, 8 MB at 16384^2) goes as ONE in-place ncclAllGather behind the group
  // instead of N - 1 send / receive pairs per rank - it is bandwidth-bound, and the ring spreads it over the links
  const bool by_allgather = nsmall > 0 && (c->small_by_allgather > 0 || (c->small_by_allgather == 0 && nsmall > 64));
  if (nsmall > 0 && !by_allgather) {
    for (int r = 0; r < c->n; ++r) {
      if (r == c->rank) continue;
      GCHK(g_api.Send(sm + (size_t)c->rank * nsmall, (size_t)nsmall, ncclDouble, r, c->comm, st));
      GCHK(g_api.Recv(sm + (size_t)r * nsmall, (size_t)nsmall, ncclDouble, r, c->comm, st));
    }
  }
  GCHK(g_api.GroupEnd());
  if (by_allgather && !group_failed) NCHK(g_api.AllGather(sm + (size_t)c->rank * nsmall, sm, (size_t)nsmall, ncclDouble, c->comm, st));
  return group_failed ? -1 : 0;
}

}  // namespace

extern "C" int euler_rccl_unique_id(void* id_out, int32_t cap) {
  if (!id_out || cap < (int32_t)sizeof(ncclUniqueId)) { eu_set_error("euler_rccl_unique_id: need %d bytes", (int)sizeof(ncclUniqueId)); return EULER_EINVAL; }
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t r = g_api.GetUniqueId(&id);
  if (r != ncclSuccess) { eu_set_error("RCCL: ncclGetUniqueId -> %s", g_api.GetErrorString(r)); return EULER_ECOMM; }
  memcpy(id_out, &id, sizeof id);
  return EULER_OK;
}

extern "C" int euler_rccl_version(void) {
  if (load_rccl()) return -1;
  int v = 0;
  return g_api.GetVersion(&v) == ncclSuccess ? v : -1;
}

void eu_rccl_release(euler_sim* S) {
  RcclComm* c = static_cast<RcclComm*>(S->rccl);
  if (!c) return;
  if (c->comm) (void)g_api.CommDestroy(c->comm);
  free(c);
  S->rccl = nullptr;
}

extern "C" int euler_set_comm_rccl(euler_sim* S, const void* unique_id, int32_t id_bytes, int32_t rank, int32_t nranks,
                                   int32_t coupling) {
  if (!S || !unique_id || id_bytes != (int32_t)sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks) {
    eu_set_error("euler_set_comm_rccl: bad argument (id of %d bytes expected)", (int)sizeof(ncclUniqueId));
    return EULER_EINVAL;
  }
  int rc = load_rccl();
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(S->stream));
  eu_p2p_release(S);
  eu_rccl_release(S);
  RcclComm* c = static_cast<RcclComm*>(calloc(1, sizeof(RcclComm)));
  if (!c) return EULER_ENOMEM;
  c->S = S; c->rank = rank; c->n = nranks;
  c->small_by_allgather = S->opt[EULER_OPT_RCCL_SMALL] == 1 ? 1 : S->opt[EULER_OPT_RCCL_SMALL] == 2 ? -1 : 0;      // EULER_OPT_RCCL_SMALL: by size (default) / always ncclAllGather / never
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  HIPCHK(hipSetDevice(S->cfg.device));
  ncclResult_t r = g_api.CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) { eu_set_error("RCCL: ncclCommInitRank(%d of %d) -> %s", rank, nranks, g_api.GetErrorString(r)); free(c); return EULER_ECOMM; }
  S->rccl = c;
  euler_comm_ops ops;
  ops.ctx = c; ops.rank = rank; ops.nranks = nranks;
  ops.allreduce = op_allreduce; ops.halo = op_halo; ops.chain = op_chain; ops.allgather = op_allgather;
  ops.exchange = S->opt[EULER_OPT_RCCL_NO_EXCHANGE] ? nullptr : op_exchange;   // (experiments: the same traffic as halo + all-gather)
  rc = eu_install_comm(S, &ops, coupling, /*allow_single=*/1);
  if (rc) eu_rccl_release(S);
  return rc;
}

extern "C" int euler_comm_calls(euler_sim* S, uint64_t out[5]) {
  if (!S || !out) return EULER_EINVAL;
  RcclComm* c = static_cast<RcclComm*>(S->rccl);
  for (int k = 0; k < 5; ++k) out[k] = c ? c->calls[k] : 0;
  return EULER_OK;
}
