// comm_p2p.hip — the latency-bound exchanges of the row-slab pressure solve as direct peer-to-peer
// mailbox writes (xGMI on an MI355X node), no collective library in the PCG loop.
//
// Every PCG iteration of the distributed solve needs three 8-byte all-reduces (dot(s,As), |r|_inf,
// dot(z,r)) and one ghost-row exchange of the search vector with the two neighbouring slabs
// (DESIGN.md "Multi-GPU").  All of them are pure latency.  Each rank owns a MAILBOX in fine-grained device
// memory that its peers map through HIP IPC; an exchange is then
//     all-reduce : one 1-block kernel - lane j stores {value, tag} into peer j's slot for this rank, polls its
//                  own slot j, and lane order fixes the order of the sum (bit-identical on every rank);
//     ghost rows : one 1-block kernel - the row is written into the neighbour's mailbox, then the own
//                  mailbox is polled for the neighbour's row.
// Every datum travels as a self-validating 16-byte granule {lo32, tag, hi32, tag} written with one
// system-scope write-through store and read with system-scope loads - the same hand-off the IC(0) band
// pipeline uses inside a GPU (k_pcg.hip; MI355X_MICROARCH "granule" hand-off) - so no fence, no
// write-back of the L2 (which holds the solver's dirty vectors) and no assumption about the order in
// which stores to a peer arrive.  The tag is the exchange's sequence number; two slot parities are enough
// because an exchange k+2 cannot begin before every rank has finished reading exchange k (it needs their
// contribution to exchange k+1, which they issue after k in stream order).  Waits are bounded (sticky
// error -> EULER_ETIMEOUT).  The bulk transfers (band hand-off rows, all-gather of p: once per solve)
// stay on the communicator that was installed first (RCCL).
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "euler_dev.h"

static_assert(sizeof(P2PBoxHeader) <= P2P_HDR_BYTES, "mailbox header");

struct P2PState {
  void* box;                          // own mailbox (fine-grained device memory)
  size_t box_bytes;
  void* peer[P2P_MAXR];               // host copy of the mapped mailboxes (own entry = box)
  void** peer_dev;                    // the same table on the device
  int rank, n, X;
  euler_comm_ops base;                // the communicator underneath (bulk transfers)
  uint64_t calls[2];
  // the neighbouring slabs' z and search-direction arrays (k_search_apply reads across the slab boundary): [0] = rank-1,
  // [1] = rank+1; sa / sb = the peer's arrays that were S->s / S->s2 at export time (the two swap roles every iteration,
  // on every rank alike)
  double *nb_z[2], *nb_sa[2], *nb_sb[2];
  void* nb_base[2][3];                // the mapped allocations (to close)
  double* orig_s;                     // this rank's S->s at export time
  int have_arrays;
};

#define P2P_NHANDLES 4                 // mailbox, z, s, s2

// stand-alone in-place all-reduce of one double (the all_zero(r) flag, euler_comm_ops.allreduce, the self-test);
// the reductions of the PCG loop do the same inside their own last block (k_pcg.hip block_finish)
__global__ __launch_bounds__(64) void k_p2p_allreduce(PcgScalars* sc, double* val, int is_max) {
  const double v = threadIdx.x == 0 ? *val : 0.0;
  const double t = is_max ? p2p_allreduce_block<true>(sc, v) : p2p_allreduce_block<false>(sc, v);
  if (threadIdx.x == 0) *val = t;
}

// Ghost rows of a band-skewed vector: my lowest row becomes rank-1's ghost row above its slab, my highest row
// rank+1's ghost row below its slab.  One granule per double.  Mailbox layout behind the header:
// [parity][side: 0 = row arriving from rank-1, 1 = from rank+1][X] granules.  Rows are read from and written
// into the skewed array directly (row (band, lane), element skew_index(g, x, 64 band + lane)).
__global__ __launch_bounds__(1024) void k_p2p_halo(PcgScalars* sc, double* s, SkewGeom g, int band_lo, int band_hi) {
  __shared__ unsigned int s_seq;
  if (threadIdx.x == 0) s_seq = ++sc->p2p_halo_seq;
  __syncthreads();
  const unsigned int seq = s_seq;
  const int rank = sc->p2p_rank, n = sc->p2p_n, X = g.X, par = seq & 1;
  const size_t row = (size_t)X * 16, side_off = P2P_HDR_BYTES + (size_t)par * 2 * row;
  if (rank > 0) {
    char* dst = static_cast<char*>(sc->p2p_boxes[rank - 1]) + side_off + row;          // their "from rank+1" row
    for (int x = threadIdx.x; x < X; x += 1024) p2p_store(dst + (size_t)x * 16, s[skew_index(g, x, 64 * band_lo)], seq);
  }
  if (rank + 1 < n) {
    char* dst = static_cast<char*>(sc->p2p_boxes[rank + 1]) + side_off;                // their "from rank-1" row
    for (int x = threadIdx.x; x < X; x += 1024) p2p_store(dst + (size_t)x * 16, s[skew_index(g, x, 64 * band_hi - 1)], seq);
  }
  const char* own = static_cast<const char*>(sc->p2p_boxes[rank]) + side_off;
  for (int side = 0; side < 2; ++side) {
    if (side == 0 ? rank == 0 : rank + 1 >= n) continue;
    const int y = side == 0 ? 64 * band_lo - 1 : 64 * band_hi;
    for (int x = threadIdx.x; x < X; x += 1024) {
      double v = 0.0;
      unsigned int spins = 0;
      while (!p2p_poll(own + side * row + (size_t)x * 16, seq, &v)) {
        if (++spins > P2P_SPIN_LIMIT) { atomicExch(sc->p2p_error, 3); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      s[skew_index(g, x, y)] = v;
    }
  }
}
// the same for the rows of euler_comm_ops.halo (contiguous buffers)
__global__ __launch_bounds__(1024) void k_p2p_halo_rows(PcgScalars* sc, int X, const double* send_lo, const double* send_hi, double* recv_lo, double* recv_hi) {
  __shared__ unsigned int s_seq;
  if (threadIdx.x == 0) s_seq = ++sc->p2p_halo_seq;
  __syncthreads();
  const unsigned int seq = s_seq;
  const int rank = sc->p2p_rank, n = sc->p2p_n, par = seq & 1;
  const size_t row = (size_t)X * 16, side_off = P2P_HDR_BYTES + (size_t)par * 2 * row;
  if (rank > 0) {
    char* dst = static_cast<char*>(sc->p2p_boxes[rank - 1]) + side_off + row;
    for (int x = threadIdx.x; x < X; x += 1024) p2p_store(dst + (size_t)x * 16, send_lo[x], seq);
  }
  if (rank + 1 < n) {
    char* dst = static_cast<char*>(sc->p2p_boxes[rank + 1]) + side_off;
    for (int x = threadIdx.x; x < X; x += 1024) p2p_store(dst + (size_t)x * 16, send_hi[x], seq);
  }
  const char* own = static_cast<const char*>(sc->p2p_boxes[rank]) + side_off;
  for (int side = 0; side < 2; ++side) {
    if (side == 0 ? rank == 0 : rank + 1 >= n) continue;
    double* out = side == 0 ? recv_lo : recv_hi;
    for (int x = threadIdx.x; x < X; x += 1024) {
      double v = 0.0;
      unsigned int spins = 0;
      while (!p2p_poll(own + side * row + (size_t)x * 16, seq, &v)) {
        if (++spins > P2P_SPIN_LIMIT) { atomicExch(sc->p2p_error, 3); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      out[x] = v;
    }
  }
}

static int p2p_allreduce(void* ctx, void* dev_f64, int32_t count, int32_t is_max) {
  euler_sim* S = static_cast<euler_sim*>(ctx);
  P2PState* p = static_cast<P2PState*>(S->p2p);
  p->calls[0]++;
  for (int k = 0; k < count; ++k)
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(64), 0, S->stream, S->sc, static_cast<double*>(dev_f64) + k, (int)is_max);
  return 0;
}

static int p2p_halo(void* ctx, void* send_lo, void* send_hi, void* recv_lo, void* recv_hi, int32_t count) {
  euler_sim* S = static_cast<euler_sim*>(ctx);
  P2PState* p = static_cast<P2PState*>(S->p2p);
  p->calls[1]++;
  if (count != p->X) { eu_set_error("p2p halo: row of %d doubles, mailbox sized for %d", (int)count, p->X); return -1; }
  hipLaunchKernelGGL(k_p2p_halo_rows, dim3(1), dim3(1024), 0, S->stream, S->sc, p->X, static_cast<const double*>(send_lo),
                     static_cast<const double*>(send_hi), static_cast<double*>(recv_lo), static_cast<double*>(recv_hi));
  return 0;
}
int eu_p2p_halo_skewed(euler_sim* S, double* s_skewed) {
  P2PState* p = static_cast<P2PState*>(S->p2p);
  p->calls[1]++;
  hipLaunchKernelGGL(k_p2p_halo, dim3(1), dim3(1024), 0, S->stream, S->sc, s_skewed, S->geom, S->band_lo, S->band_hi);
  return EULER_OK;
}
// bulk transfers: the communicator underneath, with its own context
static int p2p_chain(void* ctx, void* dev_ptr, int64_t nbytes, int32_t src, int32_t dst) {
  P2PState* p = static_cast<P2PState*>(static_cast<euler_sim*>(ctx)->p2p);
  return p->base.chain(p->base.ctx, dev_ptr, nbytes, src, dst);
}
static int p2p_allgather(void* ctx, void* dev_base, const int64_t* off, const int64_t* cnt) {
  P2PState* p = static_cast<P2PState*>(static_cast<euler_sim*>(ctx)->p2p);
  return p->base.allgather(p->base.ctx, dev_base, off, cnt);
}

// mailbox: header | ghost rows [parity][side][X] granules | band hand-off rows [direction][epoch parity][gran_stride] granule pairs
static size_t p2p_xgran_offset(const euler_sim* S, int backward, int parity) {
  return P2P_HDR_BYTES + (size_t)2 * 2 * S->X * 16 + ((size_t)backward * 2 + parity) * S->gran_stride * 16;
}
static size_t p2p_box_bytes(const euler_sim* S) { return p2p_xgran_offset(S, 2, 0); }

// Exact coupling: the forward / factor sweeps hand on upwards (rank -> rank + 1), the backward sweep downwards.  Rows
// alternate with the sweep epoch: two sweeps of one direction in a row (factor, then forward) never share a row, and a
// row comes round again only after a sweep of the other direction, which cannot finish on this rank before the
// neighbour has finished reading (its own hand-off to us comes from the kernel behind that read).
void eu_p2p_xgran(euler_sim* S, int backward, const unsigned long long** in, unsigned long long** out) {
  P2PState* p = static_cast<P2PState*>(S->p2p);
  const size_t off = p2p_xgran_offset(S, backward, (int)(S->epoch & 1u));
  const int next = backward ? p->rank - 1 : p->rank + 1;
  *in = reinterpret_cast<const unsigned long long*>(static_cast<char*>(p->box) + off);
  *out = next >= 0 && next < p->n ? reinterpret_cast<unsigned long long*>(static_cast<char*>(p->peer[next]) + off) : nullptr;
}

void eu_p2p_release(euler_sim* S) {
  P2PState* p = static_cast<P2PState*>(S->p2p);
  if (!p) return;
  if (S->stream) (void)hipStreamSynchronize(S->stream);
  for (int k = 0; k < p->n; ++k)
    if (p->peer[k] && p->peer[k] != p->box) (void)hipIpcCloseMemHandle(p->peer[k]);
  for (int side = 0; side < 2; ++side)
    for (int a = 0; a < 3; ++a)
      if (p->nb_base[side][a]) (void)hipIpcCloseMemHandle(p->nb_base[side][a]);
  if (p->peer_dev) (void)hipFree(p->peer_dev);
  if (p->box) (void)hipFree(p->box);
  if (S->has_comm && S->comm.ctx == S && p->base.allreduce) S->comm = p->base;   // back to the communicator underneath
  S->p2p_on = 0;
  free(p);
  S->p2p = nullptr;
}

// Step 1 (every rank): allocate the mailbox, hand out its IPC handle (EULER_P2P_HANDLE_BYTES bytes).
extern "C" int euler_p2p_export(euler_sim* S, void* handle_out, int32_t cap) {
  if (!S || !handle_out || cap < (int32_t)(P2P_NHANDLES * sizeof(hipIpcMemHandle_t))) { eu_set_error("euler_p2p_export: need %d bytes", (int)(P2P_NHANDLES * sizeof(hipIpcMemHandle_t))); return EULER_EINVAL; }
  HIPCHK(hipSetDevice(S->cfg.device));
  P2PState* p = static_cast<P2PState*>(S->p2p);
  if (!p) {
    p = static_cast<P2PState*>(calloc(1, sizeof(P2PState)));
    if (!p) return EULER_ENOMEM;
    p->box_bytes = p2p_box_bytes(S);
    p->X = S->X;
    hipError_t e = hipExtMallocWithFlags(&p->box, p->box_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { free(p); return eu_hip_fail(e, "hipExtMallocWithFlags(mailbox, fine-grained)", __FILE__, __LINE__); }
    e = hipMemset(p->box, 0, p->box_bytes);   // tag 0 is never used by an exchange
    if (e != hipSuccess) { (void)hipFree(p->box); free(p); return eu_hip_fail(e, "hipMemset(mailbox)", __FILE__, __LINE__); }
    S->p2p = p;
  }
  hipIpcMemHandle_t h[P2P_NHANDLES];
  HIPCHK(hipIpcGetMemHandle(&h[0], p->box));
  // the solver arrays start EU_SKEW_SLACK elements into their allocations (driver.hip)
  // (if an array cannot be exported its handle stays zero: the neighbours' open fails, and all ranks agree not to fuse)
  memset(&h[1], 0, 3 * sizeof(hipIpcMemHandle_t));
  (void)hipIpcGetMemHandle(&h[1], S->z - EU_SKEW_SLACK);
  (void)hipIpcGetMemHandle(&h[2], S->s - EU_SKEW_SLACK);
  (void)hipIpcGetMemHandle(&h[3], S->s2 - EU_SKEW_SLACK);
  p->orig_s = S->s;
  memcpy(handle_out, h, sizeof h);
  return EULER_OK;
}

// Step 2 (every rank, collectively, after a communicator has been installed): map the peers' mailboxes, prove
// the path with one all-reduce of known values, then route the scalar all-reduces and the ghost rows over it.
extern "C" int euler_p2p_connect(euler_sim* S, const void* handles, int32_t nranks) {
  if (!S || !handles) return EULER_EINVAL;
  P2PState* p = static_cast<P2PState*>(S->p2p);
  if (!p) { eu_set_error("euler_p2p_connect: call euler_p2p_export first"); return EULER_ESTATE; }
  if (!S->has_comm || S->comm.nranks != nranks || S->comm.ctx == S) { eu_set_error("euler_p2p_connect: install a communicator of %d ranks first", (int)nranks); return EULER_ESTATE; }
  if (nranks > P2P_MAXR) { eu_set_error("euler_p2p_connect: at most %d ranks", P2P_MAXR); return EULER_EINVAL; }
  if (S->cfg.precond == EULER_PRECOND_IC0_TILE2 || S->cfg.precond == EULER_PRECOND_IC0_TILE_MG) {      // (the same configuration on every rank: everybody returns here)
    eu_set_error("euler_p2p_connect: the coarse-correction preconditioners run on the default transport, not over the mailboxes"); return EULER_EINVAL;
  }
  HIPCHK(hipSetDevice(S->cfg.device));
  p->rank = S->comm.rank; p->n = nranks;
  const hipIpcMemHandle_t* hs = static_cast<const hipIpcMemHandle_t*>(handles);   // P2P_NHANDLES per rank
  // Phase 1, map: a rank whose mapping fails does NOT leave - all ranks first agree on the outcome over the communicator
  // installed before (a collective every rank reaches), so that nobody spins on a mailbox whose owner has given up.
  int map_ok = 1;
  hipError_t map_err = hipSuccess;
  for (int k = 0; k < nranks; ++k) {
    if (k == p->rank) { p->peer[k] = p->box; continue; }
    if (p->peer[k]) { (void)hipIpcCloseMemHandle(p->peer[k]); p->peer[k] = nullptr; }   // a retry maps afresh
    hipIpcMemHandle_t h;
    memcpy(&h, &hs[(size_t)k * P2P_NHANDLES], sizeof h);
    hipError_t e = hipIpcOpenMemHandle(&p->peer[k], h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { p->peer[k] = nullptr; map_ok = 0; map_err = e; }
  }
  // A retry must not be satisfied by the granules of a failed attempt (the sequence numbers start over): wipe the own
  // mailbox; the all-reduce below is also the barrier behind which no peer of the OLD attempt can still be writing.
  HIPCHK(hipMemsetAsync(p->box, 0, P2P_HDR_BYTES, S->stream));
  {
    double* flag = reinterpret_cast<double*>(S->halo_buf);
    const double mine_ok = map_ok ? 0.0 : 1.0;      // max over ranks of "I failed"
    double any_failed = 1.0;
    HIPCHK(hipMemcpyAsync(flag, &mine_ok, 8, hipMemcpyHostToDevice, S->stream));
    if (S->comm.allreduce(S->comm.ctx, flag, 1, 1) != 0) { eu_set_error("euler_p2p_connect: the communicator's all-reduce failed"); return EULER_ECOMM; }
    HIPCHK(hipMemcpyAsync(&any_failed, flag, 8, hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    if (any_failed != 0.0) {
      if (!map_ok) return eu_hip_fail(map_err, "hipIpcOpenMemHandle(peer mailbox)", __FILE__, __LINE__);
      eu_set_error("euler_p2p_connect: a peer could not map the mailboxes; staying on the installed communicator");
      return EULER_ECOMM;
    }
  }
  if (!p->peer_dev) HIPCHK(hipMalloc((void**)&p->peer_dev, sizeof(void*) * P2P_MAXR));   // (a failed earlier attempt may have left one)
  HIPCHK(hipMemcpy(p->peer_dev, p->peer, sizeof(void*) * P2P_MAXR, hipMemcpyHostToDevice));
  // the link lives in the device-resident PCG scalars (k_pcg_reset never touches these fields)
  struct { void** boxes; int* error; int rank, n; unsigned int seq, hseq; } link = {p->peer_dev, &S->ms->error, p->rank, p->n, 0u, 0u};
  static_assert(sizeof(link) == sizeof(PcgScalars) - offsetof(PcgScalars, p2p_boxes), "PcgScalars p2p tail");
  HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(S->sc) + offsetof(PcgScalars, p2p_boxes), &link, sizeof link, hipMemcpyHostToDevice, S->stream));
  // self-test on the real path: sum and max of rank + 1
  double* probe = reinterpret_cast<double*>(S->halo_buf);
  const double mine = (double)(p->rank + 1);
  double got[2] = {0, 0};
  for (int is_max = 0; is_max < 2; ++is_max) {
    HIPCHK(hipMemcpyAsync(probe, &mine, 8, hipMemcpyHostToDevice, S->stream));
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(64), 0, S->stream, S->sc, probe, is_max);
    HIPCHK(hipMemcpyAsync(&got[is_max], probe, 8, hipMemcpyDeviceToHost, S->stream));
  }
  int rc = eu_sync_marker_state(S);   // also picks up a timed-out wait
  if (rc == EULER_ETIMEOUT) {         // a peer never showed up: not fatal for the handle, the caller falls back
    (void)hipMemsetAsync(&S->ms->error, 0, sizeof(int), S->stream);
    (void)eu_sync_marker_state(S);
    eu_set_error("euler_p2p_connect: the self-test all-reduce timed out (a peer's mailbox never answered)");
    return EULER_ECOMM;
  }
  if (rc) return rc;
  if (got[0] != 0.5 * nranks * (nranks + 1) || got[1] != (double)nranks) {
    eu_set_error("euler_p2p_connect: self-test all-reduce returned %g / %g, expected %g / %g", got[0], got[1], 0.5 * nranks * (nranks + 1), (double)nranks);
    return EULER_ECOMM;
  }
  // the neighbouring slabs' z / s / s2 (fused search + apply_a across the slab boundary); failing here only costs the fusion
  p->have_arrays = 1;
  for (int side = 0; side < 2; ++side) {
    const int nb = side == 0 ? p->rank - 1 : p->rank + 1;
    if (nb < 0 || nb >= nranks) continue;
    for (int a = 0; a < 3; ++a) {
      if (p->nb_base[side][a]) { (void)hipIpcCloseMemHandle(p->nb_base[side][a]); p->nb_base[side][a] = nullptr; }
      hipIpcMemHandle_t h;
      memcpy(&h, &hs[(size_t)nb * P2P_NHANDLES + 1 + a], sizeof h);
      if (hipIpcOpenMemHandle(&p->nb_base[side][a], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { p->nb_base[side][a] = nullptr; p->have_arrays = 0; }
    }
    p->nb_z[side] = p->nb_base[side][0] ? static_cast<double*>(p->nb_base[side][0]) + EU_SKEW_SLACK : nullptr;
    p->nb_sa[side] = p->nb_base[side][1] ? static_cast<double*>(p->nb_base[side][1]) + EU_SKEW_SLACK : nullptr;
    p->nb_sb[side] = p->nb_base[side][2] ? static_cast<double*>(p->nb_base[side][2]) + EU_SKEW_SLACK : nullptr;
  }
  // The fusion changes which exchanges a rank performs, so it must be all ranks or none: agree over the mailboxes
  // (just proven).  OPT-IN: only with EULER_OPT_SLAB_FUSION set on every rank's handle.  The fused kernel reads the
  // neighbours' coarse-grained arrays across GPUs (write-through stores + system-scope loads); that has only ever run with
  // several processes on ONE device, so until a node run has validated it the ghost-row path is the default.
  {
    const double mine_ok = (p->have_arrays && S->opt[EULER_OPT_SLAB_FUSION] == 1) ? 1.0 : 0.0;
    double sum = 0.0;
    HIPCHK(hipMemcpyAsync(probe, &mine_ok, 8, hipMemcpyHostToDevice, S->stream));
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(64), 0, S->stream, S->sc, probe, 0);
    HIPCHK(hipMemcpyAsync(&sum, probe, 8, hipMemcpyDeviceToHost, S->stream));
    rc = eu_sync_marker_state(S);
    if (rc == EULER_ETIMEOUT) {       // a peer dropped out after the self-test: fall back like above, handle intact
      (void)hipMemsetAsync(&S->ms->error, 0, sizeof(int), S->stream);
      (void)eu_sync_marker_state(S);
      eu_set_error("euler_p2p_connect: a peer left before the ranks could agree on the fused search kernel");
      return EULER_ECOMM;
    }
    if (rc) return rc;
    p->have_arrays = sum == (double)nranks;
  }
  p->base = S->comm;
  S->comm.ctx = S;
  S->comm.allreduce = p2p_allreduce;
  S->comm.halo = p2p_halo;
  S->comm.chain = p2p_chain;
  S->comm.allgather = p2p_allgather;
  S->comm.exchange = nullptr;   // (the mailboxes carry the per-iteration traffic inside the kernels)
  S->p2p_on = 1;
  return EULER_OK;
}

int eu_p2p_has_neighbour_arrays(const euler_sim* S) {
  const P2PState* p = static_cast<const P2PState*>(S->p2p);
  // only with EULER_SLAB_FUSION=1 in every rank's environment (evaluated in euler_p2p_connect); the default keeps the
  // ghost-row exchange + separate update_search instead of reading across the slab boundary
  return S->p2p_on && p && p->have_arrays && !S->slab_on;   // agreed by all ranks in euler_p2p_connect (row-slab handles: arrays are windows, no mapping)
}
void eu_p2p_neighbour_arrays(euler_sim* S, const double** z_dn, const double** s_dn, const double** z_up, const double** s_up) {
  P2PState* p = static_cast<P2PState*>(S->p2p);
  const bool a_is_s = S->s == p->orig_s;          // which of the two search-direction arrays is "s" right now (all ranks alike)
  *z_dn = p->nb_z[0]; *s_dn = a_is_s ? p->nb_sa[0] : p->nb_sb[0];
  *z_up = p->nb_z[1]; *s_up = a_is_s ? p->nb_sa[1] : p->nb_sb[1];
}

extern "C" int euler_p2p_disconnect(euler_sim* S) {
  if (!S) return EULER_EINVAL;
  eu_p2p_release(S);
  return EULER_OK;
}

extern "C" int euler_p2p_calls(euler_sim* S, uint64_t out[2]) {
  if (!S || !out) return EULER_EINVAL;
  P2PState* p = static_cast<P2PState*>(S->p2p);
  out[0] = p ? p->calls[0] : 0; out[1] = p ? p->calls[1] : 0;
  return EULER_OK;
}
