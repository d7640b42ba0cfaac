// k_slab.hip — row slabs for EVERY stage of sim_step() (euler_config.slab_nranks >= 1).
//
// One process per GPU.  Rank g owns the rows of its 64-row bands, [row_lo, row_hi), of every field and the markers whose
// floor(y) lies there; nothing of the rest of the grid exists on this GPU (per-rank memory ~ 1/G).  The stage kernels are
// the single-GPU ones, launched over this rank's rows: they index with GLOBAL (x, y) through base pointers shifted by the
// window's first row (euler_dev.h), so no kernel knows about slabs.  What is new lives here:
//
//   ghost rows     u, v: 1 below / 1 above; the two count grids: 1 below / 2 above (the V-typed masks look one row further
//                  up); vtmp: 1 below (the divergence of the lowest own row); p: 1 above (the velocity update).  Exchanged
//                  with the two neighbouring ranks at the points of the substep where their producers have run.
//   dt             max u^2, max v^2 over the own rows -> all-reduce(max) -> the same float dt everywhere (max is exact).
//   marker order   The reference's marker ARRAY ORDER is observable (advect_markers shortens dt for every LATER marker,
//                  main.c:501,518; swap-with-last deletion, main.c:112; source appends, main.c:288).  Every local marker
//                  therefore carries its KEY = its index in the reference's g_markers; the local order means nothing.
//                    * dt chain: the few candidate collisions of all ranks are gathered, sorted by key and replayed by
//                      every rank (k_event_chain); a marker then moves with the dt valid at its key.
//                    * deletion: the keys deleted on all ranks are gathered and sorted; a survivor with key >= n - D takes
//                      the hole the sequential swap-with-last loop would have put it in (k_rekey: the rank formula of
//                      k_compact_markers on keys).
//                    * sources: eligible cells are counted per rank (rows are contiguous, so rank order IS row-major
//                      order); rank g's k-th eligible cell appends key n + (cells of lower ranks) + k and takes the
//                      stream's draws at that position.
//   migration      markers that left the own rows go to the neighbour with their keys (at most one slab away: the CFL
//                  bound is 0.75 cell, a slab at least 64 rows).
//
// Every exchange goes through the four operations of euler_comm_ops (include/euler.h): RCCL over xGMI on a node,
// torch.distributed / gloo in the tests (2-4 ranks sharing one GPU).  Sizes are bounded per substep (SL_* below); an overflow
// raises a sticky device error (EULER_ESTATE at the next sync) instead of corrupting anything.
#include "euler_dev.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

// Exchange capacities per rank and substep grow with the grid (SlabScratch::ev_cap / del_cap): what can hit a wall or
// fall into a sink in one substep is a front of <= 0.75 cell times a length of the order of X, four markers to a cell.
//   dt-chain candidates: 1024 + X / 2   (rare: a marker must hit a solid after crossing a cell)
//   deletions:           4096 + 8 X     (sinks / solids)
// An overflow anywhere is raised on EVERY rank - all ranks read the same counters in the gathered blocks, and local
// overflows (migration, marker capacity) travel with the next exchange of the error word (slab_error_sync) - so the job
// fails collectively instead of hanging in the next exchange.
#define SL_MAXR 16

struct SlMigrant { float x, y; unsigned int key; };

struct SlabScratch {
  int R, rank;
  unsigned int ev_cap, del_cap, sort_cap;   // per-rank capacities (above); sort_cap = power of two >= R * max(ev_cap, del_cap)
  size_t blk;                     // bytes of one rank's block in the all-gather buffer
  char* xg;                       // [R][blk]: {u64 count; payload}
  unsigned long long* sortbuf;    // [SL_SORT_CAP]
  float *ev_th, *ev_de;           // merged candidates
  unsigned int* d_sorted;         // deleted keys of all ranks, ascending
  size_t mig_cap, buf_doubles;    // migrants per direction; size of each of the four neighbour buffers in doubles
  double *send_lo, *send_hi, *recv_lo, *recv_hi;
  double* vec;                    // [R + 8] small all-reduce vector
  double* vec2;                   // [2 R] the partition check
  unsigned long long* mask2;      // second marker bit mask (local removals)
  std::vector<int64_t> ag_off, ag_cnt;
  unsigned long long global_sources;   // source cells of all ranks (fixed by the scenario)
};

#define COMM_CALL(expr) do { if ((expr) != 0) { eu_set_error("communicator callback failed: %s", #expr); return EULER_ECOMM; } } while (0)

int eu_launch_dt(euler_sim* S, float frame_time_left);     // k_grid.hip
int eu_launch_project(euler_sim* S, float dt);

// ------------------------------------------------------------------------------------------ allocation
int eu_slab_alloc(euler_sim* S) {
  SlabScratch* s = new (std::nothrow) SlabScratch();
  if (!s) return EULER_ENOMEM;
  S->slab = s;
  s->R = S->cfg.slab_nranks; s->rank = S->cfg.slab_rank;
  if (s->R > SL_MAXR) { eu_set_error("row slabs: at most %d ranks", SL_MAXR); return EULER_EINVAL; }
  s->ev_cap = 1024u + (unsigned int)S->X / 2u;
  s->del_cap = 4096u + 8u * (unsigned int)S->X;
  if (const char* e = getenv("EULER_SLAB_CAPS")) {      // tests: tiny capacities to drive the overflow path
    unsigned int a = 0, b = 0;
    if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1 && b >= 1) { s->ev_cap = a; s->del_cap = b; }
  }
  const size_t merged = (size_t)s->R * (s->ev_cap > s->del_cap ? s->ev_cap : s->del_cap);
  s->sort_cap = 1; while (s->sort_cap < merged) s->sort_cap <<= 1;
  const size_t ev_blk = 8 + (size_t)16 * s->ev_cap, del_blk = 8 + (size_t)4 * s->del_cap;
  s->blk = ((ev_blk > del_blk ? ev_blk : del_blk) + 15) & ~(size_t)15;
  s->mig_cap = (size_t)16 * S->X + 4096;
  const size_t mig_bytes = 8 + s->mig_cap * sizeof(SlMigrant), row_bytes = (size_t)S->X * 4 * 2 * 2 + 64;   // ghost rows: <= 2 fields x 2 rows of floats
  s->buf_doubles = ((mig_bytes > row_bytes ? mig_bytes : row_bytes) + 7) / 8;
  HIPCHK(hipMalloc((void**)&s->xg, s->blk * s->R));
  HIPCHK(hipMemset(s->xg, 0, s->blk * s->R));
  HIPCHK(hipMalloc((void**)&s->sortbuf, (size_t)s->sort_cap * 8));
  HIPCHK(hipMalloc((void**)&s->ev_th, (size_t)s->sort_cap * 4));
  HIPCHK(hipMalloc((void**)&s->ev_de, (size_t)s->sort_cap * 4));
  HIPCHK(hipMalloc((void**)&s->d_sorted, (size_t)s->sort_cap * 4));
  for (double** b : {&s->send_lo, &s->send_hi, &s->recv_lo, &s->recv_hi}) {
    HIPCHK(hipMalloc((void**)b, s->buf_doubles * 8));
    HIPCHK(hipMemset(*b, 0, s->buf_doubles * 8));
  }
  HIPCHK(hipMalloc((void**)&s->vec, (SL_MAXR + 8) * 8));
  HIPCHK(hipMalloc((void**)&s->vec2, 2 * SL_MAXR * 8));
  HIPCHK(hipMalloc((void**)&s->mask2, ((S->max_markers + 63) / 64) * 8));
  s->ag_off.resize(s->R); s->ag_cnt.resize(s->R);
  for (int r = 0; r < s->R; ++r) { s->ag_off[r] = (int64_t)r * (int64_t)s->blk; s->ag_cnt[r] = (int64_t)s->blk; }
  return EULER_OK;
}

size_t eu_slab_bytes(const euler_sim* S) {
  const SlabScratch* s = S->slab;
  if (!s) return 0;
  return s->blk * s->R + (size_t)s->sort_cap * 20 + 4 * s->buf_doubles * 8 + ((S->max_markers + 63) / 64) * 8 + (3 * SL_MAXR + 8) * 8;
}

void eu_slab_release(euler_sim* S) {
  SlabScratch* s = S->slab;
  if (!s) return;
  void* dev[] = {s->xg, s->sortbuf, s->ev_th, s->ev_de, s->d_sorted, s->send_lo, s->send_hi, s->recv_lo, s->recv_hi, s->vec, s->vec2, s->mask2};
  for (void* p : dev) if (p) (void)hipFree(p);
  delete s;
  S->slab = nullptr;
}

// ------------------------------------------------------------------------------------------ the partition
// Collective, at communicator install: every rank deposits its band range in an all-reduced vector and checks that the ranges tile
// [0, nbands) in rank order - an explicit partition (euler_config.slab_band_lo / hi) is the caller's, and a gap or an overlap
// would silently lose rows.
__global__ void k_partition_fill(double* v, int R, int rank, int lo, int hi) {
  for (int k = threadIdx.x; k < 2 * R; k += blockDim.x) v[k] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) { v[rank] = (double)lo; v[R + rank] = (double)hi; }
}
int eu_slab_check_partition(euler_sim* S) {
  SlabScratch* s = S->slab;
  S->part_lo[0] = S->band_lo; S->part_hi[0] = S->band_hi;
  if (!s || s->R < 2) return EULER_OK;
  hipLaunchKernelGGL(k_partition_fill, dim3(1), dim3(64), 0, S->stream, s->vec2, s->R, s->rank, S->band_lo, S->band_hi);
  COMM_CALL(S->comm.allreduce(S->comm.ctx, s->vec2, 2 * s->R, 0));
  double h[2 * SL_MAXR];
  HIPCHK(hipMemcpyAsync(h, s->vec2, sizeof(double) * 2 * s->R, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  bool ok = h[0] == 0.0 && h[2 * s->R - 1] == (double)S->geom.nbands;
  for (int r = 0; r < s->R; ++r) ok = ok && h[s->R + r] > h[r] && (r == 0 || h[r] == h[s->R + r - 1]);
  if (!ok) {
    char txt[256]; int n = 0;
    for (int r = 0; r < s->R && n < 230; ++r) n += snprintf(txt + n, sizeof txt - n, " [%d,%d)", (int)h[r], (int)h[s->R + r]);
    eu_set_error("row slabs: the ranks' band ranges do not tile the %d bands in rank order:%s", S->geom.nbands, txt);
    S->has_comm = 0;
    return EULER_EINVAL;
  }
  for (int r = 0; r < s->R && r < 64; ++r) { S->part_lo[r] = (int)h[r]; S->part_hi[r] = (int)h[s->R + r]; }
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------ ghost rows
// One exchange of the ghost rows of up to 4 row-major fields: my lowest `hi` own rows of each go to the rank below (they
// are its ghost rows above), my highest `lo` own rows go up; what arrives lands in my `lo` ghost rows below / `hi` above.
struct GhostField { void* base; int elem; int lo, hi; };   // base: the shifted (global-indexed) pointer; ghost rows below / above

static int exchange_rows(euler_sim* S, const GhostField* f, int nf) {
  SlabScratch* s = S->slab;
  const size_t X = S->X;
  const bool has_lo = s->rank > 0, has_hi = s->rank + 1 < s->R;
  hipStream_t st = S->stream;
  size_t o_dn = 0, o_up = 0;
  for (int k = 0; k < nf; ++k) {
    char* b = static_cast<char*>(f[k].base);
    const size_t rb = X * f[k].elem;
    // (a slab may own fewer rows than the neighbour has ghost rows - the grid's last band can be a single row: send what exists,
    // the rest of the slot as zeros; the receiver drops rows beyond the grid anyway)
    const int own = S->row_hi - S->row_lo;
    const int n_dn = f[k].hi < own ? f[k].hi : own, n_up = f[k].lo < own ? f[k].lo : own;
    if (has_lo && f[k].hi) {
      if (n_dn < f[k].hi) HIPCHK(hipMemsetAsync(reinterpret_cast<char*>(s->send_lo) + o_dn, 0, f[k].hi * rb, st));
      HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(s->send_lo) + o_dn, b + (size_t)S->row_lo * rb, n_dn * rb, hipMemcpyDeviceToDevice, st));
    }
    if (has_hi && f[k].lo) {
      if (n_up < f[k].lo) HIPCHK(hipMemsetAsync(reinterpret_cast<char*>(s->send_hi) + o_up, 0, f[k].lo * rb, st));
      HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(s->send_hi) + o_up, b + (size_t)(S->row_hi - n_up) * rb, n_up * rb, hipMemcpyDeviceToDevice, st));
    }
    o_dn += f[k].hi * rb; o_up += f[k].lo * rb;
  }
  const size_t bytes = o_dn > o_up ? o_dn : o_up;
  COMM_CALL(S->bulk.halo(S->bulk.ctx, s->send_lo, s->send_hi, s->recv_lo, s->recv_hi, (int32_t)((bytes + 7) / 8)));
  size_t i_lo = 0, i_hi = 0;      // from below come the neighbour's TOP rows (my ghost rows below: `lo` of them), from above its bottom rows
  for (int k = 0; k < nf; ++k) {
    char* b = static_cast<char*>(f[k].base);
    const size_t rb = X * f[k].elem;
    if (has_lo && f[k].lo) HIPCHK(hipMemcpyAsync(b + (size_t)(S->row_lo - f[k].lo) * rb, reinterpret_cast<char*>(s->recv_lo) + i_lo, f[k].lo * rb, hipMemcpyDeviceToDevice, st));
    if (has_hi && f[k].hi) {
      const int rows = S->row_hi + f[k].hi <= S->Y ? f[k].hi : S->Y - S->row_hi;      // (the grid's top: fewer rows exist)
      if (rows > 0) HIPCHK(hipMemcpyAsync(b + (size_t)S->row_hi * rb, reinterpret_cast<char*>(s->recv_hi) + i_hi, rows * rb, hipMemcpyDeviceToDevice, st));
    }
    i_lo += f[k].lo * rb; i_hi += f[k].hi * rb;
  }
  return EULER_OK;
}

static int exchange_uv(euler_sim* S) {
  const GhostField f[2] = {{S->u, 4, 1, 1}, {S->v, 4, 1, 1}};
  return exchange_rows(S, f, 2);
}
int eu_slab_exchange_uv(euler_sim* S) {
  if (!S->has_comm) { eu_set_error("row-slab handle without a communicator"); return EULER_ESTATE; }
  return exchange_uv(S);
}
static int exchange_counts(euler_sim* S) {
  const GhostField f[2] = {{S->count, 1, EU_GHOST_LO, EU_GHOST_HI}, {S->prev_count, 1, EU_GHOST_LO, EU_GHOST_HI}};
  return exchange_rows(S, f, 2);
}
// the dye (--rainbow): one ghost row of g_r, g_g, g_b either side (extrapolate(.., P) reads the 3x3, advect_p back-traces <= 0.75 cell)
int eu_slab_exchange_dye(euler_sim* S) {
  if (!S->dye[0]) return EULER_OK;
  if (!S->has_comm) { eu_set_error("row-slab handle without a communicator"); return EULER_ESTATE; }
  const GhostField f[3] = {{S->dye[0], 4, 1, 1}, {S->dye[1], 4, 1, 1}, {S->dye[2], 4, 1, 1}};
  return exchange_rows(S, f, 3);
}
static int exchange_vtmp(euler_sim* S) {
  const GhostField f[1] = {{S->vtmp, 4, 1, 0}};
  return exchange_rows(S, f, 1);
}

// ------------------------------------------------------------------------------------------ timestep
// (the sticky error word of every rank rides along: an overflow on one rank becomes every rank's error before the substep starts)
__global__ void k_maxsq_to_vec(const MarkerState* ms, double* v) {
  v[0] = (double)__uint_as_float(ms->max_u2_bits); v[1] = (double)__uint_as_float(ms->max_v2_bits); v[2] = (double)ms->error;
}
__global__ void k_maxsq_from_vec(MarkerState* ms, const double* v) {
  ms->max_u2_bits = __float_as_uint((float)v[0]); ms->max_v2_bits = __float_as_uint((float)v[1]);
  if (!ms->error && v[2] != 0.0) ms->error = (int)v[2];
}
__global__ void k_error_to_vec(const MarkerState* ms, double* v) { v[0] = (double)ms->error; }
__global__ void k_error_from_vec(MarkerState* ms, const double* v) { if (!ms->error && v[0] != 0.0) ms->error = (int)v[0]; }
// end of a frame: one more exchange of the error word, so that the final sync of euler_step fails (or passes) on all ranks alike
int eu_slab_error_sync(euler_sim* S) {
  SlabScratch* s = S->slab;
  if (!S->has_comm) return EULER_OK;
  hipLaunchKernelGGL(k_error_to_vec, dim3(1), dim3(1), 0, S->stream, S->ms, s->vec);
  COMM_CALL(S->comm.allreduce(S->comm.ctx, s->vec, 1, 1));
  hipLaunchKernelGGL(k_error_from_vec, dim3(1), dim3(1), 0, S->stream, S->ms, s->vec);
  return EULER_OK;
}
// collective: the ranks' local status codes of a host-side step (file I/O of a snapshot) -> `*worst` = the largest on every rank, so that a
// failure on ONE rank (a missing part file, a full disk) is a failure everywhere instead of a hang in the next collective
int eu_slab_status_sync(euler_sim* S, int local_rc, int* worst) {
  *worst = local_rc;
  if (!S->has_comm || !S->slab) return EULER_OK;
  SlabScratch* s = S->slab;
  const double mine = local_rc != 0 ? 1.0 : 0.0;
  HIPCHK(hipMemcpyAsync(s->vec, &mine, sizeof mine, hipMemcpyHostToDevice, S->stream));
  COMM_CALL(S->bulk.allreduce(S->bulk.ctx, s->vec, 1, 1));
  double any = 0.0;
  HIPCHK(hipMemcpyAsync(&any, s->vec, sizeof any, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  if (any != 0.0 && local_rc == 0) *worst = EULER_EIO;
  return EULER_OK;
}
// collective: do n <= 4 host-side values (integers below 2^53) agree on every rank?  max(v) and max(-v) in one all-reduce
int eu_slab_same_everywhere(euler_sim* S, const double* vals, int n, int* same) {
  *same = 1;
  if (!S->has_comm || !S->slab || n > 4) return EULER_OK;
  SlabScratch* s = S->slab;
  double v[8];
  for (int k = 0; k < n; ++k) { v[2 * k] = vals[k]; v[2 * k + 1] = -vals[k]; }
  HIPCHK(hipMemcpyAsync(s->vec, v, sizeof(double) * 2 * n, hipMemcpyHostToDevice, S->stream));
  COMM_CALL(S->bulk.allreduce(S->bulk.ctx, s->vec, 2 * n, 1));
  HIPCHK(hipMemcpyAsync(v, s->vec, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  for (int k = 0; k < n; ++k) if (v[2 * k] != -v[2 * k + 1]) *same = 0;
  return EULER_OK;
}
int eu_slab_timestep(euler_sim* S, float frame_time_left) {   // k_maxsq over the own rows has been launched
  SlabScratch* s = S->slab;
  hipLaunchKernelGGL(k_maxsq_to_vec, dim3(1), dim3(1), 0, S->stream, S->ms, s->vec);
  COMM_CALL(S->comm.allreduce(S->comm.ctx, s->vec, 3, 1));
  hipLaunchKernelGGL(k_maxsq_from_vec, dim3(1), dim3(1), 0, S->stream, S->ms, s->vec);
  return eu_launch_dt(S, frame_time_left);
}

// ------------------------------------------------------------------------------------------ sorting (one workgroup)
// bitonic sort of n = 2^k 64-bit values in global memory by one workgroup (the merged lists are small)
__device__ void bitonic_sort_u64(unsigned long long* a, unsigned int n) {
  for (unsigned int k = 2; k <= n; k <<= 1)
    for (unsigned int j = k >> 1; j > 0; j >>= 1) {
      for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned int l = i ^ j;
        if (l > i) {
          const unsigned long long x = a[i], y = a[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[l] = x; }
        }
      }
      __syncthreads();
    }
}
__device__ __forceinline__ unsigned int next_pow2(unsigned int v) { unsigned int p = 1; while (p < v) p <<= 1; return p; }

// ------------------------------------------------------------------------------------------ advect_markers: the dt chain across ranks
struct SlEvent { unsigned int key, pad; float theta, delta; };

__global__ __launch_bounds__(256) void k_pack_events(const unsigned int* __restrict__ ev_idx, const float* __restrict__ theta,
                                                     const float* __restrict__ delta, const unsigned int* __restrict__ keys,
                                                     MarkerState* ms, char* block, unsigned int cap) {
  const unsigned int n_all = ms->n_events;
  const unsigned int n = n_all > cap ? cap : n_all;
  // the header carries the TRUE count: every rank sees the overflow in the gathered blocks (k_event_chain) and fails with it
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(block) = n_all;
  SlEvent* e = reinterpret_cast<SlEvent*>(block + 8);
  for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const unsigned int i = ev_idx[k];
    e[k] = SlEvent{keys[i], 0u, theta[i], delta[i]};
  }
}

// every rank: merge the candidates of all ranks, sort them by key, replay `if (theta < dt) dt -= delta` (main.c:497-501) in
// that order -> (key, dt after it) of the collisions that fire, exactly k_marker_walk's output on the reference's array
__global__ __launch_bounds__(1024) void k_event_chain(const char* __restrict__ xg, int R, size_t blk, unsigned long long* sortbuf,
                                                      float* __restrict__ th, float* __restrict__ de, unsigned int* __restrict__ act_key,
                                                      float* __restrict__ act_dt, MarkerState* ms, float dt0, unsigned int cap) {
  __shared__ unsigned int base[SL_MAXR + 1];
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int r = 0; r < R; ++r) {
      base[r] = t;
      const unsigned long long c = *reinterpret_cast<const unsigned long long*>(xg + (size_t)r * blk);
      if (c > cap) atomicExch(&ms->error, 16);      // some rank had more candidates than fit: every rank reads this and stops
      t += (unsigned int)(c < cap ? c : cap);
    }
    base[R] = t;
  }
  __syncthreads();
  const unsigned int K = base[R];
  if (K == 0) { if (threadIdx.x == 0) { ms->n_actual = 0; ms->dt_final = dt0; } return; }
  const unsigned int np = next_pow2(K);
  for (int r = 0; r < R; ++r) {
    const SlEvent* e = reinterpret_cast<const SlEvent*>(xg + (size_t)r * blk + 8);
    const unsigned int c = base[r + 1] - base[r];
    for (unsigned int k = threadIdx.x; k < c; k += blockDim.x) {
      const unsigned int slot = base[r] + k;
      sortbuf[slot] = ((unsigned long long)e[k].key << 32) | slot;
      th[slot] = e[k].theta; de[slot] = e[k].delta;
    }
  }
  for (unsigned int k = K + threadIdx.x; k < np; k += blockDim.x) sortbuf[k] = ~0ull;
  __syncthreads();
  bitonic_sort_u64(sortbuf, np);
  if (threadIdx.x == 0) {
    float dt = dt0;
    unsigned int M = 0;
    for (unsigned int k = 0; k < K; ++k) {
      const unsigned int slot = (unsigned int)(sortbuf[k] & 0xffffffffu);
      if (th[slot] < dt) { dt = dt - de[slot]; act_key[M] = (unsigned int)(sortbuf[k] >> 32); act_dt[M] = dt; ++M; }
    }
    ms->n_actual = M; ms->dt_final = dt; ms->total_dt_events += M;
  }
}

// ------------------------------------------------------------------------------------------ migration
// markers that left the own rows: appended (any order: their keys carry the reference's order) to the neighbour's buffer
__global__ __launch_bounds__(256) void k_migrate_out(const float2* __restrict__ pos, const unsigned int* __restrict__ keys,
                                                     MarkerState* ms, int row_lo, int row_hi, char* send_lo, char* send_hi, unsigned int cap) {
  const unsigned long long n = ms->n_loc;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float2 p = pos[i];
    const int y = (int)floorf(p.y / EU_H);
    if (y >= row_lo && y < row_hi) continue;
    char* buf = y < row_lo ? send_lo : send_hi;
    const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long*>(buf), 1ull);
    if (slot < cap) reinterpret_cast<SlMigrant*>(buf + 8)[slot] = SlMigrant{p.x, p.y, keys[i]};
    else atomicExch(&ms->error, 17);
  }
}
__global__ __launch_bounds__(256) void k_migrate_in(float2* __restrict__ pos, unsigned int* __restrict__ keys, MarkerState* ms,
                                                    const char* recv_lo, const char* recv_hi, int has_lo, int has_hi,
                                                    unsigned long long capacity) {
  const unsigned long long c_lo = has_lo ? *reinterpret_cast<const unsigned long long*>(recv_lo) : 0ull;
  const unsigned long long c_hi = has_hi ? *reinterpret_cast<const unsigned long long*>(recv_hi) : 0ull;
  const unsigned long long n = ms->n_loc, tot = c_lo + c_hi;
  if (n + tot > capacity) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicExch(&ms->error, 19); return; }
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < tot; k += (unsigned long long)gridDim.x * blockDim.x) {
    const SlMigrant m = k < c_lo ? reinterpret_cast<const SlMigrant*>(recv_lo + 8)[k] : reinterpret_cast<const SlMigrant*>(recv_hi + 8)[k - c_lo];
    pos[n + k] = make_float2(m.x, m.y);
    keys[n + k] = m.key;
  }
}
__global__ void k_migrate_done(MarkerState* ms, const char* recv_lo, const char* recv_hi, int has_lo, int has_hi) {
  const unsigned long long tot = (has_lo ? *reinterpret_cast<const unsigned long long*>(recv_lo) : 0ull) +
                                 (has_hi ? *reinterpret_cast<const unsigned long long*>(recv_hi) : 0ull);
  if (!ms->error) ms->n_loc += tot;
  ms->n_recv = (unsigned int)tot;
}

// ------------------------------------------------------------------------------------------ refresh_marker_counts across ranks
// bins the local markers (main.c:104-116).  A marker outside the own rows has been handed to the neighbour: it only leaves
// this rank's array.  A marker in a sink / solid cell is deleted from the reference's array: its key joins the list all
// ranks exchange.  Both kinds are flagged in rmmask for the local compaction.
__global__ __launch_bounds__(256) void k_bin_markers_slab(const float2* __restrict__ m, const unsigned int* __restrict__ keys, MarkerState* ms,
                                                          const uint8_t* __restrict__ sink, const uint8_t* __restrict__ solid,
                                                          unsigned int* count32, unsigned long long* __restrict__ rmmask, size_t mask_words, int X,
                                                          int row_lo, int row_hi, int win_lo, int win_h, char* del_block, unsigned int del_cap) {
  const unsigned long long n = ms->n_loc;
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool rm = false, live = false;
  size_t ct = 0;
  if (i < n) {
    const float2 p = m[i];
    const int x = (int)floorf(p.x / EU_H), y = (int)floorf(p.y / EU_H);
    if (y < row_lo || y >= row_hi) rm = true;
    else {
      const size_t c = (size_t)y * X + x;
      ct = (size_t)x * win_h + (y - win_lo);            // the column-major counters of the window (k_markers.hip)
      if ((sink[c] | solid[c]) != 0) {
        rm = true;
        const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long*>(del_block), 1ull);
        if (slot < del_cap) reinterpret_cast<unsigned int*>(del_block + 8)[slot] = keys[i];      // (the counter keeps the true number: k_merge_deleted raises the overflow on every rank)
      } else live = true;
    }
  }
  bin_aggregated(count32, live, ct);     // one atomic per run of lanes binning into the same cell
  const unsigned long long b = __ballot(rm);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < mask_words) rmmask[i >> 6] = b;      // every word the launch covers: zero behind the last marker
}

// every rank: the deleted keys of all ranks, ascending
__global__ __launch_bounds__(1024) void k_merge_deleted(char* xg, int R, size_t blk, int my_rank, unsigned long long* sortbuf,
                                                        unsigned int* __restrict__ d_sorted, MarkerState* ms, unsigned int del_cap) {
  __shared__ unsigned int base[SL_MAXR + 1];
  if (threadIdx.x == 0) {
    unsigned int t = 0;
    for (int r = 0; r < R; ++r) {
      base[r] = t;
      unsigned long long c = *reinterpret_cast<const unsigned long long*>(xg + (size_t)r * blk);
      if (c > del_cap) atomicExch(&ms->error, 18);      // every rank reads the same counters: the job fails as one
      t += (unsigned int)(c < del_cap ? c : del_cap);
    }
    base[R] = t;
    ms->n_del_glob = t;
  }
  __syncthreads();
  const unsigned int K = base[R];
  if (K) {
    const unsigned int np = next_pow2(K);
    for (int r = 0; r < R; ++r) {
      const unsigned int* e = reinterpret_cast<const unsigned int*>(xg + (size_t)r * blk + 8);
      const unsigned int c = base[r + 1] - base[r];
      for (unsigned int k = threadIdx.x; k < c; k += blockDim.x) sortbuf[base[r] + k] = e[k];
    }
    for (unsigned int k = K + threadIdx.x; k < np; k += blockDim.x) sortbuf[k] = ~0ull;
    __syncthreads();
    bitonic_sort_u64(sortbuf, np);
    for (unsigned int k = threadIdx.x; k < K; k += blockDim.x) d_sorted[k] = (unsigned int)sortbuf[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(xg + (size_t)my_rank * blk) = 0ull;   // own block: ready for the next refresh
}

// Swap-with-last deletion (main.c:112) on KEYS.  The sequential loop leaves survivors with index < n' = n - D in place and
// fills the k-th hole (ascending) with the k-th survivor taken from the back (k_compact_markers): a survivor with key
// j >= n' has (n - 1 - j) - (deleted keys above j) survivors behind it and takes the hole of that rank.
__global__ __launch_bounds__(256) void k_rekey(unsigned int* __restrict__ keys, const unsigned long long* __restrict__ rmmask,
                                               const unsigned int* __restrict__ d_sorted, const MarkerState* ms) {
  const unsigned long long n = ms->n, D = ms->n_del_glob;
  if (D == 0) return;
  const unsigned long long n1 = n - D, nl = ms->n_loc;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (unsigned long long)gridDim.x * blockDim.x) {
    if ((rmmask[i >> 6] >> (i & 63)) & 1ull) continue;
    const unsigned long long j = keys[i];
    if (j < n1) continue;
    unsigned long long lo = 0, hi = D;      // deleted keys <= j
    while (lo < hi) { const unsigned long long mid = (lo + hi) >> 1; if (d_sorted[mid] <= j) lo = mid + 1; else hi = mid; }
    const unsigned long long rank = (n - 1 - j) - (D - lo);
    keys[i] = d_sorted[rank];
  }
}

// the local array closes its gaps the same way (order-free: the keys carry the order): the k-th gap takes the k-th
// remaining marker from the back
__global__ __launch_bounds__(256) void k_compact_local(float2* m, unsigned int* keys, const unsigned int* __restrict__ rm_idx,
                                                       const unsigned long long* __restrict__ rmmask, const MarkerState* ms) {
  const unsigned long long n = ms->n_loc, D = ms->n_rm;
  if (D == 0) return;
  const unsigned long long n1 = n - D;
  for (unsigned long long j = n1 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * blockDim.x) {
    if ((rmmask[j >> 6] >> (j & 63)) & 1ull) continue;
    unsigned long long lo = 0, hi = D;
    while (lo < hi) { const unsigned long long mid = (lo + hi) >> 1; if (rm_idx[mid] <= j) lo = mid + 1; else hi = mid; }
    const unsigned long long rank = (n - 1 - j) - (D - lo);
    const unsigned int dst = rm_idx[rank];
    m[dst] = m[j];
    keys[dst] = keys[j];
  }
}

// pieces shared with the single-GPU stages (k_markers.hip)
int eu_marker_rotate_counts(euler_sim* S);
int eu_marker_narrow_counts(euler_sim* S);
int eu_marker_advect_a(euler_sim* S, float dt, unsigned long long n);
int eu_marker_advect_b(euler_sim* S, unsigned long long n, const unsigned int* keys);
int eu_source_fill(euler_sim* S);

static int slab_refresh(euler_sim* S, unsigned long long n_upper) {
  SlabScratch* s = S->slab;
  eu_marker_rotate_counts(S);
  const size_t mask_words = (S->max_markers + 63) / 64;
  char* my_block = s->xg + (size_t)s->rank * s->blk;
  HIPCHK(hipMemsetAsync(my_block, 0, 8, S->stream));     // the deletion counter (the block carried this substep's dt-chain candidates before)
  LAUNCH(S, KC_MARKER_BIN, k_bin_markers_slab, dim3(eu_blocks((size_t)n_upper + 1, 256)), dim3(256), S->markers[S->cur], S->keys[S->cur], S->ms,
         S->sink, S->solid, S->count32, s->mask2, mask_words, S->X, S->row_lo, S->row_hi, S->win_lo, S->win_hi - S->win_lo, my_block, s->del_cap);
  COMM_CALL(S->bulk.allgather(S->bulk.ctx, s->xg, s->ag_off.data(), s->ag_cnt.data()));
  LAUNCH(S, KC_MARKER_COMPACT, k_merge_deleted, dim3(1), dim3(1024), s->xg, s->R, s->blk, s->rank, s->sortbuf, s->d_sorted, S->ms, s->del_cap);
  LAUNCH(S, KC_MARKER_COMPACT, k_rekey, dim3(eu_blocks((size_t)n_upper + 1, 256, 4096)), dim3(256), S->keys[S->cur], s->mask2, s->d_sorted, S->ms);
  int rc = eu_ordered_select(S, s->mask2, (size_t)((n_upper + 63) / 64), S->sel_idx, &S->ms->n_rm);
  if (rc) return rc;
  LAUNCH(S, KC_MARKER_COMPACT, k_compact_local, dim3(256), dim3(256), S->markers[S->cur], S->keys[S->cur], S->sel_idx, s->mask2, S->ms);
  return eu_marker_narrow_counts(S);
}

// ------------------------------------------------------------------------------------------ update_fluid_sources across ranks
__global__ __launch_bounds__(256) void k_source_mask_rows(const uint8_t* __restrict__ source, const uint8_t* __restrict__ count, size_t i0, size_t n,
                                                          unsigned long long* __restrict__ mask) {
  const size_t li = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool e = li < n && source[i0 + li] && count[i0 + li] < 4;
  const unsigned long long b = __ballot(e);
  if ((threadIdx.x & 63) == 0 && (li >> 6) < ((n + 63) >> 6)) mask[li >> 6] = b;
}
__global__ void k_source_count_to_vec(const MarkerState* ms, double* vec, int R, int rank) {
  for (int r = 0; r < R; ++r) vec[r] = r == rank ? (double)ms->n_events : 0.0;
}
// the substep's bookkeeping on every rank alike (the latch and the stream position are global); the draws of the own cells are
// then filled in parallel by k_source_fill (k_markers.hip), which jumps into the one sequential stream at this rank's offset
__global__ void k_source_draws_slab(MarkerState* ms, const RngJump* __restrict__ J, const double* __restrict__ vec, int R, int rank) {
  unsigned long long n = ms->n;
  const unsigned long long cap = ms->max_markers - 1;
  const int exhausted = ms->exhausted | (n == cap);
  unsigned long long e_tot = 0, k_lo = 0;
  for (int r = 0; r < R; ++r) { if (r < rank) k_lo += (unsigned long long)vec[r]; e_tot += (unsigned long long)vec[r]; }
  const unsigned long long e_loc = (unsigned long long)vec[rank];
  unsigned long long n_app = 0;
  if (!exhausted) { n_app = e_tot; if (n_app > cap - n) n_app = cap - n; }
  ms->rng0 = ms->rng_state;
  ms->rng_state = eu_rng_jump(J, ms->rng_state, 2 * n_app);
  ms->n0_append = n;
  unsigned long long mine = 0;
  if (n_app > k_lo) { mine = n_app - k_lo; if (mine > e_loc) mine = e_loc; }
  ms->n_append = (unsigned int)mine;
  ms->src_k_lo = (unsigned int)k_lo;
  n += n_app;
  ms->n = n;
  ms->exhausted = exhausted | (n == cap && n_app > 0) | (n == cap);
}
__global__ __launch_bounds__(256) void k_source_place_slab(float2* __restrict__ m, unsigned int* __restrict__ keys, uint8_t* __restrict__ count,
                                                           const unsigned int* __restrict__ elig, const float* __restrict__ draws,
                                                           MarkerState* ms, int X, size_t i0, unsigned long long capacity) {
  const unsigned int n_app = ms->n_append;
  const unsigned long long nl = ms->n_loc, n0 = ms->n0_append + ms->src_k_lo;
  if (nl + n_app > capacity) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicExch(&ms->error, 19); return; }
  for (unsigned int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_app; k += gridDim.x * blockDim.x) {
    const size_t c = i0 + elig[k];
    const int x = (int)(c % (size_t)X), y = (int)(c / (size_t)X);
    const float ry = draws[2 * k], rx = draws[2 * k + 1];
    m[nl + k] = make_float2(EU_H * (x + rx), EU_H * (y + ry));
    keys[nl + k] = (unsigned int)(n0 + k);
    count[c] = (uint8_t)(count[c] + 1);
  }
}
__global__ void k_source_done(MarkerState* ms) { if (!ms->error) ms->n_loc += ms->n_append; }

static int slab_sources(euler_sim* S) {
  SlabScratch* s = S->slab;
  if (s->global_sources == 0) return EULER_OK;   // no '?' cell anywhere: the reference's loop body never runs (else EVERY rank takes part)
  const size_t i0 = (size_t)S->row_lo * S->X, n = (size_t)(S->row_hi - S->row_lo) * S->X;
  LAUNCH(S, KC_SOURCES, k_source_mask_rows, dim3(eu_blocks(n, 256)), dim3(256), S->source, S->count, i0, n, S->cellmask64);
  int rc = eu_ordered_select(S, S->cellmask64, (n + 63) / 64, S->sel_idx, &S->ms->n_events);
  if (rc) return rc;
  hipLaunchKernelGGL(k_source_count_to_vec, dim3(1), dim3(1), 0, S->stream, S->ms, s->vec, s->R, s->rank);
  COMM_CALL(S->comm.allreduce(S->comm.ctx, s->vec, s->R, 0));
  LAUNCH(S, KC_SOURCES, k_source_draws_slab, dim3(1), dim3(1), S->ms, S->rng_jump, s->vec, s->R, s->rank);
  if (S->n_source_cells) {
    eu_source_fill(S);
    LAUNCH(S, KC_SOURCES, k_source_place_slab, dim3(eu_blocks(S->n_source_cells, 256, 2048)), dim3(256), S->markers[S->cur], S->keys[S->cur],
           S->count, S->sel_idx, S->draws, S->ms, S->X, i0, (unsigned long long)S->max_markers);
    hipLaunchKernelGGL(k_source_done, dim3(1), dim3(1), 0, S->stream, S->ms);
  }
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------ the substep
static int slab_advect_markers(euler_sim* S, float dt) {
  SlabScratch* s = S->slab;
  const unsigned long long n = S->n_markers_host;       // exact at substep start
  int rc = eu_marker_advect_a(S, dt, n);                 // every local marker with the incoming dt; candidates -> sel_idx
  if (rc) return rc;
  LAUNCH(S, KC_MARKER_EVENTS, k_pack_events, dim3(4), dim3(256), S->sel_idx, S->ev_theta, S->ev_delta, S->keys[S->cur], S->ms,
         s->xg + (size_t)s->rank * s->blk, s->ev_cap);
  COMM_CALL(S->bulk.allgather(S->bulk.ctx, s->xg, s->ag_off.data(), s->ag_cnt.data()));
  LAUNCH(S, KC_MARKER_EVENTS, k_event_chain, dim3(1), dim3(1024), s->xg, s->R, s->blk, s->sortbuf, s->ev_th, s->ev_de, S->act_idx, S->act_dt, S->ms, dt, s->ev_cap);
  if ((rc = eu_marker_advect_b(S, n, S->keys[S->cur]))) return rc;
  HIPCHK(hipMemcpyAsync(S->keys[S->cur ^ 1], S->keys[S->cur], (size_t)n * sizeof(unsigned int), hipMemcpyDeviceToDevice, S->stream));
  S->cur ^= 1;
  // hand the markers that left the own rows to the neighbours
  const bool has_lo = s->rank > 0, has_hi = s->rank + 1 < s->R;
  HIPCHK(hipMemsetAsync(s->send_lo, 0, 8, S->stream));      // the migrant counters (the buffers also carry the ghost rows)
  HIPCHK(hipMemsetAsync(s->send_hi, 0, 8, S->stream));
  LAUNCH(S, KC_MARKER_COMPACT, k_migrate_out, dim3(eu_blocks((size_t)n, 256, 4096)), dim3(256), S->markers[S->cur], S->keys[S->cur], S->ms,
         S->row_lo, S->row_hi, reinterpret_cast<char*>(s->send_lo), reinterpret_cast<char*>(s->send_hi), (unsigned int)s->mig_cap);
  COMM_CALL(S->bulk.halo(S->bulk.ctx, s->send_lo, s->send_hi, s->recv_lo, s->recv_hi, (int32_t)s->buf_doubles));
  LAUNCH(S, KC_MARKER_COMPACT, k_migrate_in, dim3(64), dim3(256), S->markers[S->cur], S->keys[S->cur], S->ms, reinterpret_cast<const char*>(s->recv_lo),
         reinterpret_cast<const char*>(s->recv_hi), (int)has_lo, (int)has_hi, (unsigned long long)S->max_markers);
  hipLaunchKernelGGL(k_migrate_done, dim3(1), dim3(1), 0, S->stream, S->ms, reinterpret_cast<const char*>(s->recv_lo), reinterpret_cast<const char*>(s->recv_hi),
                     (int)has_lo, (int)has_hi);
  return EULER_OK;
}

int eu_slab_substep(euler_sim* S, float dt) {
  if (!S->has_comm) { eu_set_error("row-slab handle without a communicator"); return EULER_ESTATE; }
  SlabScratch* s = S->slab;
  int rc;
  // advect_markers + migration, refresh_marker_counts, update_fluid_sources (main.c:855-864)
  if ((rc = slab_advect_markers(S, dt))) return rc;
  const unsigned long long n_upper = S->n_markers_host + 2 * s->mig_cap;      // local markers after migration: an upper bound for the launches
  if ((rc = slab_refresh(S, n_upper < S->max_markers ? n_upper : S->max_markers))) return rc;
  // --rainbow: extrapolate(g_r / g / b, P) sits between the refresh and the sources (main.c:859-864).  It reads the count grids' ghost
  // rows of THIS refresh, so with the dye those travel first (and once more behind the sources, which add to the counts)
  if (S->dye[0]) {
    if ((rc = exchange_counts(S))) return rc;
    if ((rc = eu_launch_dye_extrapolate(S))) return rc;
  }
  if ((rc = slab_sources(S))) return rc;
  if ((rc = eu_launch_dye_sources(S))) return rc;
  if ((rc = exchange_counts(S))) return rc;
  if ((rc = eu_slab_exchange_dye(S))) return rc;
  // extrapolate, zero_bounds (main.c:865-868): u, v of the own rows; then their ghost rows
  if ((rc = eu_launch_extrapolate(S))) return rc;
  if ((rc = exchange_uv(S))) return rc;
  // --rainbow: advect_p reads g_u, g_v before anything overwrites them (main.c:871-882); the copied-back rows' ghosts follow
  if (S->dye[0]) {
    if ((rc = eu_launch_dye_advect(S, dt))) return rc;
    if ((rc = eu_slab_exchange_dye(S))) return rc;
  }
  // advect_u / advect_v / body forces / zero_bounds (main.c:871-889) -> utmp, vtmp of the own rows
  if ((rc = eu_launch_advect_velocity(S, dt))) return rc;
  if (S->cfg.viscosity > 0.f) {      // the diffusion extension reads the advected velocities one row below and above the own rows
    const GhostField f[2] = {{S->utmp, 4, 1, 1}, {S->vtmp, 4, 1, 1}};
    if ((rc = exchange_rows(S, f, 2))) return rc;
    if ((rc = eu_launch_diffuse(S, dt))) return rc;
  }
  if ((rc = exchange_vtmp(S))) return rc;
  // project (main.c:893): distributed PCG over the band slabs; p's ghost row inside; u, v of the own rows
  if ((rc = eu_launch_project(S, dt))) return rc;
  return exchange_uv(S);
}

// euler_stage on a row-slab handle (collective): ONE stage of the substep over this rank's rows together with the exchanges that belong to it - the slices of
// eu_slab_substep above, in its order, so that the stages called 0 .. 5 with one dt ARE a substep.  Without the dye (its steps interleave with the stages).
int eu_slab_stage(euler_sim* S, int stage, float dt) {
  if (!S->has_comm) { eu_set_error("row-slab handle without a communicator"); return EULER_ESTATE; }
  if (S->dye[0]) { eu_set_error("euler_stage on a row-slab handle: not with euler_config.rainbow (the dye's steps sit between the stages)"); return EULER_ESTATE; }
  SlabScratch* s = S->slab;
  int rc;
  switch (stage) {
    case EULER_STAGE_ADVECT_MARKERS: return slab_advect_markers(S, dt);
    case EULER_STAGE_REFRESH_COUNTS: {
      const unsigned long long n_upper = S->n_markers_host + 2 * s->mig_cap;
      return slab_refresh(S, n_upper < S->max_markers ? n_upper : S->max_markers);
    }
    case EULER_STAGE_SOURCES: if ((rc = slab_sources(S))) return rc; return exchange_counts(S);
    case EULER_STAGE_EXTRAPOLATE: if ((rc = eu_launch_extrapolate(S))) return rc; return exchange_uv(S);
    case EULER_STAGE_ADVECT_VELOCITY:
      if ((rc = eu_launch_advect_velocity(S, dt))) return rc;
      if (S->cfg.viscosity > 0.f) {
        const GhostField f[2] = {{S->utmp, 4, 1, 1}, {S->vtmp, 4, 1, 1}};
        if ((rc = exchange_rows(S, f, 2))) return rc;
        if ((rc = eu_launch_diffuse(S, dt))) return rc;
      }
      return exchange_vtmp(S);
    case EULER_STAGE_PROJECT: if ((rc = eu_launch_project(S, dt))) return rc; return exchange_uv(S);
    default: eu_set_error("unknown stage %d", stage); return EULER_EINVAL;
  }
}

__global__ void k_set_vec0(double* v, double x) { v[0] = x; }
// after a snapshot has been loaded into a slab handle (snapshot.hip): what a scenario load agrees on collectively
int eu_slab_after_restore(euler_sim* S) {
  SlabScratch* s = S->slab;
  if (!S->has_comm) { eu_set_error("row-slab handle: install the communicator before loading a snapshot"); return EULER_ESTATE; }
  hipLaunchKernelGGL(k_set_vec0, dim3(1), dim3(1), 0, S->stream, s->vec, (double)S->n_source_cells);
  COMM_CALL(S->comm.allreduce(S->comm.ctx, s->vec, 1, 0));
  double tot = 0.0;
  HIPCHK(hipMemcpyAsync(&tot, s->vec, 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  s->global_sources = (unsigned long long)tot;
  return EULER_OK;
}

// sim_init's refresh_marker_counts (main.c:268) + the ghost rows the first substep reads
int eu_slab_after_load(euler_sim* S) {
  SlabScratch* s = S->slab;
  hipLaunchKernelGGL(k_set_vec0, dim3(1), dim3(1), 0, S->stream, s->vec, (double)S->n_source_cells);
  COMM_CALL(S->comm.allreduce(S->comm.ctx, s->vec, 1, 0));
  double tot = 0.0;
  HIPCHK(hipMemcpyAsync(&tot, s->vec, 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  s->global_sources = (unsigned long long)tot;
  int rc = slab_refresh(S, S->n_markers_host);
  if (rc) return rc;
  return exchange_counts(S);
}
