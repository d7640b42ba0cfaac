// k_pcg.hip — the pressure solve: preconditioned conjugate gradient on the masked 5-point
// Laplacian (reference project(), main.c:709-767, and its kernels main.c:580-702).
//
// HBM layout.  Every solver array (b, p, r, z, s, q, precon, cell mask) is private to the solver,
// so it is stored BAND-SKEWED rather than row-major (struct SkewGeom, euler_dev.h):
//     element (row y, column x)  ->  band b = y / 64, lane l = y % 64, record t = x + l
//                                    index  = (b * TS + (t & ~1)) * 64 + 2 * l + (t & 1)
// (T = X + 63 rounded up to even records per band, band stride TS = roundup32(T) + 64 records).
// A record (the 64 elements of one t) is exactly what one wave touches in one step of the IC(0)
// wavefront sweeps (lane l at column t - l).  Records are stored in PAIRS: a lane's elements of
// records 2P and 2P+1 are adjacent, so one 16-byte access per lane serves two steps of a sweep
// (1 KB per wave, perfectly coalesced, no LDS transposition; a lone wave pays per memory
// instruction, not per byte).  The backward sweep walks the same pairs downwards.  The 5-point
// neighbours are the pair partner / the facing element of the adjacent pair (left, right) and the
// same of the neighbouring lane (down, up): still coalesced; band-crossing neighbours of lane 0 /
// lane 63 go through the index function.  Padding entries (t - l outside [0,X), rows >= Y) carry
// mask 0 for ever: every kernel treats them as non-fluid cells, which removes all edge predicates.
//
// Device-resident control: alpha, beta, sigma, the residual norm, the iteration count and the
// `done` flag live in PcgScalars in HBM; every kernel reads them first and returns at once after
// convergence, so the host enqueues iterations without a round trip and polls `done` every few
// iterations.
//
// Bit-exactness: every element-wise kernel evaluates the reference's expression in the
// reference's association order (-ffp-contract=off); the IC(0) sweeps have no reduction, so any
// dependency-respecting schedule gives the sequential sweep's bits; dot() is either replayed in
// the reference's row-major order (EULER_DOT_SEQUENTIAL) or reduced in a fixed tree.
#include "euler_dev.h"
#include "k_mg.h"

#include <stdlib.h>
#include <cmath>
#include <type_traits>
#include <vector>

#define RED_THREADS 1024
typedef double sw_d2 __attribute__((ext_vector_type(2)));   // a lane's pair of records (16-byte accesses)

__device__ __forceinline__ bool pcg_idle(const PcgScalars* sc) { return sc->done || !sc->nonzero; }

#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138

// lane l <- neighbouring lane's v; the lane without a source (0 for shr, 63 for shl) receives `edge`
template <int CTRL>
__device__ __forceinline__ double wave_shift_inject(double v, double edge) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------
// scalar epilogues of the reductions
enum { FIN_SIGMA_INIT = 0, FIN_ALPHA, FIN_RNORM, FIN_BETA, FIN_STORE_ONLY, FIN_TO_COMM };
#define FIN_VIA_P2P 0x100   // multi-rank with mailboxes: the last block all-reduces its total peer to peer, then applies the epilogue

__device__ __forceinline__ void pcg_scalar_step(PcgScalars* sc, int op, double v) {
  switch (op) {
    case FIN_SIGMA_INIT: sc->sigma = v; break;                                        // main.c:748
    case FIN_ALPHA: sc->zs = v; sc->alpha_prev = sc->alpha; sc->alpha = sc->sigma / v; sc->alpha_hist[sc->iters & 7] = sc->alpha; sc->iters += 1; break;     // main.c:750-752
    case FIN_RNORM: sc->rnorm = v; if (v <= sc->tol) sc->done = 1; break;             // main.c:756
    case FIN_BETA: sc->sigma_new = v; sc->beta = v / sc->sigma; sc->sigma = v; break; // main.c:762-765
    case FIN_TO_COMM: sc->comm_val = v; if (sc->comm_slot) *sc->comm_slot = v; break;   // multi-rank: the epilogue runs after the exchange
    default: sc->sigma_new = v; break;
  }
}

// fixed-shape block reductions of an NT-thread block; result valid in thread 0
template <int NT = RED_THREADS>
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double sw[NT / 64];
  v = eu_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int k = 0; k < NT / 64; ++k) t += sw[k];
  return t;
}
template <int NT = RED_THREADS>
__device__ __forceinline__ double block_max(double v) {
  __shared__ double sm[NT / 64];
  v = eu_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int k = 0; k < NT / 64; ++k) t = sm[k] > t ? sm[k] : t;
  return t;
}

// "Last block reduces": each block publishes its partial (8-byte agent-scope atomic store =
// write-through), drains, takes a ticket; the block that draws the last ticket folds ALL partials
// in index order (so the result does not depend on arrival order: deterministic) and applies the
// scalar epilogue.  Saves one launch + one kernel boundary per reduction (3 per PCG iteration).
// Hand-off form: 8-byte agent atomics on both sides (MI355X_MICROARCH "valid forms").
template <bool IS_MAX, int NT = RED_THREADS>
__device__ __forceinline__ void block_finish(double v_block, double* partial, unsigned int* counter, PcgScalars* sc, int op) {
  __shared__ int am_last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&partial[blockIdx.x]),
                       (unsigned long long)__double_as_longlong(v_block), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    am_last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!am_last) return;
  double v = 0.0;
  for (unsigned int i = threadIdx.x; i < gridDim.x; i += NT) {
    const double w = __longlong_as_double((long long)__hip_atomic_load(
        reinterpret_cast<unsigned long long*>(&partial[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (IS_MAX) v = w > v ? w : v; else v += w;
  }
  v = IS_MAX ? block_max<NT>(v) : block_sum<NT>(v);
  if (op & FIN_VIA_P2P) { v = p2p_allreduce_block<IS_MAX>(sc, v); op &= 0xff; }   // op is uniform: every thread of this block is here
  if (threadIdx.x == 0) {
    pcg_scalar_step(sc, op, v);
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next reduction
  }
}

// second stage: one workgroup folds the per-block partials in a fixed order
template <bool IS_MAX>
__global__ __launch_bounds__(RED_THREADS) void k_reduce_final(const double* __restrict__ partial, int n, PcgScalars* sc,
                                                              int op, int force) {
  if (!force && pcg_idle(sc)) return;
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += RED_THREADS) {
    const double w = partial[i];
    if (IS_MAX) v = w > v ? w : v; else v += w;
  }
  v = IS_MAX ? block_max(v) : block_sum(v);
  if (threadIdx.x == 0) pcg_scalar_step(sc, op, v);
}

// What a neighbouring RANK reads across a slab boundary (k_search_apply<true>: the edge rows of z and of the search
// direction) is stored once more WRITE-THROUGH at system scope by the kernel that has it in registers anyway, and every
// thread drains its stores before the block joins the reduction whose all-reduce releases the readers: the remote loads
// then do not depend on what a kernel boundary flushes.  `edges` bit 0 / 1: this slab has a neighbour below / above.
__device__ __forceinline__ void st_system(const double* p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ bool slab_edge_row(size_t e, int l, int TS, int nb_local, int edges) {
  if (l != 0 && l != 63) return false;
  const int band = (int)(e / ((size_t)TS * 64));
  return l == 0 ? ((edges & 1) && band == 0) : ((edges & 2) && band == nb_local - 1);
}

// dot(a,b) over fluid cells -> per-block partials (tree mode); layout-agnostic.  EDGES: `a` is z of a slab (see above).
template <bool EDGES>
__global__ __launch_bounds__(RED_THREADS) void k_dot_partial(const double* __restrict__ a, const double* __restrict__ b,
                                                             const uint8_t* __restrict__ mask, size_t S,
                                                             double* __restrict__ partial, PcgScalars* sc, int force,
                                                             unsigned int* counter, int fin_op, int TS, int nb_local, int edges) {
  if (!force && pcg_idle(sc)) return;
  const size_t chunk = (((S + gridDim.x - 1) / gridDim.x) + 1) & ~(size_t)1;   // S is even: whole 16-byte pairs per thread
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < S ? lo + chunk : S;
  double t = 0.0;
  for (size_t i = lo + 2 * (size_t)threadIdx.x; i < hi; i += 2 * RED_THREADS) {
    const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + i);
    if (!((mm | (mm >> 8)) & CM_FLUID)) continue;
    const sw_d2 av = *reinterpret_cast<const sw_d2*>(a + i), bv = *reinterpret_cast<const sw_d2*>(b + i);
    if (mm & CM_FLUID) t += av.x * bv.x;
    if ((mm >> 8) & CM_FLUID) t += av.y * bv.y;
    if (EDGES && slab_edge_row(i, (int)((i & 127) >> 1), TS, nb_local, edges)) {
      if (mm & CM_FLUID) st_system(a + i, av.x);
      if ((mm >> 8) & CM_FLUID) st_system(a + i + 1, av.y);
    }
  }
  if (EDGES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every thread's write-through stores have landed (block_sum syncs)
  t = block_sum(t);
  block_finish<false>(t, partial, counter, sc, fin_op);
}

// dot(a,b) replayed in the reference's ROW-MAJOR order (main.c:629-639): bit-identical; for small grids (and every bit-exact test).
// The running sum is ONE dependent chain of additions - nothing can split it - so the kernel is built around keeping that chain fed: wave 0's first lane
// adds while waves 1..3 gather the NEXT tile's products (a[i] b[i] of fluid cells, +0.0 elsewhere) into the other half of a double buffer, one
// barrier per tile.  Two things shorten the chain itself, both exact: a group of 64 consecutive cells without fluid is skipped (its ballot is 0), and inside a
// group every entry is added, fluid or not - x + (+0.0) = x for every x except -0.0, and the sum is never -0.0 (it starts as +0.0, and +0.0 + -0.0 = +0.0) -
// which makes the inner loop branch-free: 64 LDS reads in flight, 64 dependent v_add_f64.  Round 3's form (a branch and an LDS round trip per cell, the
// gather not overlapped) took ~1.5 ms per dot at 384 x 448 and was most of the GPU suite's wall time.
#define SEQ_TILE 4096
__global__ __launch_bounds__(256) void k_dot_sequential(const double* __restrict__ a, const double* __restrict__ b,
                                                        const uint8_t* __restrict__ mask, SkewGeom g, PcgScalars* sc,
                                                        int op, int force) {
  if (!force && pcg_idle(sc)) return;
  __shared__ double prod[2][SEQ_TILE];
  __shared__ unsigned long long any[2][SEQ_TILE / 64];
  const size_t C = (size_t)g.X * g.Y;
  const int ntiles = (int)((C + SEQ_TILE - 1) / SEQ_TILE);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double total = 0.0;   // `double total = 0.f`
  for (int tile = -1; tile < ntiles; ++tile) {
    if (wave > 0 && tile + 1 < ntiles) {            // gather tile + 1
      const int buf = (tile + 1) & 1;
      const size_t base = (size_t)(tile + 1) * SEQ_TILE;
      for (int grp = wave - 1; grp < SEQ_TILE / 64; grp += 3) {
        const size_t c = base + (size_t)grp * 64 + lane;
        bool f = false;
        double pr = 0.0;
        if (c < C) {
          const size_t i = skew_index(g, (int)(c % g.X), (int)(c / g.X));
          f = (mask[i] & CM_FLUID) != 0;
          if (f) pr = a[i] * b[i];
        }
        prod[buf][grp * 64 + lane] = pr;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) any[buf][grp] = bal;
      }
    }
    if (threadIdx.x == 0 && tile >= 0) {            // the chain over tile
      const int buf = tile & 1;
      for (int grp = 0; grp < SEQ_TILE / 64; ++grp) {
        if (any[buf][grp] == 0ull) continue;
        const double* pp = &prod[buf][grp * 64];
        double v[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) v[k] = pp[k];
#pragma unroll
        for (int k = 0; k < 64; ++k) total += v[k];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) pcg_scalar_step(sc, op, total);
}

// ------------------------------------------------------------------------------------------
// apply_a (main.c:679-691): z = A s on fluid cells (other entries of z stay 0).
// In tree mode the block also leaves its partial of dot(z,s).
__global__ __launch_bounds__(RED_THREADS) void k_apply_a(const double* __restrict__ s, double* __restrict__ z,
                                                         const uint8_t* __restrict__ mask, SkewGeom g,
                                                         double* __restrict__ partial, PcgScalars* sc, int force,
                                                         unsigned int* counter, int fin_op) {
  if (!force && pcg_idle(sc)) return;
  const size_t S = g.S;
  const size_t chunk = (((S + gridDim.x - 1) / gridDim.x) + 127) & ~(size_t)127;   // whole record pairs per block
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < S ? lo + chunk : S;
  double t = 0.0;
  // One thread per PAIR of records of a lane (euler_dev.h): elements i = (t even, l) and i+1 = (t+1, l) come
  // with one 16-byte load and are each other's right / left neighbour.  The others:
  //   even element: left (t-1, l) = i-127, up (t+1, l+1) = i+3, down (t-1, l-1) = i-129
  //   odd element : right (t+2, l) = i+128, up (t+2, l+1) = i+130, down (t, l-1) = i-2
  // (the adjacent band's lane 0 / 63 through the index function when l = 63 / 0).
  for (size_t i = lo + 2 * (size_t)threadIdx.x; i < hi; i += 2 * RED_THREADS) {
    const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + i);
    const unsigned int m0 = mm & 0xff, m1 = mm >> 8;
    if (!((m0 | m1) & CM_FLUID)) continue;
    const sw_d2 c = *reinterpret_cast<const sw_d2*>(s + i);
    const int l = (int)((i & 127) >> 1);
    sw_d2 o = {0.0, 0.0};
    if (m0 & CM_FLUID) {
      double v = (double)(int)(m0 >> CM_DIAG_SHIFT) * c.x;
      v = v - ((m0 & CM_RIGHT) ? c.y : 0.0);
      if (m0 & CM_UP) {
        size_t up = i + 3;
        if (l == 63) { int band, tt, ll; skew_decode(g, i, band, tt, ll); up = skew_index(g, tt - 63, 64 * (band + 1)); }
        v = v - s[up];
      } else {
        v = v - 0.0;
      }
      v = v - ((m0 & CM_LEFT) ? s[i - 127] : 0.0);
      if (m0 & CM_DOWN) {
        size_t dn = i - 129;
        if (l == 0) { int band, tt, ll; skew_decode(g, i, band, tt, ll); dn = skew_index(g, tt, 64 * band - 1); }
        v = v - s[dn];
      } else {
        v = v - 0.0;
      }
      o.x = v;
      t += v * c.x;
    }
    if (m1 & CM_FLUID) {
      double v = (double)(int)(m1 >> CM_DIAG_SHIFT) * c.y;
      v = v - ((m1 & CM_RIGHT) ? s[i + 128] : 0.0);
      if (m1 & CM_UP) {
        size_t up = i + 130;
        if (l == 63) { int band, tt, ll; skew_decode(g, i + 1, band, tt, ll); up = skew_index(g, tt - 63, 64 * (band + 1)); }
        v = v - s[up];
      } else {
        v = v - 0.0;
      }
      v = v - ((m1 & CM_LEFT) ? c.x : 0.0);
      if (m1 & CM_DOWN) {
        size_t dn = i - 2;
        if (l == 0) { int band, tt, ll; skew_decode(g, i + 1, band, tt, ll); dn = skew_index(g, tt, 64 * band - 1); }
        v = v - s[dn];
      } else {
        v = v - 0.0;
      }
      o.y = v;
      t += v * c.y;
    }
    // z of a non-fluid cell stays what it is (0): write the pair only where both are fluid, else element-wise
    if ((m0 & m1) & CM_FLUID) *reinterpret_cast<sw_d2*>(z + i) = o;
    else if (m0 & CM_FLUID) z[i] = o.x;
    else z[i + 1] = o.y;
  }
  t = block_sum(t);
  if (fin_op >= 0) block_finish<false>(t, partial, counter, sc, fin_op);   // fin_op < 0: dot(z,s) is replayed sequentially
}

// update_search (main.c:669-677) of one iteration fused into apply_a (main.c:679-691) of the next:
//     s' = z + beta s   and   out = A s'   in one pass, out going to a scratch array (the forward solve's q, dead here).
// A cell needs s' of its four neighbours, which it recomputes from z and the old s (the same expression its owner
// evaluates: identical bits) - hence s' goes to a SECOND array, or a neighbour could read a half-updated s.
// Saves a launch and 9 bytes per cell and iteration.  Same pair-per-thread structure as k_apply_a.
// Several ranks (row slabs): the cells across a slab boundary belong to the neighbouring rank, and so do their z and s.
// With the neighbours' arrays mapped (comm_p2p.hip) the kernel reads those two values where they live - a handful of
// system-scope loads over xGMI for the lanes on the slab's edge rows - and forms the neighbour's s' with the owner's
// expression.  No ghost-row exchange, no extra launch: every rank's z is final before anyone gets here (the all-reduce
// behind dot(z,r) separates the backward sweeps from this kernel) and nobody overwrites z or the old s before the
// all-reduce at the end of this kernel.
struct SlabNeighbours {
  const double *z_dn, *s_dn, *z_up, *s_up;   // SLAB 1: the arrays of rank-1 / rank+1, offset like the local ones; null = no such rank
  int nb_local;                              // bands of this slab
  // SLAB 2 (the default with several ranks in tile-local mode): the neighbouring slabs' edge rows as COMPACT rows of X doubles
  // indexed by the column - z as it arrived with the iteration's one exchange, s as this rank keeps it up to date ITSELF: the
  // ghost cell's s' = z + beta s is formed here with the owner's expression (identical bits) and stored for the next iteration,
  // so only z ever travels.  null = no such rank.
  const double *zrow_dn, *srow_dn, *zrow_up, *srow_up;
  double *snew_dn, *snew_up;
};
__device__ __forceinline__ double ld_system(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// PMODE (tile-local mode, where no other kernel of the iteration touches p - k_precond_tile does the rest of main.c:753-765):
//   0  the reference's structure (k_update_pr applies p += alpha s)
//   1  tile-local mode, an odd iteration: p is left alone
//   N = 2, 4, 8  an iteration k >= N, k a multiple of N: p = (..(p + alpha_(k-N) s_(k-N)) + ..) + alpha_(k-1) s_(k-1) - the N fmadds of
//      main.c:753 that are due, in their order, hence the reference's bits.  s_(k-1) is this pass's s_old; s_(k-N) sits in the array the
//      pass is about to overwrite with s_k (the search directions turn through a ring of N arrays), read by the thread that overwrites
//      it; the N - 2 in between come from `hist`.  p is read and written every N-th iteration: (2 w + (N - 1) w) / N bytes per cell and
//      iteration - 12 (N = 2, rounds 3-4), 10 (N = 4), 9 (N = 8) instead of 16.
//
// Schedule.  A wave walks a run of SA_RUN consecutive pair-records of one band (lane = row) with a three-deep window of s'
// in registers - the pair before, the pair itself, the pair after - so every element of z and s is loaded ONCE, by one
// 16-byte access per lane, and the four neighbours of a cell come out of the window: left / right are the lane's own
// registers, down / up the neighbouring lane's by a DPP wave shift (the band-skewed layout puts the lower / upper row's
// value of the same column one record earlier / later in the neighbouring lane).  Only lane 0 / lane 63 look outside
// their band (two lanes of one 8-byte load each per element).  Against one thread per pair gathering six neighbours with
// twelve strided 8-byte loads (round 1) that is 11 instead of 19 memory instructions per pair, and the HBM traffic drops
// from 1.28x to the algorithmic bytes.
//
// INTERIOR chunks (EU_CHUNK_INTERIOR in the list entry: every cell fluid, four fluid neighbours, a_diag 4 - most of a deep tank)
// take a second instantiation of the run body with the masks as compile-time constants: no mask loads, no selects.
#define SA_THREADS 256
struct SaPair { sw_d2 z, so; double ez0, es0, ez1, es1; };   // one pair-record of a lane + the out-of-band vertical neighbours of its two elements

// SA_RUN pair-records per wave: 8 (more waves in flight - at 1024^2 runs of 32 would leave 300 waves for 256 CUs - and the granularity
// of the active-chunk list); 16 / 32 remain for experiments (sa_run)
// COARSE (two-level preconditioner, k_coarse.hip): z stands for z + P y wherever s' = z + beta s is formed - y of the cell's coarse
// cell is looked up (a lane's columns of one run cross at most one coarse column boundary) and added first, like the oracle's z += P y.
__device__ __forceinline__ double pick3(int t, int tb1, int tb2, double v0, double v1, double v2) {
  double v = v0;
  v = t >= tb1 ? v1 : v;
  v = t >= tb2 ? v2 : v;
  return v;
}
struct CoarseRef { const double* y; int shift, nx, ny, band0; };   // y[ny][nx] over coarse cells of (1 << shift)^2 grid cells (COARSE 1) / over the multilevel mode's level-0 nodes (COARSE 2); band0: the global index of the arrays' band 0
__device__ __forceinline__ double pick4(int q, double v0, double v1, double v2, double v3) {
  double v = v0;
  v = q >= 1 ? v1 : v;
  v = q >= 2 ? v2 : v;
  v = q >= 3 ? v3 : v;
  return v;
}
struct SaHist { const double* s[6]; };   // PMODE N: the arrays of s_(k-N+1) .. s_(k-2) (N - 2 of them), offset like s_old
// STORE false: A s' is not stored (`out` is ignored) - k_precond_tile<16, true> forms it again from s' instead of reading it back
// COARSE 2 (multilevel preconditioner, k_mg.hip): z + P_0 x_0 with P_0 bilinear from the level-0 nodes - the lane combines its two node rows once per run (four node columns
// cover the run and its window), every cell then interpolates along its row; the expression is mg_interp0's, so a cell gets the same bits whoever forms its s'.
template <int SLAB, int PMODE, int SA_RUN, int COARSE = 0, bool STORE = true>   // SLAB 1: several ranks, the neighbouring slabs' arrays are mapped; 2: their edge rows as compact rows (nbr); 0: nbr is ignored
// (multilevel mode, one GPU, seven of eight passes: four waves per SIMD - 128 registers, one of them spilled - since the lanes' level-0 node values live in LDS: 212 -> 197 us at 8192^2;
// forced onto the 150 registers of the select-chain form the same bound cost 100 bytes of scratch and 288 us)
// (the tile-local mode's pass has 98 registers, four waves; squeezed to 96 for five - 12 bytes of scratch - it takes 180 us instead of 169)
__global__ __launch_bounds__(SA_THREADS, (COARSE == 2 && PMODE == 1 && SLAB == 0) ? 4 : 1) void k_search_apply(const double* __restrict__ s_old, const double* __restrict__ z,
                                                             double* __restrict__ s_new, double* __restrict__ out,
                                                             const uint8_t* __restrict__ mask, SkewGeom g,
                                                             double* __restrict__ partial, PcgScalars* sc, int force,
                                                             unsigned int* counter, int fin_op, SlabNeighbours nbr,
                                                             double* __restrict__ p, double* s_new_base, double* s_old_base,
                                                             const unsigned int* __restrict__ chunk_list,      // SA_RUN == 8 only: the solve's active runs
                                                             CoarseRef cref, SaHist hist) {
  __shared__ double s_cy[COARSE == 2 ? SA_THREADS / 64 : 1][COARSE == 2 ? MG_NI + 1 : 1][64];      // multilevel mode: a lane's node values of the run (own row / the row across the band boundary)
  __shared__ double s_ce[COARSE == 2 ? SA_THREADS / 64 : 1][COARSE == 2 ? MG_NI + 1 : 1][64];
  if (!force && pcg_idle(sc)) return;
  const double beta = sc->beta;
  (void)s_new_base; (void)s_old_base;
  // PMODE N: alpha of the iterations k - N .. k - 1 (k = the iterations counted so far: this launch's own alpha is written by its LAST block)
  constexpr int NPA = PMODE >= 2 ? PMODE : 2;
  double al[NPA] = {};
  if (PMODE >= 2) {
    const int k = sc->iters;
#pragma unroll
    for (int j = 0; j < NPA; ++j) al[j] = sc->alpha_hist[(k - NPA + j) & 7];
  }
  const int lane = threadIdx.x & 63;
  const int TS = g.TS, npairs = TS / 2;
  const int nb_local = (int)(g.S / ((size_t)TS * 64));
  const int cpb = (npairs + SA_RUN - 1) / SA_RUN, total = nb_local * cpb;
  const int n_waves = gridDim.x * (SA_THREADS / 64);
  const bool edge_lane = lane == 0 || lane == 63;
  double t = 0.0;
  // with the list of active chunks (euler_dev.h) a wave never visits an empty run, and a run's masks are loaded with its data
  const bool listed = chunk_list != nullptr;
  const int ntb16 = g.T / 16;
  const int todo = listed ? (int)sc->n_chunks : total;
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;
  for (int i = blockIdx.x * (SA_THREADS / 64) + (threadIdx.x >> 6); i < todo; i += n_waves) {
    int c = i, per = cpb;
    bool interior = false;
    if (listed) { const unsigned int ent = chunk_list[i]; interior = (ent & EU_CHUNK_INTERIOR) != 0; c = (int)(ent & ~EU_CHUNK_INTERIOR); per = ntb16; }
    const int lb = c / per, P0 = (c % per) * SA_RUN, P1 = P0 + SA_RUN < npairs ? P0 + SA_RUN : npairs;
    const size_t bbase = (size_t)lb * TS * 64 + 2 * lane;     // element (band, record 0, lane)
    auto run = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      // the cell masks of the run; a run without fluid is skipped whole
      unsigned int mm[SA_RUN];
      unsigned int any = 0;
#pragma unroll
      for (int j = 0; j < SA_RUN; ++j) {
        if (FULL) mm[j] = CM_INTERIOR | (CM_INTERIOR << 8);
        else mm[j] = P0 + j < P1 ? (unsigned int)*reinterpret_cast<const unsigned short*>(mask + bbase + (size_t)(P0 + j) * 128) : 0u;
        any |= mm[j];
      }
      if (!FULL && !listed && !__ballot(((any | (any >> 8)) & CM_FLUID) != 0)) return;
      // where lane 0 / lane 63 find the row below / above their band (the adjacent band's lane 63 / lane 0), relative to pair 0:
      // even element (record 2P):  below = record 2P + 63 of band - 1, above = record 2P - 63 of band + 1; odd element: + 1
      const bool up_remote = SLAB == 1 && nbr.z_up && lb + 1 == nb_local, dn_remote = SLAB == 1 && nbr.z_dn && lb == 0;
      const double* ez = lane == 0 ? (dn_remote ? nbr.z_dn : z) : (up_remote ? nbr.z_up : z);
      const double* es = lane == 0 ? (dn_remote ? nbr.s_dn : s_old) : (up_remote ? nbr.s_up : s_old);
      const bool remote = lane == 0 ? dn_remote : up_remote;
      const long long e0_base = lane == 0 ? ((long long)(lb - 1) * TS + 62) * 64 + 127 : ((long long)(lb + 1) * TS - 64) * 64 + 1;
      const long long e1_base = lane == 0 ? ((long long)(lb - 1) * TS + 64) * 64 + 126 : ((long long)(lb + 1) * TS - 62) * 64;
      const unsigned int vbit = lane == 0 ? CM_DOWN : CM_UP;
      // SLAB 2: lane 0 of the slab's first band / lane 63 of its last one find the row across the slab boundary in the compact
      // rows, at the cell's column: even element (record 2P) of lane 0 sits in column 2P, of lane 63 in column 2P - 63
      const bool ghost = SLAB == 2 && (lane == 0 ? (nbr.zrow_dn != nullptr && lb == 0) : (nbr.zrow_up != nullptr && lb + 1 == nb_local));
      const double* gz = lane == 0 ? nbr.zrow_dn : nbr.zrow_up;
      const double* gs = lane == 0 ? nbr.srow_dn : nbr.srow_up;
      double* gsn = lane == 0 ? nbr.snew_dn : nbr.snew_up;
      const int gcol = lane == 0 ? 0 : -63;
      // COARSE: this lane's columns in the run (pairs P0 - 1 .. P1: 20 records at most) start in aggregate column Ja, reach Ja + 1 at record
      // tb1 and Ja + 2 at record tb2 (aggregates of 16: three columns; of 64 and more: two); the lane's row decides the aggregate row
      double cy0 = 0.0, cy1 = 0.0, cy2 = 0.0, ce0 = 0.0, ce1 = 0.0, ce2 = 0.0;
      double (*cyv)[64] = s_cy[COARSE == 2 ? threadIdx.x >> 6 : 0];
      double (*cev)[64] = s_ce[COARSE == 2 ? threadIdx.x >> 6 : 0];
      int ctb1 = 0x7fffffff, ctb2 = 0x7fffffff, cJb = 0;
      (void)cJb;
      if (COARSE == 2) {
        // the lane's columns in the run and its window (records 2 (P0 - 1) .. 2 P1 + 1: 20 at most) lie between the node columns Jb .. Jb + MG_NI (spacing MG_G0); the column of
        // record t lies between Jb + m and Jb + m + 1 from record ctb[m] on.  Nodes beyond the grid repeat the outermost one (constant there)
        cJb = (2 * (P0 - 1) - lane - MG_G0 / 2) >> MG_LOG;
        ctb1 = MG_G0 * (cJb + 1) + lane + MG_G0 / 2;
        const int row = (cref.band0 + lb) * 64 + lane;
        const int hi_ = cref.nx - 1;
        int i0, i1;
        double fy;
        mg_cell_w(row, cref.ny, i0, i1, fy);
        const double* y0 = cref.y + (size_t)i0 * cref.nx;
        const double* y1 = cref.y + (size_t)i1 * cref.nx;
#pragma unroll
        for (int q = 0; q <= MG_NI; ++q) { const int c = cJb + q < 0 ? 0 : (cJb + q < hi_ ? cJb + q : hi_); cyv[q][lane] = mg_rows(y0[c], y1[c], fy); }
#pragma unroll
        for (int q = 0; q <= MG_NI; ++q) cev[q][lane] = 0.0;
        if (edge_lane) {      // the row across the band boundary
          const int re = row + (lane == 0 ? -1 : 1);
          if (re >= 0 && re < MG_G0 * cref.ny) {
            mg_cell_w(re, cref.ny, i0, i1, fy);
            y0 = cref.y + (size_t)i0 * cref.nx; y1 = cref.y + (size_t)i1 * cref.nx;
#pragma unroll
            for (int q = 0; q <= MG_NI; ++q) { const int c = cJb + q < 0 ? 0 : (cJb + q < hi_ ? cJb + q : hi_); cev[q][lane] = mg_rows(y0[c], y1[c], fy); }
          }
        }
      }
      // (the node values sit in LDS, [node column][lane]: a record picks its interval by arithmetic and reads two of them - conflict-free, and twenty registers fewer than arrays
      // with a chain of selects per record)
      auto p0y = [&](int t, double (*av)[64]) __attribute__((always_inline)) {      // (P_0 x_0) at the lane's column of record t, from a row's node values
        const double f = (double)((t - lane - MG_G0 / 2) & (MG_G0 - 1)) * (1.0 / MG_G0);
        int m = t >= ctb1 ? ((t - ctb1) >> MG_LOG) + 1 : 0;
        m = m < MG_NI - 1 ? m : MG_NI - 1;
        return mg_lerp_x(av[m][lane], av[m + 1][lane], f);
      };
      if (COARSE == 1) {
        const int xa = 2 * (P0 - 1) - lane, Ja = (xa > 0 ? xa : 0) >> cref.shift;
        const int row = (cref.band0 + lb) * 64 + lane;
        const size_t I = (size_t)(row >> cref.shift) * cref.nx;
        ctb1 = ((Ja + 1) << cref.shift) + lane;
        ctb2 = ((Ja + 2) << cref.shift) + lane;
        cy0 = Ja < cref.nx ? cref.y[I + Ja] : 0.0;
        cy1 = Ja + 1 < cref.nx ? cref.y[I + Ja + 1] : 0.0;
        cy2 = Ja + 2 < cref.nx ? cref.y[I + Ja + 2] : 0.0;
        if (edge_lane) {      // the row across the band boundary may belong to the neighbouring aggregate row (same columns)
          const int re = row + (lane == 0 ? -1 : 1);
          const bool inside = re >= 0 && (re >> cref.shift) < cref.ny;
          const size_t Ie = (size_t)((inside ? re : row) >> cref.shift) * cref.nx;
          ce0 = (inside && Ja < cref.nx) ? cref.y[Ie + Ja] : 0.0;
          ce1 = (inside && Ja + 1 < cref.nx) ? cref.y[Ie + Ja + 1] : 0.0;
          ce2 = (inside && Ja + 2 < cref.nx) ? cref.y[Ie + Ja + 2] : 0.0;
        }
      }
      auto load_pair = [&](int P, SaPair& d, unsigned int m) __attribute__((always_inline)) {
        d.ez0 = d.es0 = d.ez1 = d.es1 = 0.0;
        if (P < 0 || P >= npairs) { d.z = sw_d2{0.0, 0.0}; d.so = sw_d2{0.0, 0.0}; return; }   // outside the band: never a fluid cell's neighbour
        d.z = *reinterpret_cast<const sw_d2*>(z + bbase + (size_t)P * 128);
        d.so = *reinterpret_cast<const sw_d2*>(s_old + bbase + (size_t)P * 128);
        if (edge_lane) {
          if ((m & CM_FLUID) && (m & vbit)) {
            const long long k = e0_base + (long long)P * 128;
            if (SLAB == 2 && ghost) { d.ez0 = gz[2 * P + gcol]; d.es0 = gs[2 * P + gcol]; }
            else if (SLAB == 1 && remote) { d.ez0 = ld_system(ez + k); d.es0 = ld_system(es + k); } else { d.ez0 = ez[k]; d.es0 = es[k]; }
          }
          if (((m >> 8) & CM_FLUID) && ((m >> 8) & vbit)) {
            const long long k = e1_base + (long long)P * 128;
            if (SLAB == 2 && ghost) { d.ez1 = gz[2 * P + 1 + gcol]; d.es1 = gs[2 * P + 1 + gcol]; }
            else if (SLAB == 1 && remote) { d.ez1 = ld_system(ez + k); d.es1 = ld_system(es + k); } else { d.ez1 = ez[k]; d.es1 = es[k]; }
          }
          if (COARSE == 1) {      // (harmless where nothing was loaded: the value is then never selected)
            d.ez0 = d.ez0 + pick3(2 * P, ctb1, ctb2, ce0, ce1, ce2);
            d.ez1 = d.ez1 + pick3(2 * P + 1, ctb1, ctb2, ce0, ce1, ce2);
          }
          if (COARSE == 2) {
            d.ez0 = d.ez0 + p0y(2 * P, cev);
            d.ez1 = d.ez1 + p0y(2 * P + 1, cev);
          }
        }
        if (COARSE == 1) {        // z + P y of the lane's own two cells (records 2P, 2P + 1)
          d.z.x = d.z.x + pick3(2 * P, ctb1, ctb2, cy0, cy1, cy2);
          d.z.y = d.z.y + pick3(2 * P + 1, ctb1, ctb2, cy0, cy1, cy2);
        }
        if (COARSE == 2) {
          d.z.x = d.z.x + p0y(2 * P, cyv);
          d.z.y = d.z.y + p0y(2 * P + 1, cyv);
        }
      };
      auto sprime = [&](const SaPair& d) __attribute__((always_inline)) { return sw_d2{d.z.x + beta * d.so.x, d.z.y + beta * d.so.y}; };   // s' = z + beta s (main.c:674)
      SaPair A, B, Cn;
      load_pair(P0 - 1, A, 0u);
      load_pair(P0, B, mm[0]);
      load_pair(P0 + 1, Cn, SA_RUN > 1 ? mm[1] : 0u);
      double prev_y = sprime(A).y;
      sw_d2 cur = sprime(B);
#pragma unroll
      for (int j = 0; j < SA_RUN; ++j) {
        const int P = P0 + j;
        if (P < P1) {                                       // (wave-uniform)
          SaPair D;
          load_pair(P + 2 <= P1 ? P + 2 : -1, D, j + 2 < SA_RUN ? mm[j + 2] : 0u);      // the pair after the next, in flight while this one computes
          const unsigned int m0 = mm[j] & 0xff, m1 = mm[j] >> 8;
          const sw_d2 nxt = sprime(Cn);
          // the rows below / above: the neighbouring lane's registers (every lane takes part: a lane whose own pair holds no
          // fluid still serves its neighbours); lane 0 / 63 inject what they fetched from the adjacent band
          const double e0 = B.ez0 + beta * B.es0, e1 = B.ez1 + beta * B.es1;
          if (SLAB == 2 && ghost && edge_lane) {            // the ghost cells' s' for the next iteration (the owner forms the same bits)
            if ((mm[j] & CM_FLUID) && (mm[j] & vbit)) gsn[2 * P + gcol] = e0;
            if (((mm[j] >> 8) & CM_FLUID) && ((mm[j] >> 8) & vbit)) gsn[2 * P + 1 + gcol] = e1;
          }
          const double dn0 = wave_shift_inject<DPP_WAVE_SHR1>(prev_y, e0), up0 = wave_shift_inject<DPP_WAVE_SHL1>(cur.y, e0);
          const double dn1 = wave_shift_inject<DPP_WAVE_SHR1>(cur.x, e1), up1 = wave_shift_inject<DPP_WAVE_SHL1>(nxt.x, e1);
          if ((m0 | m1) & CM_FLUID) {
            const size_t i = bbase + (size_t)P * 128;
            sw_d2 cc = B.so, o = {0.0, 0.0};                // cc: the pair's s' (a non-fluid element keeps its old value, +0)
            if (PMODE >= 2) {
              sw_d2 pv = *reinterpret_cast<const sw_d2*>(p + i);
              constexpr int NP = PMODE >= 2 ? PMODE : 2;
              sw_d2 sv[NP];
              sv[0] = *reinterpret_cast<const sw_d2*>(s_new + i);      // s of N iterations ago, about to be overwritten
#pragma unroll
              for (int j = 1; j < NP - 1; ++j) sv[j] = *reinterpret_cast<const sw_d2*>(hist.s[j - 1] + i);
              sv[NP - 1] = B.so;
#pragma unroll
              for (int j = 0; j < NP; ++j) {
                if (m0 & CM_FLUID) pv.x = pv.x + sv[j].x * al[j];
                if (m1 & CM_FLUID) pv.y = pv.y + sv[j].y * al[j];
              }
              *reinterpret_cast<sw_d2*>(p + i) = pv;        // a non-fluid partner is written back unchanged
            }
            if (m0 & CM_FLUID) cc.x = cur.x;
            if (m1 & CM_FLUID) cc.y = cur.y;
            if (m0 & CM_FLUID) {                            // apply_a (main.c:679-691): diag, right, up, left, down
              double v = (double)(int)(m0 >> CM_DIAG_SHIFT) * cc.x;
              v = v - ((m0 & CM_RIGHT) ? cc.y : 0.0);
              v = v - ((m0 & CM_UP) ? up0 : 0.0);
              v = v - ((m0 & CM_LEFT) ? prev_y : 0.0);
              v = v - ((m0 & CM_DOWN) ? dn0 : 0.0);
              o.x = v;
              t += v * cc.x;
            }
            if (m1 & CM_FLUID) {
              double v = (double)(int)(m1 >> CM_DIAG_SHIFT) * cc.y;
              v = v - ((m1 & CM_RIGHT) ? nxt.x : 0.0);
              v = v - ((m1 & CM_UP) ? up1 : 0.0);
              v = v - ((m1 & CM_LEFT) ? cc.x : 0.0);
              v = v - ((m1 & CM_DOWN) ? dn1 : 0.0);
              o.y = v;
              t += v * cc.y;
            }
            if ((m0 & m1) & CM_FLUID) *reinterpret_cast<sw_d2*>(s_new + i) = cc;
            else if (m0 & CM_FLUID) s_new[i] = cc.x;
            else s_new[i + 1] = cc.y;
            if (STORE) {
              if ((m0 & m1) & CM_FLUID) *reinterpret_cast<sw_d2*>(out + i) = o;
              else if (m0 & CM_FLUID) out[i] = o.x;
              else out[i + 1] = o.y;
            }
            if (SLAB == 1 && slab_edge_row(i, lane, TS, nb_local, (nbr.z_dn ? 1 : 0) | (nbr.z_up ? 2 : 0))) {   // the rows the neighbours will read
              if (m0 & CM_FLUID) st_system(s_new + i, cc.x);
              if (m1 & CM_FLUID) st_system(s_new + i + 1, cc.y);
            }
          }
          prev_y = cur.y; cur = nxt;
          B = Cn; Cn = D;
        }
      }
    };
    if (SA_RUN == 8 && interior) run(yes_t()); else run(no_t());
  }
  if (SLAB == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // landed before this block joins the all-reduce (block_sum syncs)
  t = block_sum<SA_THREADS>(t);
  if (fin_op >= 0) block_finish<false, SA_THREADS>(t, partial, counter, sc, fin_op);   // fin_op < 0: dot(out, s') is replayed sequentially
}

// p += alpha s ; r -= alpha z (fmadd x2, main.c:753-754) ; per-block max |r| (inf_norm, main.c:654-667)
__global__ __launch_bounds__(RED_THREADS) void k_update_pr(double* __restrict__ p, double* __restrict__ r,
                                                           const double* __restrict__ s, const double* __restrict__ z,
                                                           const uint8_t* __restrict__ mask, size_t S,
                                                           double* __restrict__ partial, PcgScalars* sc, int force,
                                                           double alpha_arg, unsigned int* counter, int fin_op) {
  if (!force && pcg_idle(sc)) return;
  const double alpha = force ? alpha_arg : sc->alpha;
  const double nalpha = -alpha;
  const size_t chunk = (((S + gridDim.x - 1) / gridDim.x) + 1) & ~(size_t)1;   // whole 16-byte pairs per thread
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < S ? lo + chunk : S;
  double mx = 0.0;
  for (size_t i = lo + 2 * (size_t)threadIdx.x; i < hi; i += 2 * RED_THREADS) {
    const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + i);
    const bool f0 = (mm & CM_FLUID) != 0, f1 = ((mm >> 8) & CM_FLUID) != 0;
    if (!(f0 | f1)) continue;
    const sw_d2 sv = *reinterpret_cast<const sw_d2*>(s + i), zv = *reinterpret_cast<const sw_d2*>(z + i);
    sw_d2 pv = *reinterpret_cast<const sw_d2*>(p + i), rv = *reinterpret_cast<const sw_d2*>(r + i);
    if (f0) { pv.x = pv.x + sv.x * alpha; rv.x = rv.x + zv.x * nalpha; const double a = fabs(rv.x); if (a > mx) mx = a; }
    if (f1) { pv.y = pv.y + sv.y * alpha; rv.y = rv.y + zv.y * nalpha; const double a = fabs(rv.y); if (a > mx) mx = a; }
    *reinterpret_cast<sw_d2*>(p + i) = pv;      // a non-fluid partner is written back unchanged
    *reinterpret_cast<sw_d2*>(r + i) = rv;
  }
  mx = block_max(mx);
  if (fin_op >= 0) block_finish<true>(mx, partial, counter, sc, fin_op);
}

// max |r| only (EULER_OP_INF_NORM_R)
__global__ __launch_bounds__(RED_THREADS) void k_inf_norm(const double* __restrict__ r, const uint8_t* __restrict__ mask,
                                                          size_t S, double* __restrict__ partial) {
  const size_t chunk = (S + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < S ? lo + chunk : S;
  double mx = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += RED_THREADS)
    if (mask[i] & CM_FLUID) { const double a = fabs(r[i]); if (a > mx) mx = a; }
  mx = block_max(mx);
  if (threadIdx.x == 0) partial[blockIdx.x] = mx;
}

// s = z + beta s (update_search, main.c:669-677); with COPY: s = z (the memcpy at main.c:746)
template <bool COPY>
__global__ __launch_bounds__(256) void k_update_search(double* __restrict__ s, const double* __restrict__ z,
                                                       const uint8_t* __restrict__ mask, size_t S, const PcgScalars* sc,
                                                       int force, double beta_arg) {
  if (!force && pcg_idle(sc)) return;
  const double beta = force ? beta_arg : sc->beta;
  for (size_t i = 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < S; i += 2 * (size_t)gridDim.x * blockDim.x) {
    const sw_d2 zv = *reinterpret_cast<const sw_d2*>(z + i);
    if (COPY) { *reinterpret_cast<sw_d2*>(s + i) = zv; continue; }
    const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + i);
    const bool f0 = (mm & CM_FLUID) != 0, f1 = ((mm >> 8) & CM_FLUID) != 0;
    if (!(f0 | f1)) continue;
    sw_d2 sv = *reinterpret_cast<const sw_d2*>(s + i);
    if (f0) sv.x = zv.x + beta * sv.x;
    if (f1) sv.y = zv.y + beta * sv.y;
    *reinterpret_cast<sw_d2*>(s + i) = sv;      // a non-fluid partner is written back unchanged
  }
}

// the slab's edge rows of a skewed array once more, write-through at system scope (see st_system): row y0 and / or y1
__global__ __launch_bounds__(256) void k_publish_edge_rows(double* arr, SkewGeom g, int y0, int y1, const PcgScalars* sc) {
  if (pcg_idle(sc)) return;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= g.X) return;
  const int y = blockIdx.y == 0 ? y0 : y1;
  if (y < 0) return;
  double* p = arr + skew_index(g, x, y);
  st_system(p, *p);
}

// Jacobi stand-in preconditioner (not the reference's iterates; roofline comparison only)
__global__ __launch_bounds__(256) void k_jacobi(const double* __restrict__ r, double* __restrict__ z,
                                                const uint8_t* __restrict__ mask, size_t S, const PcgScalars* sc, int force) {
  if (!force && pcg_idle(sc)) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (size_t)gridDim.x * blockDim.x) {
    const uint8_t m = mask[i];
    const int d = m >> CM_DIAG_SHIFT;
    z[i] = (m & CM_FLUID) ? r[i] / (double)(d ? d : 1) : 0.0;
  }
}

// ==========================================================================================
// IC(0): E^-1 factor, forward solve, backward solve (apply_preconditioner, main.c:580-627).
//
// Cell (x,y) depends on its left and lower neighbours (forward) or right and upper (backward):
// a 2-D recurrence with no reduction, so every dependency-respecting schedule reproduces the
// sequential sweep bit for bit.
//
// NOTE on the reference's coefficients: get_a_minus_i(y,x) = get_a_plus_i(y,x-1) = is_fluid(y,x)
// ? -1 : 0 (main.c:561-575) is ALWAYS -1 for the fluid cell being visited, whatever its left or
// lower neighbour is.  Hence (a) the E^-1 recurrence reads the STALE precon[] of neighbours that
// are no longer fluid (precon[] persists, main.c:577, and is only written on fluid cells), and
// (b) the forward solve needs no neighbour mask: q is +0 on non-fluid cells.  The backward solve
// uses get_a_plus_i/j(y,x) = is_fluid of the right/upper neighbour.
enum { SW_FACTOR = 0, SW_FORWARD = 1, SW_BACKWARD = 2 };

struct SweepArgs {
  SkewGeom g;
  const uint8_t* mask;
  const unsigned int* fbits_fwd;   // [nbands][fb_stride][64]: bit j of word (band, g, lane) = fluid flag of the lane's cell in
  const unsigned int* fbits_bwd;   // step 8g+j of the forward / backward sweep (k_pack_fbits); one dword load per block
  int fb_stride;
  double* pre;            // precon: in/out for SW_FACTOR, in otherwise
  const double* in;       // r (forward) / q (backward); unused for factor
  double* out;            // q (forward) / z (backward); unused for factor
  unsigned long long* granules;   // [nbands][gran_stride][2] tagged hand-off of a band's edge row
  int gran_stride;
  // k_sweep_skew<OP, true> (exact coupling over the peer-to-peer mailboxes, comm_p2p.hip): the slab's first band takes
  // its boundary row from the own mailbox (written by the previous slab's last band on ANOTHER GPU while both kernels
  // run), the slab's last band announces into the next slab's mailbox; one row of gran_stride granule pairs each
  const unsigned long long* xg_in;
  unsigned long long* xg_out;
  const int4* ranges;     // per band: active block ranges {fwd B0, fwd B1, bwd B0, bwd B1} (k_band_ranges)
  int band_lo, nb_local;  // this rank's bands [band_lo, band_lo + nb_local)
  int couple;             // 1: the first/last local band is coupled to the neighbouring rank's band
  unsigned int* ticket;
  unsigned int ticket_base;
  unsigned int epoch;
  const PcgScalars* sc;
  int force;
  int* error;
  unsigned long long* timeline;   // [nbands][8] {entry, first block ready, exit, blocks << 32 | stalled blocks, 4 development words} (euler_sweep_timeline)
  int tile_w;                     // k_sweep_simple only: > 0 = tile-local IC(0) with tiles of tile_w records (the cross-check of k_precond_tile)
  // forward sweep, tree-dot mode, one rank: dot(z, r) of the preconditioner application this sweep starts, formed HERE as
  // dot(q, q) - z = L^-T q and q = L^-1 r, so z.r = (L^-T q).(L q) = q.q exactly in real arithmetic (the backward solve applies
  // the transpose of the forward solve's L: same precon, symmetric couplings), a sum of squares with no cancellation.  It
  // replaces a launch that re-read z and r (17 B per cell); EULER_DOT_SEQUENTIAL keeps the reference's own dot(z, r).
  int fin_qq;                     // scalar epilogue (FIN_SIGMA_INIT / FIN_BETA) or -1
  double* qq_partial;             // [bands of the launch]
  unsigned int* qq_counter;
  PcgScalars* sc_w;
};

// one wave per band arrives with its partial; the last one folds all partials in band order (deterministic) and applies the epilogue
__device__ __forceinline__ void sweep_qq_arrive(const SweepArgs& a, int ord, double lane_sum) {
  const int lane = threadIdx.x & 63;
  const double v = eu_wave_sum(lane_sum);
  int last = 0;
  if (lane == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.qq_partial[ord]), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(a.qq_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)a.nb_local - 1;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  double t = 0.0;
  for (int k = lane; k < a.nb_local; k += 64)
    t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&a.qq_partial[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  t = eu_wave_sum(t);
  if (lane == 0) {
    pcg_scalar_step(a.sc_w, a.fin_qq, t);
    __hip_atomic_store(a.qq_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int OP>
__device__ __forceinline__ double sweep_cell(uint8_t m, double in, double pre_here, double own_val, double own_pre,
                                             double nb_val, double nb_pre) {
  // own_* : previous cell of the same row in sweep order (left for forward, right for backward)
  // nb_*  : same column in the previous row in sweep order (below for forward, above for backward)
  if (OP == SW_FACTOR) {
    if (!(m & CM_FLUID)) return pre_here;           // untouched (stale) entry
    const double a = (double)(int)(m >> CM_DIAG_SHIFT);
    const double cl = -1.0 * own_val;               // get_a_minus_i * precon[y][x-1]
    const double cb = -1.0 * nb_val;                // get_a_minus_j * precon[y-1][x]
    double e = a - cl * cl - cb * cb;
    if (e < 0.25 * a) e = (a != 0.0) ? a : 1.0;
    return 1.0 / sqrt(e);
  } else if (OP == SW_FORWARD) {
    if (!(m & CM_FLUID)) return 0.0;
    const double t = in - -1.0 * own_pre * own_val - -1.0 * nb_pre * nb_val;
    return t * pre_here;
  } else {
    if (!(m & CM_FLUID)) return 0.0;
    const double cr = (m & CM_RIGHT) ? -1.0 : 0.0, cu = (m & CM_UP) ? -1.0 : 0.0;
    const double t = in - cr * pre_here * own_val - cu * pre_here * nb_val;
    return t * pre_here;
  }
}

// --- debug / cross-check schedule: one workgroup, one barrier per anti-diagonal ---------------
template <int OP>
__global__ __launch_bounds__(1024) void k_sweep_simple(SweepArgs a) {
  if (!a.force && pcg_idle(a.sc)) return;
  const SkewGeom g = a.g;
  const int X = g.X, Y = g.Y;
  constexpr bool BWD = OP == SW_BACKWARD;
  double* dst = OP == SW_FACTOR ? a.pre : a.out;
  for (int d = 0; d < X + Y - 1; ++d) {
    for (int yl = threadIdx.x; yl < Y; yl += 1024) {
      const int xl = d - yl;
      if (xl < 0 || xl >= X) continue;
      const int x = BWD ? X - 1 - xl : xl, y = BWD ? Y - 1 - yl : yl;
      const size_t i = skew_index(g, x, y);
      const uint8_t m = a.mask[i];
      if (OP == SW_FACTOR && !(m & CM_FLUID)) continue;
      double r = 0.0;
      if (m & CM_FLUID) {   // fluid cells are interior: the neighbours exist
        const size_t io = skew_index(g, BWD ? x + 1 : x - 1, y), in_ = skew_index(g, x, BWD ? y + 1 : y - 1);
        double own_val = OP == SW_FACTOR ? a.pre[io] : dst[io];
        double nb_val = OP == SW_FACTOR ? a.pre[in_] : dst[in_];
        double own_pre = OP == SW_FORWARD ? a.pre[io] : 0.0, nb_pre = OP == SW_FORWARD ? a.pre[in_] : 0.0;
        if (a.tile_w > 0) {   // tile-local IC(0): a cut coupling carries what the wavefront carries into a tile
          const int l = y & 63, t = x + l;
          const bool cut = (BWD ? t + 1 : t) % a.tile_w == 0;
          if (cut) { own_val = 0.0; own_pre = 1.0; }                                   // forward: (-1 * 1) * (+0) = -0.0
          if (cut || l == (BWD ? 63 : 0)) { nb_val = 0.0; nb_pre = 1.0; }
        }
        r = sweep_cell<OP>(m, OP == SW_FACTOR ? 0.0 : a.in[i], a.pre[i], own_val, own_pre, nb_val, nb_pre);
      }
      dst[i] = r;
    }
    __syncthreads();
  }
}

// --- production schedule: one workgroup per 64-row band = a compute wave + two helper waves ---------
// Compute wave.  Lane l owns row 64 b + l.  Forward: records t = 0, 1, ..., lane l is at column t - l,
// the row below arrives from lane l-1 (DPP wave_shr:1), the previous column is the lane's own
// register.  Backward: records T-1, T-2, ..., the row above arrives from lane l+1 (DPP wave_shl:1).
// The unit of work is 8 steps = one hand-off block, fully unrolled; records come in pairs (a lane's elements of
// records 2P, 2P+1 are adjacent), so every stream moves two steps per 16-byte access, and the operands of the
// next two or three blocks are in flight into rotating register sets while a block computes (fluid flags: 8
// steps to a dword).  A lone wave is bound by instruction ISSUE and by the latency of whatever it waits
// for, so the compute wave touches global memory only for its streams and everything about the band
// hand-off lives in the helper waves (own SIMDs, own vmcnt):
//   * the compute wave drops every step's carry row into an LDS ring (two rows per ds_write2st64_b64); the
//     ANNOUNCE wave gathers the edge lane's (63 forward / 0 backward: logical column s-63) values block by
//     block and publishes them to the next band as 16-byte granule pairs {lo, epoch, hi, epoch}
//     (write-through stores), up to 8 groups per store;
//   * the FETCH wave polls the previous band's granules with 4 loads in flight, each covering up to 8 blocks
//     ahead of the compute wave, and parks validated boundary values in a second LDS ring; the compute wave
//     reads a block's 8 values as broadcasts two steps before the previous block ends - they become the `old`
//     operand of the DPP shift, i.e. what the lane without a shift source receives.
// The waves talk through three monotonic LDS counters (blocks computed / boundary blocks deposited /
// groups announced); a wave's LDS operations execute in order, so "data, then counter" needs no fence.
// Bands take their order from a ticket, so a band only ever waits on a band that is already
// running: no residency assumption, no deadlock; every spin is bounded (sticky error -> ETIMEOUT).
// One band per workgroup (= per CU) on purpose: the CU's vector-memory path is shared - a second compute wave
// on the CU costs each +18 % per step, four run 2.2x slower (tools/micro/step_bench2).
// Measured and rejected: staging the compute wave's streams through LDS as well (a load wave feeding an
// operand ring by LDS-DMA, a store wave draining a result ring; the compute wave without any global access).
// A stand-alone model of the step promised 27 ns instead of 40; the real kernel, with its per-block
// bookkeeping and five waves on the CU, ran 40 ns/step for a lone band and 50-55 ns with neighbours, and the
// load wave could not keep the ring full from HBM at 8192^2 (1.6x slower sweeps).  Bit-exact, but not faster.
// Measured and rejected: placing consecutive bands on one XCD (every 8th workgroup) with write-through or
// with plain granule stores - the hand-off lag does not move (4.6-4.8 us per band either way).
// Measured and rejected at 8192^2 (per-step time there is ~1.25x / 1.4x that of an L2-resident grid):
// a fourth wave touching the coming records' cache lines 10 blocks ahead (L2 prefetch: no gain forward,
// 1.2x slower backward - the touches cross the same per-CU memory path), and padding the band stride
// against HBM channel aliasing (no effect).
// Measured and rejected (round 1, last experiment): polling through the SCALAR memory path.  tools/micro/poll_bench: one hop
// costs 480-640 ns with vector sc1 polls (more behind this kernel's deep prefetch queue: ~1.1 us), 450 ns with
// `s_load_dwordx4 glc` polls whatever the CU's vector traffic does, same- or cross-XCD (sc1 stores; sc0 loads and plain
// cross-XCD stores read stale).  But a scalar round trip carries at most ~256 B (64 SGPRs) = the granules of two blocks and
// must be retired whole (out-of-order returns): a single scalar poller delivered a block every ~400 ns, the compute wave
// needs one every 210-300 ns, and the sweeps ran 1.6x slower (72 -> 123 us).  What is left to try: two scalar pollers on
// alternate blocks (a fourth wave), or tag-free compact rows behind a drained progress word.
// Records t >= T of a band and the 32 records in front of each array are dead padding (mask 0):
// the loop runs whole groups of 3 blocks and prefetches unconditionally.
#define SW_BLK 8
#define SW_RING 64            // carry rows kept in LDS (8 blocks)
#define SW_BND_RING 16        // boundary blocks kept in LDS
#define SW_SPIN_LIMIT (1u << 22)
#ifndef SW_TRACE_HANDOFF
#define SW_TRACE_HANDOFF 0   // development build: time stamps of one hand-off (column block 40) in the timeline words 4..7
#endif
#define SW_TRACE_CB 40

struct SweepShared {
  double pub[SW_RING][64];            // carry rows of the last SW_RING steps (ring slot = step & 63)
  double bnd[SW_BND_RING][SW_BLK];    // boundary values, ring slot = (block - B0) & 15
  unsigned int dep_done, pub_done;    // helper -> compute: boundary blocks deposited, groups announced (one 8-byte read)
  unsigned int comp_done;             // compute -> helper: blocks computed
  unsigned int abort;
  int ord;
};
__device__ __forceinline__ unsigned int lds_get(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_put(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define SW_COMPILER_FENCE() asm volatile("" ::: "memory")

template <int OP, bool XG = false>
__global__ __launch_bounds__(192) void k_sweep_skew(SweepArgs a) {
  __shared__ SweepShared sh;
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;                  // 0 compute, 1 announce, 2 fetch boundaries
  if (threadIdx.x == 0) {
    sh.ord = (int)(atomicAdd(a.ticket, 1u) - a.ticket_base);        // position in the band pipeline
    sh.dep_done = 0; sh.pub_done = 0; sh.comp_done = 0; sh.abort = 0;
  }
  __syncthreads();
  const int ord = __builtin_amdgcn_readfirstlane(sh.ord);
  if (!a.force && pcg_idle(a.sc)) return;
  const unsigned long long t_entry = wall_clock64();
  constexpr bool BWD = OP == SW_BACKWARD;
  constexpr int CTRL = BWD ? DPP_WAVE_SHL1 : DPP_WAVE_SHR1;
  constexpr int EDGE = BWD ? 0 : 63;                  // the lane whose results the next band needs
  const SkewGeom g = a.g;
  const int X = g.T - 63, T = g.T, TS = g.TS, nb = g.nbands;   // X: hand-off columns live in step space, [0, T - 63) (T is even: g.X or g.X + 1)
  // `ord` counts this launch's (= this rank's) bands in sweep order; gord is the position in the
  // global band pipeline, which also names the hand-off rows (forwarded rank to rank when coupled)
  const int band = BWD ? a.band_lo + a.nb_local - 1 - ord : a.band_lo + ord;
  const int gord = BWD ? nb - 1 - band : band;
  const bool has_prev = ord > 0 || (a.couple && gord > 0);            // a band before us in sweep order
  const bool publish = ord + 1 < a.nb_local || (a.couple && gord + 1 < nb);
  unsigned long long* gr_out = a.granules + (size_t)gord * a.gran_stride * 2;
  const unsigned long long* gr_in = a.granules + (size_t)(has_prev ? gord - 1 : 0) * a.gran_stride * 2;
  if (XG) {   // the band pipeline continues across GPUs: same granules, same epochs, system-scope accesses (below)
    if (ord + 1 == a.nb_local && gord + 1 < nb) gr_out = a.xg_out;
    if (ord == 0 && gord > 0) gr_in = a.xg_in;
  }

  // Active range (forward / backward solves only).  Outside the 32-step-aligned block range
  // [B0, B1) every cell of the band is non-fluid, so its results are constants that are already in
  // memory (q, z = +0, zeroed per solve) and the values it would hand on are CONST (z: +0; the
  // forward carry m = (-1*precon)*(+0) = -0.0).  The band runs only [B0, B1); an empty band returns at
  // once, and nobody waits for it.  The previous band's range tells which column blocks it
  // announces: [pB0 - 8, pB1 - 8); outside that window the consumer substitutes CONST and does
  // not poll.  The factor sweep always runs the full range (a stale precon is not a constant).
  constexpr bool RANGED = OP != SW_FACTOR;
  constexpr double CONST = OP == SW_FORWARD ? -0.0 : 0.0;
  constexpr int BODY_HALF = 4;                                          // blocks a range is a multiple of: the loop body
  const int full_blocks = BODY_HALF * (((T + SW_BLK - 1) / SW_BLK + BODY_HALF - 1) / BODY_HALF);
  const int ncolblk = (X + SW_BLK - 1) / SW_BLK;
  int B0 = 0, B1 = full_blocks, win_lo = 0, win_hi = ncolblk;
  if (RANGED && a.ranges) {
    const int4 mine = a.ranges[band];
    B0 = BWD ? mine.z : mine.x; B1 = BWD ? mine.w : mine.y;
    if (B0 >= B1) {                                    // no fluid in this band
      if (OP == SW_FORWARD && a.fin_qq >= 0 && role == 0) sweep_qq_arrive(a, ord, 0.0);
      return;
    }
    if (has_prev) {
      const int4 prv = a.ranges[BWD ? band + 1 : band - 1];
      const int pB0 = BWD ? prv.z : prv.x, pB1 = BWD ? prv.w : prv.y;
      win_lo = pB0 - 8 > 0 ? pB0 - 8 : 0;
      win_hi = pB1 - 8 < ncolblk ? pB1 - 8 : ncolblk;   // empty producer: win_hi <= win_lo
    }
  }
  const int NBLK = B1 - B0;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

  // =========================== helper waves: the band hand-off ===================================
  // wave 1 announces this band's edge values to the next band, wave 2 fetches the previous band's
  if (role == 1) {
    if (!publish) return;
    const int t8 = lane >> 3, k8 = lane & 7;            // this lane serves group (next + t8), column k8 of it
    auto announce = [&](int col, double v, bool on) {
      if (on && col >= 0 && col < X) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        const u32x4 gq = {(unsigned int)bits, a.epoch, (unsigned int)(bits >> 32), a.epoch};
        if (XG) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(&gr_out[(size_t)col * 2]), "v"(gq) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&gr_out[(size_t)col * 2]), "v"(gq) : "memory");
      }
    };
    int next_pub = 0;                                   // groups announced so far (relative to B0)
    unsigned int spins = 0;
    while (next_pub < NBLK) {
      const int cdone = (int)lds_get(&sh.comp_done);
      SW_COMPILER_FENCE();
      if (next_pub < cdone) {
        // the group of block b = logical columns 8(b-8) .. 8(b-8)+7 of the edge row, produced in steps
        // 8b-1 .. 8b+6, is complete once block b is (up to 8 groups per pass)
        int n = cdone - next_pub; n = n < 8 ? n : 8;
        const int b = B0 + next_pub + t8;
        const double v = sh.pub[(SW_BLK * b - 1 + k8) & (SW_RING - 1)][EDGE];
        announce(SW_BLK * (b - 8) + k8, v, t8 < n);
        if (SW_TRACE_HANDOFF && lane == 0 && SW_TRACE_CB + 8 - B0 >= next_pub && SW_TRACE_CB + 8 - B0 < next_pub + n)
          a.timeline[(size_t)ord * 8 + 5] = wall_clock64();
        next_pub += n;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ring rows are read: the compute wave may reuse them
        lds_put(&sh.pub_done, (unsigned int)next_pub);
        spins = 0;
        continue;
      }
      if (lds_get(&sh.abort)) return;
      if (++spins > (SW_SPIN_LIMIT << 3)) { if (lane == 0) atomicExch(a.error, 2); lds_put(&sh.abort, 1u); return; }
#ifdef SW_ANNOUNCE_SLEEP       // the announce wave yields while it has nothing to announce (tools/r03: sweep ablations)
      __builtin_amdgcn_s_sleep(SW_ANNOUNCE_SLEEP);
#endif
    }
    // The edge row's column of the very last step (8*B1 - 1) opens the group of a block that never runs.
    // The next band reads it only when X + 63 is a multiple of 16 (factor sweep: full range, full window).
    announce(SW_BLK * (B1 - 8), sh.pub[(SW_BLK * B1 - 1) & (SW_RING - 1)][EDGE], lane == 0);
    return;
  }
  if (role == 2) {
    if (!has_prev) return;
    // Four polls are kept in flight (re-issued as they are retired, so they space themselves a quarter
    // of a round trip apart): a granule is then seen about half a round trip after it lands instead
    // of one and a half.  Each poll covers up to 8 blocks from the deposit front at its issue; every
    // lane always loads (clamped address) so that the in-order vmcnt bookkeeping is exact.
    const int t8 = lane >> 3, k8 = lane & 7;            // this lane serves block (base + t8), column k8 of it
    int next_dep = 0;                                   // boundary blocks deposited so far (relative to B0)
    unsigned int spins = 0;
    struct Poll { u32x4 gv; int base, n; };
    Poll q0, q1, q2, q3;
    auto issue = [&](Poll& q) {
      const int cdone = (int)lds_get(&sh.comp_done);
      SW_COMPILER_FENCE();
      int n = cdone + SW_BND_RING - 2 - next_dep;               // ring slots the compute wave is done with
      n = n < NBLK - next_dep ? n : NBLK - next_dep;
      q.n = n < 8 ? n : 8; q.base = next_dep;
      const int blk = B0 + next_dep + t8, xl = SW_BLK * blk + k8;
      const bool want = t8 < q.n && blk >= win_lo && blk < win_hi && xl < X;
      const unsigned long long* p = &gr_in[want ? (size_t)xl * 2 : 0];
      if (XG) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(q.gv) : "v"(p) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(q.gv) : "v"(p) : "memory");
    };
    auto retire = [&](Poll& q) {                        // the oldest of the four polls in flight
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(q.gv) :: "memory");
      const int blk = B0 + q.base + t8, xl = SW_BLK * blk + k8;
      const bool mine = t8 < q.n;
      const bool want = mine && blk >= win_lo && blk < win_hi && xl < X;   // else: nothing is announced there, CONST
      const bool ready = mine && (!want || (q.gv[1] == a.epoch && q.gv[3] == a.epoch));
      const unsigned long long notready = ~__ballot(ready);
      const int m = notready ? (__ffsll((long long)notready) - 1) >> 3 : 8;      // leading blocks whose 8 columns are all there
      const int fresh0 = next_dep - q.base;                                        // blocks a younger poll's elder already deposited
      if (m > fresh0) {
        if (t8 >= fresh0 && t8 < m) sh.bnd[(q.base + t8) & (SW_BND_RING - 1)][k8] = want ? __hiloint2double((int)q.gv[2], (int)q.gv[0]) : CONST;
        SW_COMPILER_FENCE();
        if (SW_TRACE_HANDOFF && lane == 0 && SW_TRACE_CB - B0 >= next_dep && SW_TRACE_CB - B0 < q.base + m)
          a.timeline[(size_t)ord * 8 + 6] = wall_clock64();
        next_dep = q.base + m;
        lds_put(&sh.dep_done, (unsigned int)next_dep);
        spins = 0;
      } else {
        ++spins;
      }
    };
    issue(q0); issue(q1); issue(q2); issue(q3);
    while (next_dep < NBLK) {
      retire(q0); issue(q0);
      retire(q1); issue(q1);
      retire(q2); issue(q2);
      retire(q3); issue(q3);
#ifdef SW_FETCH_SLEEP          // fewer polls per microsecond (tools/r03: sweep ablations)
      __builtin_amdgcn_s_sleep(SW_FETCH_SLEEP);
#endif
      if (lds_get(&sh.abort) || spins > SW_SPIN_LIMIT) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0.gv), "+v"(q1.gv), "+v"(q2.gv), "+v"(q3.gv) :: "memory");   // before their registers are reused
    if (spins > SW_SPIN_LIMIT) { if (lane == 0) atomicExch(a.error, 2); lds_put(&sh.abort, 1u); }
    return;
  }

  // ====================================== compute wave ========================================
  unsigned int stalls = 0;
  // Per-lane stream pointers at (this band, first record pair of the range, this lane).  Records come in
  // pairs (euler_dev.h): a lane's elements of records 2P and 2P+1 are adjacent, so ONE 16-byte access per
  // stream serves two steps - a lone wave pays per memory instruction (~16 cycles each whatever the width).
  // A block = 8 steps = 4 pairs = 4 KB per 8-byte stream; pair p of the block is addressed with the
  // immediate offset p * PSTEP (backward: descending).  T is even, so the backward sweep starts on the odd
  // record of a pair: step 2p is the pair's odd element (.y), step 2p+1 its even one (.x).
  constexpr int PSTEP = BWD ? -1024 : 1024;           // bytes from pair to pair for an 8-byte stream
  const size_t pair0 = (size_t)band * TS * 64 + (size_t)(BWD ? T - 2 - SW_BLK * B0 : SW_BLK * B0) * 64 + 2 * lane;   // element index
  const char* p_in = reinterpret_cast<const char*>((OP == SW_FACTOR ? a.pre : a.in) + pair0);   // operands of the block being prefetched
  const char* p_pre = reinterpret_cast<const char*>(a.pre + pair0);
  const char* p_msk = reinterpret_cast<const char*>(a.mask + pair0);      // factor: 2 mask bytes per pair
  const unsigned int* p_fb = (BWD ? a.fbits_bwd : a.fbits_fwd) + ((size_t)band * a.fb_stride + B0) * 64 + lane;
  char* p_out = reinterpret_cast<char*>((OP == SW_FACTOR ? a.pre : a.out) + pair0);   // results of the block being computed

  // Operand sets in rotation: while block k computes from one set, the records of the next DIST blocks are in
  // flight into the others (HBM latency under load exceeds one block time).  Forward and backward: four sets, distance 3
  // (round 1's backward sweep streamed its coefficients {a_i precon, a_j precon} as a third and fourth 16-byte load per pair
  // and had registers for three sets only; round 2 rebuilds them from precon and two flag bits: 2 loads per pair like the
  // forward sweep, 16 B per cell less traffic, and room for the fourth set).  Factor: two sets, distance 1 (compiler-managed loads).
  constexpr int DIST = OP == SW_FACTOR ? 1 : 3;
  struct Operands { sw_d2 in[4], pre[4]; int m[4]; unsigned int fb; };    // per pair: .x = even record, .y = odd record
  Operands opA, opB, opC, opD;
  // forward / backward: the record loads are issued BY HAND (inline asm) and retired by counted
  // s_waitcnt in front of each pair of steps.  hipcc's own wait insertion loses track of the issue order at
  // control-flow joins and then waits for every operation older than this block's loads -
  // including the result stores issued a few cycles earlier, i.e. a store round trip per block.
  // Memory operations of a wave retire in issue order, and the order here is fixed by construction:
  //     fetch(k):   fb, then per pair p the LOADS_PER_PAIR loads          (LOADS = 4 * LOADS_PER_PAIR + 1)
  //     compute(k): one 16-byte result store behind each pair of steps
  //   => before pair p of block k everything up to the pair's last load is needed, and behind it were issued
  //      (3 - p) * LOADS_PER_PAIR loads of fetch(k), then per block of prefetch distance [4 older stores and] one
  //      whole fetch, and p stores: vmcnt((3 - p) * LOADS_PER_PAIR + DIST * LOADS + p) is exact for the first
  //      blocks and never waits for a younger fetch.
  // tools/check_sweep_isa.py proves on the generated ISA that no in-flight operand is ever touched.
  constexpr int LOADS_PER_PAIR = 2;
  constexpr int LOADS = 4 * LOADS_PER_PAIR + 1;
  auto fetch_block = [&](Operands& o) {
    if constexpr (OP == SW_FACTOR) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        o.in[pp] = sw_d2{0.0, 0.0};
        o.pre[pp] = *reinterpret_cast<const sw_d2*>(p_pre + pp * PSTEP);
        o.m[pp] = (int)*reinterpret_cast<const unsigned short*>(p_msk + pp * (PSTEP / 8));   // the two cell-mask bytes of the pair
      }
      o.fb = 0u;
    } else {
      // only the flags, 8 steps to a dword (bits 0-7 fluid; backward: bits 8-15 fluid to the right, 16-23 fluid above) - a byte
      // load per step costs as much as the rest of the step
      asm volatile("global_load_dword %0, %1, off" : "=&v"(o.fb) : "v"(p_fb) : "memory");
#define SW_LOAD_PAIR(P)                                                                                                     \
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(o.in[P]) : "v"(p_in), "n"((P) * PSTEP));             \
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(o.pre[P]) : "v"(p_pre), "n"((P) * PSTEP));           \
      o.m[P] = 0;
      SW_LOAD_PAIR(0) SW_LOAD_PAIR(1) SW_LOAD_PAIR(2) SW_LOAD_PAIR(3)
#undef SW_LOAD_PAIR
      asm volatile("" ::: "memory");
    }
    p_in += 4 * PSTEP; p_pre += 4 * PSTEP; p_msk += 4 * (PSTEP / 8); p_fb += 64;
  };

  // wait (rarely) until the helper's counter reaches `target`
  auto await = [&](const unsigned int* cnt, int target) {
    ++stalls;
    for (unsigned int spins = 0; (int)lds_get(cnt) < target;) {
      if (lds_get(&sh.abort)) break;
      if (++spins > SW_SPIN_LIMIT) { if (lane == 0) atomicExch(a.error, 2); lds_put(&sh.abort, 1u); break; }
      __builtin_amdgcn_s_sleep(1);
    }
    SW_COMPILER_FENCE();
  };

  // Steady state is straight-line code: a lone wave pays ~30 cycles for every taken branch, so the
  // per-band facts (is there a band before us / after us) select one of four instantiations of the loop.
  unsigned long long t_first = 0;
  double qq = 0.0;           // forward: the lane's share of dot(q, q)
  auto sweep = [&](auto hp_c, auto pb_c) {
    constexpr bool HP = decltype(hp_c)::value;    // a band before us in sweep order: boundary values from the helper
    constexpr bool PB = decltype(pb_c)::value;    // a band after us: carry rows for the helper
    fetch_block(opA);                             // (per instantiation: an in-flight operand must never be copied)
    if (DIST >= 2) fetch_block(opB);
    if (DIST >= 3) fetch_block(opC);
    // Loop-carried state.  What travels between cells is, per operation:
    //   factor   : precon itself (left neighbour = own register, lower neighbour = lane-1)
    //   forward  : m = (-1*precon)*q of a cell - exactly the term its right neighbour (same lane, next
    //              step) AND its upper neighbour (lane+1, next step) subtract (main.c:607-609), so it
    //              is formed once and shifted; the precon of the lower row is never loaded
    //   backward : z (the coefficients belong to the consuming cell, main.c:620-622)
    double own = CONST;      // carried value of the previous column of this row
    qq = 0.0;
    double out = CONST;      // carried value this lane hands to the next lane
    // boundary values of the block about to run / of the one after it (what the lane without a shift source
    // receives); two sets like the operands, so that the next block's are read half a block ahead
    double beA[SW_BLK], beB[SW_BLK];
    auto read_boundary = [&](int rel, double (&be)[SW_BLK]) {
      const double* slot = sh.bnd[rel & (SW_BND_RING - 1)];
#pragma unroll
      for (int j = 0; j < SW_BLK; ++j) be[j] = HP ? slot[j] : CONST;
    };
    if (PB) sh.pub[(SW_BLK * B0 - 1) & (SW_RING - 1)][lane] = CONST;   // "step B0*8 - 1": the band was all non-fluid before its range
    if (HP) await(&sh.dep_done, 1);
    read_boundary(0, beA);
    t_first = wall_clock64();

    // one hand-off block = 8 steps: compute from `cur`, refill `nxt` with the block after the next
    auto run_block = [&](int blk, Operands& cur, Operands& nxt, double (&be)[SW_BLK], double (&be_next)[SW_BLK]) {
      const int rel = blk - B0;
      fetch_block(nxt);
      // the helpers' progress, read well ahead of its use: boundary blocks deposited, groups announced
      unsigned long long prog = 0;
      if (HP || PB) prog = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&sh.dep_done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // Pin the software pipeline: all loads of block blk+1 are issued here, ahead of the compute phase
      // (left alone, hipcc's scheduler sinks them between the steps and waits for them a few
      // instructions later; measured 1.4-2x slower per step).
      __builtin_amdgcn_sched_barrier(0);
      double* ring = &sh.pub[(SW_BLK * blk) & (SW_RING - 1)][lane];
      double prev_carry = CONST, prev_res = 0.0;
      auto step = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int P = j >> 1;                         // the pair of records this step belongs to
        constexpr bool ODD = ((j & 1) != 0) != BWD;       // does the step use the pair's odd record (.y)?
        if constexpr (OP != SW_FACTOR && (j & 1) == 0) {  // retire this pair's records (see fetch_block)
          constexpr int N0 = (3 - P) * LOADS_PER_PAIR + DIST * LOADS + P, N = N0 < 63 ? N0 : 63;
          asm volatile("s_waitcnt vmcnt(%3)" : "+v"(cur.in[P]), "+v"(cur.pre[P]), "+v"(cur.fb) : "n"(N) : "memory");
        }
        // the edge lane consumes logical column s = 8*blk + j of the previous band
        const double nbv = wave_shift_inject<CTRL>(out, be[j]);   // 2 DPP moves
        const double cin = ODD ? cur.in[P].y : cur.in[P].x, cpre = ODD ? cur.pre[P].y : cur.pre[P].x;
        // sign-extended fluid flag (0 / -1): masking is two v_and
        const int cm = OP == SW_FACTOR ? ((cur.m[P] >> (ODD ? 8 : 0)) & 0xff) : ((int)(cur.fb << (31 - j)) >> 31);
        double res, carry;
        if (OP == SW_FACTOR) {               // main.c:586-600; own / nbv are precon of the left / lower cell
          const double aa = (double)(cm >> CM_DIAG_SHIFT);
          const double cl = -1.0 * own, cb = -1.0 * nbv;
          double e = aa - cl * cl - cb * cb;
          if (e < 0.25 * aa) e = (aa != 0.0) ? aa : 1.0;
          res = (cm & CM_FLUID) ? 1.0 / sqrt(e) : cpre;      // non-fluid: the stale entry stays
          carry = res;
        } else if (OP == SW_FORWARD) {       // main.c:602-613: t = r - (-1*pre_l)*q_l - (-1*pre_b)*q_b
          const double t = cin - own - nbv;
          const double qv = t * cpre;
          res = __hiloint2double(__double2hiint(qv) & cm, __double2loint(qv) & cm);   // +0 on non-fluid cells
          carry = -1.0 * cpre * res;         // this cell's term in its right and upper neighbours
          qq = __builtin_fma(res, res, qq);  // dot(q, q) = dot(z, r) (SweepArgs::fin_qq): one instruction, not part of the bit-exact arithmetic
        } else {                             // main.c:615-626: t = q - (a_i*pre)*z_r - (a_j*pre)*z_u
          // the cell's coefficients a_i precon, a_j precon (main.c:621-622), a = -1 / 0 by the fluid flag of the right / upper
          // neighbour: (-1.0 or +0.0) * precon, the reference's product, from two sign-extended flag bits (off the carried chain)
          const int fr = (int)(cur.fb << (23 - j)) >> 31, fu = (int)(cur.fb << (15 - j)) >> 31;
          const double kr = __hiloint2double((int)0xBFF00000 & fr, 0) * cpre, ku = __hiloint2double((int)0xBFF00000 & fu, 0) * cpre;
          const double t = cin - kr * own - ku * nbv;
          const double zv = t * cpre;
          res = __hiloint2double(__double2hiint(zv) & cm, __double2loint(zv) & cm);   // +0 on non-fluid cells
          carry = res;
        }
        // the pair's two results leave with one 16-byte store behind its second step ({even, odd} record order)
        if (j & 1) *reinterpret_cast<sw_d2*>(p_out + P * PSTEP) = BWD ? sw_d2{res, prev_res} : sw_d2{prev_res, res};
        prev_res = res;
        own = carry;
        out = carry;
        // the announce wave gathers the edge lane's entry; two rows per LDS instruction (ds_write2st64_b64)
        if (PB && (j & 1)) { ring[(j - 1) * 64] = prev_carry; ring[j * 64] = carry; }
        prev_carry = carry;
      };
      step(std::integral_constant<int, 0>()); step(std::integral_constant<int, 1>()); step(std::integral_constant<int, 2>());
      step(std::integral_constant<int, 3>());
      step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 5>());
      if (HP) {
        // two steps before the block ends (an LDS round trip): the next block's boundary values (deposited once
        // dep_done > rel+1; past the range: whatever is there).  A starved wave waits here, as late as possible.
        if (__builtin_expect(rel + 1 < NBLK && (int)(unsigned int)prog < rel + 2, 0)) await(&sh.dep_done, rel + 2);
        read_boundary(rel + 1, be_next);
        if (SW_TRACE_HANDOFF && blk + 1 == SW_TRACE_CB && lane == 0) a.timeline[(size_t)ord * 8 + 7] = wall_clock64();
      }
      step(std::integral_constant<int, 6>()); step(std::integral_constant<int, 7>());
      __builtin_amdgcn_sched_barrier(0);
      p_out += 4 * PSTEP;
      if (HP || PB) { SW_COMPILER_FENCE(); lds_put(&sh.comp_done, (unsigned int)(rel + 1)); }
      if (SW_TRACE_HANDOFF && blk == SW_TRACE_CB + 8 && lane == 0) a.timeline[(size_t)ord * 8 + 4] = wall_clock64();
      // block blk+1 overwrites the ring rows of block blk-7, which the groups up to block blk-6 read
      // (announced once pub_done >= rel-5)
      if (PB && __builtin_expect(rel + 1 < NBLK && (int)(unsigned int)(prog >> 32) < rel - 5, 0)) await(&sh.pub_done, rel - 5);
    };

    // At block k the wave computes from set k mod (DIST + 1) and refills the set of block k - 1.  The body is 4 blocks
    // and the ranges are multiples of 4 in every sweep.  (A mid-body exit is correct too, but hipcc then merges the tails of
    // the rare wait paths and tools/check_sweep_isa.py, which follows every branch both ways, can no longer prove it.)
    for (int blk = B0; blk < B1; blk += 4) {
      if (DIST == 3) {          // forward, backward: four sets
        run_block(blk, opA, opD, beA, beB);
        run_block(blk + 1, opB, opA, beB, beA);
        run_block(blk + 2, opC, opB, beA, beB);
        run_block(blk + 3, opD, opC, beB, beA);
      } else {                  // factor: two sets
        run_block(blk, opA, opB, beA, beB);
        run_block(blk + 1, opB, opA, beB, beA);
        run_block(blk + 2, opA, opB, beA, beB);
        run_block(blk + 3, opB, opA, beB, beA);
      }
    }
    // retire the prefetch that ran past the range before anything else reuses its registers (the kernel
    // end would wait for it anyway; it also keeps tools/check_sweep_isa.py's path exploration exact)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;
  if (has_prev) { if (publish) sweep(yes_t(), yes_t()); else sweep(yes_t(), no_t()); }
  else          { if (publish) sweep(no_t(), yes_t()); else sweep(no_t(), no_t()); }
  if (OP == SW_FORWARD && a.fin_qq >= 0) sweep_qq_arrive(a, ord, qq);
  if (lane == 0) {
    unsigned long long* tl = a.timeline + (size_t)ord * 8;
    tl[0] = t_entry; tl[1] = t_first; tl[2] = wall_clock64(); tl[3] = ((unsigned long long)(B1 - B0) << 32) | stalls;
  }
}


// ==========================================================================================
// Tile-local IC(0) (EULER_PRECOND_IC0_TILE): the three recurrences above restricted to blocks - a block = the cells of one
// band whose records fall into one tile [k W, (k+1) W), a parallelogram of 64 rows x W columns.  In the band-skewed layout
// a tile is W consecutive records = one contiguous piece of every solver array, and it is small enough to live in the
// REGISTERS of one wave (W = 16: 16 doubles per lane and vector).  So one wave loads a tile's r, A s and precon once, and
// does everything the PCG iteration needs between two apply_a passes without another trip to memory:
//     r -= alpha A s (fmadd, main.c:754) ; max |r| (inf_norm, main.c:756) ; q = L^-1 r ; z = L^-T q (main.c:602-626) ; dot(z, r)
// = kernels K2, K3, K4 and dot() of SURVEY 8d in ONE pass: 5 w + 1 = 41 bytes per cell instead of 6w+1 + 3w+1 + 4w+1 + ... .
// (p += alpha s rides along with the next apply_a pass, which reads s anyway: k_search_apply<.., true>.)
// The wavefront inside the tile is the one of k_sweep_skew (lane l at record t, lower / upper row by DPP wave shifts), fully
// unrolled with static register indices; tiles are independent, so thousands of waves stream at once and the kernel is
// bound by HBM bandwidth, not by a dependency chain.  The compiler schedules it (nothing to hand-issue: occupancy hides
// the latency).  Each wave walks tiles wave_id, wave_id + n_waves, ...; per-lane sums are folded in that fixed order,
// then per wave, per block, and by the last block over all blocks in index order: deterministic.
#define PT_THREADS 256
struct TileArgs {
  SkewGeom g;
  const uint8_t* mask;
  double* pre;            // factor: in/out; solve: in
  double* r;              // solve: r (updated in place when rupd)
  const double* as;       // A s of this iteration (rupd only)
  double* z;
  int band_lo, nb_local;
  int rupd;               // 1: r -= alpha A s first and report max |r| (a PCG iteration); 0: z = M^-1 r only (start of a solve)
  int sweeps;             // 0: the last iteration of the budget - only r and its norm are needed
  int fin_dot;            // scalar epilogue of dot(z, r): FIN_SIGMA_INIT / FIN_BETA / FIN_STORE_ONLY, -1 = none (replayed sequentially)
  int via;                // 0 single rank, FIN_VIA_P2P, or FIN_TO_COMM (multi-rank: how the two results reach the other ranks)
  double* part_max; double* part_dot;
  unsigned int* counter;
  const unsigned int* list;   // W == 16 inside a solve: the ascending list of active tiles (euler_dev.h "Active chunks"), else null
  const double* table;        // W == 16: E^-1 of an interior tile (k_tile_table); with it, listed interior tiles skip masks and precon
  PcgScalars* sc;
  int force;
  double alpha_arg;       // force: alpha of the r update (single building block, tests)
  double* pair_slot;      // FIN_TO_COMM: where this rank's {max |r|, dot(z,r)} go (its slot of the all-gather buffer)
  // several ranks, compact ghost rows (k_search_apply SLAB 2): the slab's lowest / highest row of z goes out as a row of X doubles,
  // written by the lanes that hold it (lane 0 of band edge_lo, lane 63 of band edge_hi); null / -1 = no neighbour on that side
  double *zsend_lo, *zsend_hi;
  int edge_lo, edge_hi;
  int reverse;            // walk the tiles in descending order
  double* cpart;          // two-level mode (k_coarse.hip): per tile, the sums of the (updated) r over its fluid cells by coarse column: [tile][3]; multilevel mode (k_mg.hip): [tile][MG_PART = 72] = [group][row slot][column slot] (k_mg.h)
  int cshift;             // two-level mode: log2 of the coarse cell width in grid cells (the multilevel mode's node spacing is the constant MG_G0: nothing reads this there)
  int cmode, cnx, cny;    // cmode 2: the multilevel mode's bilinear restriction onto cnx x cny nodes of level 0
  // RECOMP (k_precond_tile<16, true>): `as` is the search direction s' itself and the pass forms A s' from it.  On row slabs the rows across
  // the slab boundary are the compact ghost rows of s' that k_search_apply SLAB 2 keeps (indexed by the column); null = no neighbouring slab
  const double *gs_lo, *gs_hi;
};

// fixed-shape reductions of a PT_THREADS block; result valid in thread 0
__device__ __forceinline__ void tile_block_reduce(double& mx, double& sm) {
  __shared__ double s_mx[PT_THREADS / 64], s_sm[PT_THREADS / 64];
  mx = eu_wave_max(mx);
  sm = eu_wave_sum(sm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_sm[threadIdx.x >> 6] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = s_mx[0]; sm = s_sm[0];
    for (int k = 1; k < PT_THREADS / 64; ++k) { mx = s_mx[k] > mx ? s_mx[k] : mx; sm += s_sm[k]; }
  }
}

// one cell of the E^-1 recurrence (main.c:586-600): own / nbv = precon of the left / lower neighbour
__device__ __forceinline__ double factor_step(double aa, double own, double nbv) {
  const double cl = -1.0 * own, cb = -1.0 * nbv;
  double e = aa - cl * cl - cb * cb;
  if (e < 0.25 * aa) e = (aa != 0.0) ? aa : 1.0;
  return 1.0 / sqrt(e);
}

// E^-1 of an INTERIOR tile of 16 records (every cell fluid with a_diag 4): the recurrence starts from precon 0 on the tile's left
// edge and below lane 0 and sees the same coefficients everywhere, so its 16 x 64 values are the same for every interior tile of
// every solve.  Computed once per handle by the arithmetic of k_factor_tile (same bits); k_precond_tile keeps it in LDS instead of
// streaming 8 bytes per cell of precon from HBM.  One wave.  tab[P][lane] = {record 2P, record 2P + 1}.
__global__ __launch_bounds__(64) void k_tile_table(double* __restrict__ tab) {
  const int lane = threadIdx.x & 63;
  double own = 0.0, out = 0.0;
  sw_d2 pp[8];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double nbv = wave_shift_inject<DPP_WAVE_SHR1>(out, 0.0);
    const double res = factor_step(4.0, own, nbv);
    own = res; out = res;
    if (j & 1) pp[j >> 1].y = res; else pp[j >> 1].x = res;
  }
#pragma unroll
  for (int P = 0; P < 8; ++P) reinterpret_cast<sw_d2*>(tab)[P * 64 + lane] = pp[P];
}
int eu_launch_tile_table(euler_sim* S) {
  hipLaunchKernelGGL(k_tile_table, dim3(1), dim3(64), 0, S->stream, S->tile_table);
  return EULER_OK;
}

// sums over aligned groups of lanes: two quad permutes (and the mirror of the half row), fixed order
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum(double v) {      // over an aligned group of MG_LG lanes (k_mg.h), in every lane of it
  v = v + dpp_move<0xB1>(v);       // quad_perm [1, 0, 3, 2]
  v = v + dpp_move<0x4E>(v);       // quad_perm [2, 3, 0, 1]
  if (MG_LG == 8) v = v + dpp_move<0x141>(v);      // row_half_mirror
  return v;
}

template <int W>
__global__ __launch_bounds__(PT_THREADS) void k_factor_tile(TileArgs a) {
  if (!a.force && pcg_idle(a.sc)) return;
  const int lane = threadIdx.x & 63;
  const int ntb = a.g.T / W, total = a.nb_local * ntb;
  const int n_waves = gridDim.x * (PT_THREADS / 64);
  for (int tile = blockIdx.x * (PT_THREADS / 64) + (threadIdx.x >> 6); tile < total; tile += n_waves) {
    const int band = a.band_lo + tile / ntb, k = tile % ntb;
    const size_t base = ((size_t)band * a.g.TS + (size_t)k * W) * 64 + 2 * lane;
    unsigned int mm[W / 2];
    unsigned int any = 0;
#pragma unroll
    for (int P = 0; P < W / 2; ++P) { mm[P] = *reinterpret_cast<const unsigned short*>(a.mask + base + P * 128); any |= mm[P]; }
    if (!__ballot(((any | (any >> 8)) & CM_FLUID) != 0)) continue;     // no fluid in this tile: precon stays what it is
    sw_d2 pp[W / 2];
#pragma unroll
    for (int P = 0; P < W / 2; ++P) pp[P] = *reinterpret_cast<const sw_d2*>(a.pre + base + P * 128);
    double own = 0.0, out = 0.0;      // a tile starts where a band starts: precon 0 to the left and below
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
      const double cpre = (j & 1) ? pp[j >> 1].y : pp[j >> 1].x;
      const double nbv = wave_shift_inject<DPP_WAVE_SHR1>(out, 0.0);
      const double aa = (double)(cm >> CM_DIAG_SHIFT);               // main.c:586-600
      const double res = (cm & CM_FLUID) ? factor_step(aa, own, nbv) : cpre;     // non-fluid: the stale entry stays (and is what the neighbours read)
      own = res; out = res;
      if (j & 1) pp[j >> 1].y = res; else pp[j >> 1].x = res;
    }
#pragma unroll
    for (int P = 0; P < W / 2; ++P) *reinterpret_cast<sw_d2*>(a.pre + base + P * 128) = pp[P];
  }
}

// RECOMP (W == 16, inside a solve): the r update's A s' is not read back from memory - k_search_apply did not store it - but formed
// again from s' with the expression of k_search_apply / k_apply_a (main.c:679-691: diag, right, up, left, down; identical bits): the
// tile's 16 records of s' plus the pair-record before and after it, the rows below lane 0 / above lane 63 by ONE load (lanes 0-15 fetch
// the 16 values below, lanes 48-63 the 16 above; v_readlane hands them to the DPP shifts).  8 bytes per cell and iteration that are
// neither written nor read: 8192^2, k_search_apply 296-314 -> 232-239 us, this kernel 210-214 -> 209-223 us.
#ifndef PT_RECOMP_BLOCKS
#define PT_RECOMP_BLOCKS 1
#endif
// CMODE: 0 no coarse part, 1 two-level mode (three sums per tile), 2 multilevel mode (the bilinear restriction: 48 sums per tile) - a template parameter so that the
// tile-local mode's own instantiation keeps its registers (151: three waves per SIMD)
template <int W, bool RECOMP = false, int CMODE = 0>
__global__ __launch_bounds__(PT_THREADS, RECOMP ? PT_RECOMP_BLOCKS : 1) void k_precond_tile(TileArgs a) {
  if (!a.force && pcg_idle(a.sc)) return;
  const int lane = threadIdx.x & 63;
  const int ntb = a.g.T / W, total = a.nb_local * ntb;
  const int n_waves = gridDim.x * (PT_THREADS / 64);
  const double nalpha = -(a.force ? a.alpha_arg : a.sc->alpha);
  double mx = 0.0, dsum = 0.0;
  const bool listed = W == 16 && a.list != nullptr;
  const int todo = listed ? (int)a.sc->n_chunks : total;
  // E^-1 of an interior tile, the same for all of them (k_tile_table): in LDS for the whole launch
  constexpr int TABP = W == 16 ? 8 : 1;
  __shared__ sw_d2 s_tab[TABP][64];
  __shared__ double s_cpart[CMODE == 2 ? PT_THREADS / 64 : 1][CMODE == 2 ? 16 * 2 * MG_NSLOT : 1];      // multilevel mode: a wave's partial sums on their way out
  (void)s_cpart;
  const bool have_tab = W == 16 && listed && a.table != nullptr;
  if (have_tab) {
    for (int k = threadIdx.x; k < TABP * 64; k += PT_THREADS) (&s_tab[0][0])[k] = reinterpret_cast<const sw_d2*>(a.table)[k];
    __syncthreads();
  }
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;
  for (int i = blockIdx.x * (PT_THREADS / 64) + (threadIdx.x >> 6); i < todo; i += n_waves) {
    // (reverse: start where the previous pass - k_search_apply, ascending - ended, i.e. on what the Infinity Cache still holds)
    const int ii = a.reverse ? todo - 1 - i : i;
    const unsigned int ent = listed ? a.list[ii] : (unsigned int)ii;
    const bool interior = have_tab && (ent & EU_CHUNK_INTERIOR) != 0;
    const int tile = (int)(listed ? ent & ~EU_CHUNK_INTERIOR : ent);
    const int band = a.band_lo + tile / ntb, k = tile % ntb;
    const size_t base = ((size_t)band * a.g.TS + (size_t)k * W) * 64 + 2 * lane;
    auto run = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;      // interior tile: masks are constants, precon comes from the table
      unsigned int mm[W / 2];
      unsigned int any = 0;
#pragma unroll
      for (int P = 0; P < W / 2; ++P) {
        mm[P] = FULL ? (unsigned int)(CM_INTERIOR | (CM_INTERIOR << 8)) : (unsigned int)*reinterpret_cast<const unsigned short*>(a.mask + base + P * 128);
        any |= mm[P];
      }
      if (!FULL && !listed && !__ballot(((any | (any >> 8)) & CM_FLUID) != 0)) return;     // no fluid in this tile: r, z stay +0 there
      sw_d2 rr[W / 2], qq[W / 2], pp[W / 2];
#pragma unroll
      for (int P = 0; P < W / 2; ++P) {
        rr[P] = *reinterpret_cast<const sw_d2*>(a.r + base + P * 128);
        if (RECOMP && a.rupd) continue;              // (E^-1 is fetched behind the r update: PT_RECOMP_LATE)
        if (!a.sweeps) pp[P] = sw_d2{0.0, 0.0};      // (the r update alone - the parity mode's use of this kernel - needs no E^-1)
        else if (FULL) pp[P] = s_tab[P < TABP ? P : 0][lane];
        else pp[P] = *reinterpret_cast<const sw_d2*>(a.pre + base + P * 128);
        if (!RECOMP && a.rupd) qq[P] = *reinterpret_cast<const sw_d2*>(a.as + base + P * 128);
      }
      if (RECOMP && a.rupd) {
        const int npairs = a.g.TS / 2, P0 = k * (W / 2);
        sw_d2 ss[W / 2 + 2];
#pragma unroll
        for (int P = -1; P <= W / 2; ++P)
          ss[P + 1] = (P0 + P >= 0 && P0 + P < npairs) ? *reinterpret_cast<const sw_d2*>(a.as + base + (long long)P * 128) : sw_d2{0.0, 0.0};
        // lane L < 16: s' of the cell below lane 0's cell of record k W + L (column k W + L, row 64 band - 1);
        // lane L >= 48: of the cell above lane 63's cell of record k W + L - 48 (column k W + L - 48 - 63, row 64 (band + 1))
        double ev = 0.0;
        if (lane < 16) {
          const int x = k * W + lane;
          if (x < a.g.X) {
            if (a.gs_lo && band == a.band_lo) ev = a.gs_lo[x];
            else if (band > 0) ev = a.as[skew_index(a.g, x, 64 * band - 1)];
          }
        } else if (lane >= 48) {
          const int x = k * W + lane - 48 - 63;
          if (x >= 0 && x < a.g.X) {
            if (a.gs_hi && band == a.band_lo + a.nb_local - 1) ev = a.gs_hi[x];
            else if (band + 1 < a.g.nbands) ev = a.as[skew_index(a.g, x, 64 * (band + 1))];
          }
        }
        const int ev_lo = __double2loint(ev), ev_hi = __double2hiint(ev);
#pragma unroll
        for (int P = 0; P < W / 2; ++P) {
          const unsigned int m0 = mm[P] & 0xff, m1 = mm[P] >> 8;
          const sw_d2 cc = ss[P + 1];
          const double prev_y = ss[P].y, nxt_x = ss[P + 2].x;
          const double d0 = __hiloint2double(__builtin_amdgcn_readlane(ev_hi, 2 * P), __builtin_amdgcn_readlane(ev_lo, 2 * P));
          const double d1 = __hiloint2double(__builtin_amdgcn_readlane(ev_hi, 2 * P + 1), __builtin_amdgcn_readlane(ev_lo, 2 * P + 1));
          const double u0 = __hiloint2double(__builtin_amdgcn_readlane(ev_hi, 48 + 2 * P), __builtin_amdgcn_readlane(ev_lo, 48 + 2 * P));
          const double u1 = __hiloint2double(__builtin_amdgcn_readlane(ev_hi, 49 + 2 * P), __builtin_amdgcn_readlane(ev_lo, 49 + 2 * P));
          const double dn0 = wave_shift_inject<DPP_WAVE_SHR1>(prev_y, d0), up0 = wave_shift_inject<DPP_WAVE_SHL1>(cc.y, u0);
          const double dn1 = wave_shift_inject<DPP_WAVE_SHR1>(cc.x, d1), up1 = wave_shift_inject<DPP_WAVE_SHL1>(nxt_x, u1);
          double v = (double)(int)(m0 >> CM_DIAG_SHIFT) * cc.x;      // apply_a (main.c:679-691): diag, right, up, left, down
          v = v - ((m0 & CM_RIGHT) ? cc.y : 0.0);
          v = v - ((m0 & CM_UP) ? up0 : 0.0);
          v = v - ((m0 & CM_LEFT) ? prev_y : 0.0);
          v = v - ((m0 & CM_DOWN) ? dn0 : 0.0);
          // r -= alpha A s' (fmadd, main.c:754) and max |r| over fluid cells, as below
          if (m0 & CM_FLUID) { rr[P].x = rr[P].x + v * nalpha; const double w = fabs(rr[P].x); if (w > mx) mx = w; }
          v = (double)(int)(m1 >> CM_DIAG_SHIFT) * cc.y;
          v = v - ((m1 & CM_RIGHT) ? nxt_x : 0.0);
          v = v - ((m1 & CM_UP) ? up1 : 0.0);
          v = v - ((m1 & CM_LEFT) ? cc.x : 0.0);
          v = v - ((m1 & CM_DOWN) ? dn1 : 0.0);
          if (m1 & CM_FLUID) { rr[P].y = rr[P].y + v * nalpha; const double w = fabs(rr[P].y); if (w > mx) mx = w; }
          *reinterpret_cast<sw_d2*>(a.r + base + P * 128) = rr[P];
        }
      }
      if (!RECOMP && a.rupd) {      // r -= alpha z (fmadd, main.c:754, evaluated as r + z * (-alpha) like k_update_pr) and max |r| over fluid cells
#pragma unroll
        for (int P = 0; P < W / 2; ++P) {
          if (mm[P] & CM_FLUID) { rr[P].x = rr[P].x + qq[P].x * nalpha; const double v = fabs(rr[P].x); if (v > mx) mx = v; }
          if ((mm[P] >> 8) & CM_FLUID) { rr[P].y = rr[P].y + qq[P].y * nalpha; const double v = fabs(rr[P].y); if (v > mx) mx = v; }
          *reinterpret_cast<sw_d2*>(a.r + base + P * 128) = rr[P];
        }
      }
      if (!a.sweeps) return;
      if (W == 16 && CMODE == 2) {
        // multilevel mode (k_mg.hip): P_0^T r of this tile, P_0 bilinear from the nodes at the cells (G0 J + G0 / 2, G0 I + G0 / 2), G0 = MG_G0 = 8.  The eight lanes of a GROUP
        // (k_mg.h) lie between the same two node rows I0, I0 + 1; a lane's 16 columns between at most MG_NSEG + 1 node columns starting at Jb (its own), the group's between
        // MG_NSLOT = 4 starting at Jq.  A record's weight of the right-hand node is rec + b with b constant over a segment, so a lane accumulates sum rv and sum rec rv per
        // segment, turns them into its node-column sums, shifts them to the group's slots and multiplies by its two row weights (weights in 1 / G0)
        const int quad = lane >> 2, G = (lane + 4) >> 3;      // the group: lanes 8 G - 4 .. 8 G + 3 (k_mg.h)
        const int uy = 64 * band + lane - MG_G0 / 2, I0 = uy >> MG_LOG;
        double wy1 = (double)(uy & (MG_G0 - 1)), wy0 = (double)MG_G0 - wy1;
        if (I0 < 0) { wy0 = 0.0; wy1 = (double)MG_G0; }
        if (I0 >= a.cny - 1) { wy0 = (double)MG_G0; wy1 = 0.0; }
        const int x0 = 16 * k - lane - MG_G0 / 2, Jb = x0 >> MG_LOG, tbase = MG_G0 * Jb - x0;      // tbase in (-G0, 0]: records >= tbase + G0 m lie in segment m
        const int Jq = 2 * k - G - 1;      // the group's first node column: Jb - Jq is 1 for the group's first lane (whose third segment is empty), 0 for the others
        double sg_s[MG_NSEG], sg_t[MG_NSEG];
#pragma unroll
        for (int m = 0; m < MG_NSEG; ++m) { sg_s[m] = 0.0; sg_t[m] = 0.0; }
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
          const double rv = (cm & CM_FLUID) ? ((j & 1) ? rr[j >> 1].y : rr[j >> 1].x) : 0.0;
          const double jv = rv * (double)j;
#pragma unroll
          for (int m = 0; m < MG_NSEG; ++m) {
            const bool in = (m == 0 || j >= tbase + MG_G0 * m) && (m == MG_NSEG - 1 || j < tbase + MG_G0 * (m + 1));
            sg_s[m] += in ? rv : 0.0; sg_t[m] += in ? jv : 0.0;
          }
        }
        double cn[MG_NSEG + 1];
#pragma unroll
        for (int m = 0; m <= MG_NSEG; ++m) cn[m] = 0.0;
#pragma unroll
        for (int m = 0; m < MG_NSEG; ++m) {
          double u1 = sg_t[m] + (double)(-tbase - MG_G0 * m) * sg_s[m], u0 = (double)MG_G0 * sg_s[m] - u1;      // weights of the nodes Jb + m + 1 / Jb + m
          if (Jb + m < 0) { u0 = 0.0; u1 = (double)MG_G0 * sg_s[m]; }                                             // beyond the outermost nodes: constant
          if (Jb + m >= a.cnx - 1) { u0 = (double)MG_G0 * sg_s[m]; u1 = 0.0; }
          cn[m] += u0; cn[m + 1] += u1;
        }
        const bool shifted = Jb != Jq;
        wy0 *= 1.0 / (MG_G0 * MG_G0); wy1 *= 1.0 / (MG_G0 * MG_G0);
        // Two DPP steps add the products over a quad; the two quads of a group meet in a row of LDS on the way out, and the tile's MG_PART sums leave as ONE contiguous
        // piece of LDS; they leave as nine pieces of eight doubles (below; single scattered doubles cost 44 us per pass at 8192^2, rounds 3-5 wrote one piece of 72 per tile)
        double* sp = s_cpart[threadIdx.x >> 6];
#pragma unroll
        for (int q = 0; q < MG_NSLOT; ++q) {
          const double lo = cn[q], hi = q >= 1 ? cn[q - 1] : 0.0;
          const double cs = shifted ? hi : lo;
          const double p0 = group_sum(wy0 * cs), p1 = group_sum(wy1 * cs);
          if ((lane & 3) == 0) { sp[quad * 2 * MG_NSLOT + q] = p0; sp[quad * 2 * MG_NSLOT + MG_NSLOT + q] = p1; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
          // [band][group][tile][row slot][column slot] (round 6; [band][tile][group][row slot][column slot] before): a tile leaves nine whole 64-byte lines, next to its
          // neighbours' - what a node row of k_mg_down1 gathers (two tiles' slots per node, node after node) is then CONTIGUOUS: the gather fetched 3.5 x the bytes it used
          double* cp = a.cpart + ((size_t)(tile / ntb) * MG_NGRP * ntb + (size_t)k) * (2 * MG_NSLOT);
#pragma unroll
          for (int u = 0; u < (MG_PART + 63) / 64; ++u) {
            const int e = lane + 64 * u;
            if (e < MG_PART) {
              const int g = e / (2 * MG_NSLOT), w = e % (2 * MG_NSLOT);      // group g = quads 2 g - 1 and 2 g (the half groups: quad 0 / quad 15 alone)
              const double va = g > 0 ? sp[(2 * g - 1) * 2 * MG_NSLOT + w] : 0.0, vb = g < 8 ? sp[(2 * g) * 2 * MG_NSLOT + w] : 0.0;
              cp[((size_t)g * ntb) * (2 * MG_NSLOT) + w] = va + vb;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      } else if (W == 16 && CMODE == 1) {      // P^T r of this tile: cell (lane, record k W + j) sits in column k W + j - lane; at most three coarse columns per tile
        const int xl = k * W - 63, J0 = (xl > 0 ? xl : 0) >> a.cshift;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
          const double rv = (cm & CM_FLUID) ? ((j & 1) ? rr[j >> 1].y : rr[j >> 1].x) : 0.0;
          const int x = k * W + j - lane;
          const int b = ((x > 0 ? x : 0) >> a.cshift) - J0;
          c0 += b == 0 ? rv : 0.0; c1 += b == 1 ? rv : 0.0; c2 += b == 2 ? rv : 0.0;
        }
        c0 = eu_wave_sum(c0); c1 = eu_wave_sum(c1); c2 = eu_wave_sum(c2);
        if (lane == 0) { double* cp = a.cpart + (size_t)tile * 3; cp[0] = c0; cp[1] = c1; cp[2] = c2; }
      }
      if (RECOMP && a.rupd) {      // the window of s' is dead: E^-1 takes its registers (the barrier keeps the compiler from hoisting these loads above the r update)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int P = 0; P < W / 2; ++P) pp[P] = FULL ? s_tab[P < TABP ? P : 0][lane] : *reinterpret_cast<const sw_d2*>(a.pre + base + P * 128);
      }
      // L q = r (main.c:602-613).  What travels from cell to cell is m = (-1 * precon) * q, the term both consumers subtract.
      double own = -0.0, out = -0.0;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
        const double cin = (j & 1) ? rr[j >> 1].y : rr[j >> 1].x, cpre = (j & 1) ? pp[j >> 1].y : pp[j >> 1].x;
        const double nbv = wave_shift_inject<DPP_WAVE_SHR1>(out, -0.0);
        const double t = cin - own - nbv;
        const double qv = t * cpre;
        const double res = (cm & CM_FLUID) ? qv : 0.0;
        const double carry = -1.0 * cpre * res;
        own = carry; out = carry;
        if (j & 1) qq[j >> 1].y = res; else qq[j >> 1].x = res;
      }
      // L^T z = q (main.c:615-626), from the tile's last record down; dot(z, r) on the fly
      own = 0.0; out = 0.0;
#pragma unroll
      for (int j = W - 1; j >= 0; --j) {
        const int cm = (int)((mm[j >> 1] >> ((j & 1) * 8)) & 0xff);
        const double cin = (j & 1) ? qq[j >> 1].y : qq[j >> 1].x, cpre = (j & 1) ? pp[j >> 1].y : pp[j >> 1].x;
        const double crr = (j & 1) ? rr[j >> 1].y : rr[j >> 1].x;
        const double nbv = wave_shift_inject<DPP_WAVE_SHL1>(out, 0.0);
        const double kr = ((cm & CM_RIGHT) ? -1.0 : 0.0) * cpre, ku = ((cm & CM_UP) ? -1.0 : 0.0) * cpre;
        const double t = cin - kr * own - ku * nbv;
        const double zv = t * cpre;
        const double res = (cm & CM_FLUID) ? zv : 0.0;
        own = res; out = res;
        if (cm & CM_FLUID) dsum += res * crr;
        if (j & 1) qq[j >> 1].y = res; else qq[j >> 1].x = res;
      }
#pragma unroll
      for (int P = 0; P < W / 2; ++P) *reinterpret_cast<sw_d2*>(a.z + base + P * 128) = qq[P];
      if (band == a.edge_lo || band == a.edge_hi) {      // (wave-uniform) the rows the neighbouring slabs need, as compact rows
        const bool lo = band == a.edge_lo && lane == 0, hi = band == a.edge_hi && lane == 63;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const double zv = (j & 1) ? qq[j >> 1].y : qq[j >> 1].x;
          const int x_lo = k * W + j, x_hi = k * W + j - 63;      // the column of lane 0 / lane 63 in record k W + j
          if (lo && x_lo < a.g.X) a.zsend_lo[x_lo] = zv;
          if (hi && x_hi >= 0 && x_hi < a.g.X) a.zsend_hi[x_hi] = zv;
        }
      }
    };
    if (W == 16 && interior) run(yes_t()); else run(no_t());
  }
  // ---- the two reductions: block -> partials -> the last block folds them in index order and applies the scalar epilogues
  tile_block_reduce(mx, dsum);
  __shared__ int am_last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.part_max[blockIdx.x]), (unsigned long long)__double_as_longlong(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.part_dot[blockIdx.x]), (unsigned long long)__double_as_longlong(dsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    am_last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!am_last) return;
  double vmax = 0.0, vsum = 0.0;
  for (unsigned int i = threadIdx.x; i < gridDim.x; i += PT_THREADS) {
    const double m = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&a.part_max[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const double d = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&a.part_dot[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    vmax = m > vmax ? m : vmax; vsum += d;
  }
  tile_block_reduce(vmax, vsum);
  if (a.via == FIN_VIA_P2P) {      // uniform: every thread of this block is here
    if (a.rupd) vmax = p2p_allreduce_block<true>(a.sc, vmax);
    if (a.fin_dot >= 0) vsum = p2p_allreduce_block<false>(a.sc, vsum);
  }
  if (threadIdx.x == 0) {
    if (a.via == FIN_TO_COMM) { a.pair_slot[0] = vmax; a.pair_slot[1] = vsum; }   // the epilogues run after the all-gather (k_pair_fold)
    else {
      if (a.rupd) pcg_scalar_step(a.sc, FIN_RNORM, vmax);
      if (a.fin_dot >= 0 && !(a.rupd && a.sc->done)) pcg_scalar_step(a.sc, a.fin_dot, vsum);
    }
    __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// the p += alpha s (fmadd, main.c:753) that are still due when a solve ends: with `iters` iterations counted and N at a time applied by the
// k_search_apply pass of every N-th iteration k >= N (the terms below k), those from the last such k (or from 0) to iters - 1, in their order
struct SRing { const double* s[8]; int n, steps; };      // the ring: s_j sits in s[j % n]
__global__ __launch_bounds__(256) void k_finish_p(double* __restrict__ p, const uint8_t* __restrict__ mask, size_t e_lo, size_t S,
                                                  const PcgScalars* sc, SRing ring) {
  if (!sc->nonzero || sc->iters == 0) return;
  const int n_it = sc->iters;
  const int from = n_it - 1 >= ring.steps ? (n_it - 1) / ring.steps * ring.steps : 0;
  const int cnt = n_it - from;      // 1 .. steps
  double al[8];
  const double* sp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int it = from + (j < cnt ? j : 0); al[j] = sc->alpha_hist[it & 7]; sp[j] = ring.s[it % ring.n] + e_lo; }
  for (size_t i = 2 * ((size_t)blockIdx.x * blockDim.x + threadIdx.x); i < S; i += 2 * (size_t)gridDim.x * blockDim.x) {
    const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + i);
    const bool f0 = (mm & CM_FLUID) != 0, f1 = ((mm >> 8) & CM_FLUID) != 0;
    if (!(f0 | f1)) continue;
    sw_d2 pv = *reinterpret_cast<const sw_d2*>(p + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < cnt) {      // (uniform)
        const sw_d2 sv = *reinterpret_cast<const sw_d2*>(sp[j] + i);
        if (f0) pv.x = pv.x + sv.x * al[j];
        if (f1) pv.y = pv.y + sv.y * al[j];
      }
    }
    *reinterpret_cast<sw_d2*>(p + i) = pv;
  }
}

// (of the last multi-kernel solve: its ring and scalars stand until the next solve starts)
int eu_launch_finish_p(euler_sim* S) {
  if (S->s_ring_n <= 0) return EULER_OK;
  SRing ring;
  for (int k = 0; k < 8; ++k) ring.s[k] = S->s_ring[k < S->s_ring_n ? k : 0];
  ring.n = S->s_ring_n; ring.steps = S->s_ring_n;
  LAUNCH(S, KC_UPDATE_PR, k_finish_p, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), S->p + S->e_lo, S->cellmask + S->e_lo, S->e_lo, S->e_cnt, S->sc, ring);
  return EULER_OK;
}

// ==========================================================================================
// host-side launch helpers
static SweepArgs make_sweep_args(euler_sim* S, int op, int force) {
  SweepArgs a;
  a.g = S->geom;
  a.mask = S->cellmask; a.fbits_fwd = S->fbits_fwd; a.fbits_bwd = S->fbits_bwd; a.fb_stride = S->fb_stride; a.pre = S->precon;
  a.in = op == SW_FORWARD ? S->r : S->q;
  a.out = op == SW_FORWARD ? S->q : S->z;
  a.granules = S->granules; a.gran_stride = S->gran_stride; a.ticket = S->ticket;
  a.xg_in = nullptr; a.xg_out = nullptr;
  a.ranges = S->band_ranges;
  a.band_lo = S->band_lo; a.nb_local = S->band_hi - S->band_lo; a.couple = S->has_comm && S->couple;
  a.ticket_base = S->ticket_base; a.epoch = S->epoch;
  a.sc = S->sc; a.force = force; a.error = &S->ms->error;
  a.timeline = S->sweep_timeline;
  a.tile_w = eu_is_tile(S) ? S->tile_w : 0;
  a.fin_qq = -1; a.qq_partial = S->partial; a.qq_counter = S->red_counter; a.sc_w = S->sc;
  return a;
}

// ---- active ranges of the bands (per solve) ---------------------------------------------------
// For each 64-row band: the first / last record t = x + lane that holds a fluid cell, turned into
// 32-step aligned block ranges of the forward (step = t) and backward (step = T-1-t) sweeps; the
// upper end leaves at least one all-non-fluid step inside the range (see k_sweep_skew).  Computed
// from the row-major count grid, which every rank holds in full.
__global__ __launch_bounds__(1024) void k_band_ranges(const uint8_t* __restrict__ count, int X, int Y, int T, int4* __restrict__ ranges, int band0) {
  __shared__ int s_lo, s_hi;
  const int band = band0 + blockIdx.x;
  if (threadIdx.x == 0) { s_lo = 0x7fffffff; s_hi = -1; }
  __syncthreads();
  int lo = 0x7fffffff, hi = -1;
  const int rows = Y - band * 64 < 64 ? Y - band * 64 : 64;
  for (int x = threadIdx.x; x < X; x += 1024) {
    for (int l0 = 0; l0 < rows; l0 += 8) {          // 8 independent loads in flight per thread
      uint8_t c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = l0 + k < rows ? count[(size_t)(band * 64 + l0 + k) * X + x] : (uint8_t)0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c[k]) { const int t = x + l0 + k; lo = t < lo ? t : lo; hi = t > hi ? t : hi; }
    }
  }
  if (hi >= 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
  __syncthreads();
  if (threadIdx.x == 0) {
    int4 r = make_int4(0, 0, 0, 0);
    if (s_hi >= 0) {
      const int f0 = s_lo / 32 * 32, f1 = (s_hi + 2 + 31) / 32 * 32;                       // forward steps [f0, f1): whole groups of 4 blocks
      const int b0 = (T - 1 - s_hi) / 32 * 32, b1 = (T - 1 - s_lo + 2 + 31) / 32 * 32;     // backward steps [b0, b1): whole groups of 4 blocks
      r = make_int4(f0 / 8, f1 / 8, b0 / 8, b1 / 8);
    }
    ranges[band] = r;
  }
}
// flags of the sweeps, 8 steps to a dword: word (band, g, lane) bit j = fluid flag of the lane's cell in step 8g + j
// of the forward sweep (record 8g + j) / of the backward sweep (record T-1 - 8g - j); 0 outside [0, T).  The backward word
// also carries the cell's CM_RIGHT (bits 8-15) and CM_UP (bits 16-23) flags
__global__ __launch_bounds__(256) void k_pack_fbits(const uint8_t* __restrict__ cellmask, SkewGeom g, unsigned int* __restrict__ fwd,
                                                    unsigned int* __restrict__ bwd, int fb_stride, int band_lo, int nb_local) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)nb_local * fb_stride * 64) return;
  const int lane = (int)(i & 63);
  const int gi = (int)((i >> 6) % fb_stride), band = band_lo + (int)((i >> 6) / fb_stride);
  const uint8_t* base = cellmask + (size_t)band * g.TS * 64 + 2 * lane;   // paired records: (t & ~1) * 64 + 2 * lane + (t & 1)
  unsigned int wf = 0, wb = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int tf = 8 * gi + j, tb = g.T - 1 - 8 * gi - j;
    if (tf < g.T && (base[(size_t)(tf & ~1) * 64 + (tf & 1)] & CM_FLUID)) wf |= 1u << j;
    if (tb >= 0) {      // backward: the cell's fluid flag and the two neighbour flags its coefficients are built from
      const unsigned int m = base[(size_t)(tb & ~1) * 64 + (tb & 1)];
      if (m & CM_FLUID) wb |= 1u << j;
      if (m & CM_RIGHT) wb |= 1u << (8 + j);
      if (m & CM_UP) wb |= 1u << (16 + j);
    }
  }
  const size_t o = ((size_t)band * fb_stride + gi) * 64 + lane;
  fwd[o] = wf; bwd[o] = wb;
}
int eu_launch_band_ranges(euler_sim* S) {
  if (eu_is_tile(S)) return EULER_OK;   // no band pipeline, no packed flags: k_precond_tile reads the cell mask
  // a rank needs the ranges of its own bands and of the band before / after its slab (the hand-off windows)
  const bool nbrs = S->has_comm && S->couple && !S->slab_on;   // (a row-slab handle does not hold the neighbours' count rows)
  const int b0 = nbrs && S->band_lo > 0 ? S->band_lo - 1 : S->band_lo, b1 = nbrs && S->band_hi < S->geom.nbands ? S->band_hi + 1 : S->band_hi;
  LAUNCH(S, KC_BUILD_SYSTEM, k_band_ranges, dim3(b1 - b0), dim3(1024), S->count, S->X, S->Y, S->geom.T, S->band_ranges, b0);
  const int nbl = S->band_hi - S->band_lo;
  const size_t n = (size_t)nbl * S->fb_stride * 64;
  LAUNCH(S, KC_BUILD_SYSTEM, k_pack_fbits, dim3((unsigned)((n + 255) / 256)), dim3(256), S->cellmask, S->geom, S->fbits_fwd, S->fbits_bwd,
         S->fb_stride, S->band_lo, nbl);
  return EULER_OK;
}

// ---- multi-rank helpers ---------------------------------------------------------------------
__global__ void k_scalar_epilogue(PcgScalars* sc, int op, int force) {   // after the all-reduce of comm_val
  if (!force && pcg_idle(sc)) return;
  pcg_scalar_step(sc, op, sc->comm_val);
}
__global__ void k_nonzero_to_comm(PcgScalars* sc) { sc->comm_val = sc->nonzero ? 1.0 : 0.0; }
__global__ void k_nonzero_from_comm(PcgScalars* sc) { sc->nonzero = sc->comm_val != 0.0; }

// one grid row (band, lane) of a skewed array <-> a contiguous buffer of X doubles
__global__ __launch_bounds__(256) void k_pack_row(const double* __restrict__ skew, double* __restrict__ row, SkewGeom g, int band, int lane) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < g.X) row[x] = skew[skew_index(g, x, band * 64 + lane)];
}
__global__ __launch_bounds__(256) void k_unpack_row(double* __restrict__ skew, const double* __restrict__ row, SkewGeom g, int band, int lane) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < g.X) skew[skew_index(g, x, band * 64 + lane)] = row[x];
}

// the edge rows of two skewed vectors <-> the neighbour buffers, one launch: blockIdx.y = 2 * side + vector.
// PACK: lowest own row (band_lo, lane 0) -> lo buffer, highest (band_hi - 1, lane 63) -> hi buffer; else the rows that arrived
// land in the adjacent bands' storage, (band_lo - 1, lane 63) and (band_hi, lane 0).
template <bool PACK>
__global__ __launch_bounds__(256) void k_halo_rows2(double* a, double* b, double* lo, double* hi, SkewGeom g, int band_lo, int band_hi,
                                                    int has_lo, int has_hi) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= g.X) return;
  const int side = blockIdx.y >> 1, vec = blockIdx.y & 1;
  if (side == 0 ? !has_lo : !has_hi) return;
  double* arr = vec ? b : a;
  double* buf = (side ? hi : lo) + (size_t)vec * g.X;
  const int y = PACK ? (side ? 64 * band_hi - 1 : 64 * band_lo) : (side ? 64 * band_hi : 64 * band_lo - 1);
  double* e = arr + skew_index(g, x, y);
  if (PACK) buf[x] = *e; else *e = buf[x];
}

#define COMM_CALL(expr) do { if ((expr) != 0) { eu_set_error("communicator callback failed: %s", #expr); return EULER_ECOMM; } } while (0)

// ---- "ghost rows": the default with several ranks in tile-local mode (no mailboxes) ------------------------------------------
// A PCG iteration has TWO exchange points, each ONE call of euler_comm_ops.exchange (RCCL: one group of sends / receives):
//   G1  behind k_precond_tile:  the slab's edge rows of z to the two neighbours + {max |r|, dot(z,r)} of every rank to every rank
//   G2  behind k_search_apply:  the partial of dot(s, A s) of every rank to every rank
// and every rank folds what arrived in rank order (identical bits everywhere: alpha, beta and `done` stay in step without a
// broadcast).  Against round 2 (halo(z, s) - all-reduce(alpha) - all-gather{max |r|, dot}: three exchanges, two pack / unpack
// launches): the edge rows leave k_precond_tile as compact rows and are read by k_search_apply as compact rows (no pack, no
// unpack), and s does not travel at all - a rank keeps its ghost rows of s current by itself (k_search_apply SLAB 2).
static inline bool tile_fused(const euler_sim* S);
static inline bool ghost_mode(const euler_sim* S) { return S->has_comm && !S->p2p_on && tile_fused(S); }
enum { XR_ZSEND_LO = 0, XR_ZSEND_HI, XR_ZRECV_LO, XR_ZRECV_HI, XR_GS_LO0, XR_GS_LO1, XR_GS_HI0, XR_GS_HI1 };
static inline double* xrow(const euler_sim* S, int k) { return S->xrows + (size_t)k * S->xrow_len; }
static int comm_exchange(euler_sim* S, double* send_lo, double* send_hi, double* recv_lo, double* recv_hi, int count, double* small, int nsmall) {
  if (S->bulk.exchange) {
    COMM_CALL(S->bulk.exchange(S->bulk.ctx, send_lo, send_hi, recv_lo, recv_hi, count, small, nsmall));
    return EULER_OK;
  }
  // a communicator without the fused operation: the same traffic as a neighbour exchange and an all-gather
  if (count > 0) COMM_CALL(S->bulk.halo(S->bulk.ctx, send_lo, send_hi, recv_lo, recv_hi, count));
  if (nsmall > 0) {
    int64_t off[64], cnt[64];
    for (int r = 0; r < S->bulk.nranks && r < 64; ++r) { off[r] = (int64_t)8 * nsmall * r; cnt[r] = (int64_t)8 * nsmall; }
    COMM_CALL(S->bulk.allgather(S->bulk.ctx, small, off, cnt));
  }
  return EULER_OK;
}
__global__ void k_alpha_fold(PcgScalars* sc, const double* __restrict__ vals, int R, int force) {   // G2: dot(s, A s) = sum over ranks, in rank order
  if (!force && pcg_idle(sc)) return;
  double v = 0.0;
  for (int r = 0; r < R; ++r) v += vals[r];
  pcg_scalar_step(sc, FIN_ALPHA, v);
}

static int comm_allreduce_scalar(euler_sim* S, int is_max) {
  COMM_CALL(S->comm.allreduce(S->comm.ctx, &S->sc->comm_val, 1, is_max));
  return EULER_OK;
}
// rank-local reduction result (left in comm_val by FIN_TO_COMM) -> all-reduce -> scalar epilogue
static int comm_finish(euler_sim* S, int fin_op, int is_max, int force) {
  if (S->p2p_on) return EULER_OK;   // the reduction's last block already exchanged the partials and applied the epilogue
  if (fin_op == FIN_ALPHA && ghost_mode(S)) {      // G2
    int rc = comm_exchange(S, nullptr, nullptr, nullptr, nullptr, 0, S->alpha_buf, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_alpha_fold, dim3(1), dim3(1), 0, S->stream, S->sc, S->alpha_buf, S->comm.nranks, force);
    return EULER_OK;
  }
  int rc = comm_allreduce_scalar(S, is_max);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scalar_epilogue, dim3(1), dim3(1), 0, S->stream, S->sc, fin_op, force);
  return EULER_OK;
}
static inline int fin_or_comm(const euler_sim* S, int fin_op) {
  return S->has_comm ? (S->p2p_on ? (fin_op | FIN_VIA_P2P) : (int)FIN_TO_COMM) : fin_op;
}

// ghost rows of the search vector s for apply_a: my lowest row goes to rank-1, my highest to rank+1
static int comm_halo_s(euler_sim* S) {
  if (S->p2p_on) return eu_p2p_halo_skewed(S, S->s);   // one launch: rows leave from and land in the skewed array
  const int X = S->X, nbk = (X + 255) / 256;
  double *send_lo = S->halo_buf, *send_hi = S->halo_buf + X, *recv_lo = S->halo_buf + 2 * X, *recv_hi = S->halo_buf + 3 * X;
  const bool has_lo = S->band_lo > 0, has_hi = S->band_hi < S->geom.nbands;
  if (has_lo) hipLaunchKernelGGL(k_pack_row, dim3(nbk), dim3(256), 0, S->stream, S->s, send_lo, S->geom, S->band_lo, 0);
  if (has_hi) hipLaunchKernelGGL(k_pack_row, dim3(nbk), dim3(256), 0, S->stream, S->s, send_hi, S->geom, S->band_hi - 1, 63);
  COMM_CALL(S->comm.halo(S->comm.ctx, send_lo, send_hi, recv_lo, recv_hi, X));
  if (has_lo) hipLaunchKernelGGL(k_unpack_row, dim3(nbk), dim3(256), 0, S->stream, S->s, recv_lo, S->geom, S->band_lo - 1, 63);
  if (has_hi) hipLaunchKernelGGL(k_unpack_row, dim3(nbk), dim3(256), 0, S->stream, S->s, recv_hi, S->geom, S->band_hi, 0);
  return EULER_OK;
}

template <int OP>
static int launch_sweep(euler_sim* S, int cls, int force, int fin_qq = -1) {
  if (S->cfg.sweep_mode != EULER_SWEEP_SIMPLE) {
    constexpr bool BWD = OP == SW_BACKWARD;
    const int nb = S->geom.nbands, nbl = S->band_hi - S->band_lo;
    const bool chain = S->has_comm && S->couple;
    const int64_t row_bytes = (int64_t)S->gran_stride * 2 * 8;
    // global pipeline positions of my first / last band in sweep order
    const int g_first = BWD ? nb - S->band_hi : S->band_lo, g_last = g_first + nbl - 1;
    const int r = S->comm.rank, prev_rank = BWD ? r + 1 : r - 1, next_rank = BWD ? r - 1 : r + 1;
    S->epoch += 1;
    if (chain && S->p2p_on) {
      // exact coupling over the mailboxes: all slabs launch at once and the band pipeline runs on across the GPUs -
      // the previous slab's last band announces straight into this rank's mailbox while this kernel is running
      SweepArgs a = make_sweep_args(S, OP, force);
      eu_p2p_xgran(S, BWD ? 1 : 0, &a.xg_in, &a.xg_out);
      LAUNCH(S, cls, (k_sweep_skew<OP, true>), dim3(nbl), dim3(192), a);
      S->ticket_base += (unsigned)nbl;
      return EULER_OK;
    }
    if (chain && g_first > 0)   // the edge row of the band before mine arrives from the previous slab
      COMM_CALL(S->comm.chain(S->comm.ctx, S->granules + (size_t)(g_first - 1) * S->gran_stride * 2, row_bytes, prev_rank, r));
    SweepArgs a = make_sweep_args(S, OP, force);
    a.fin_qq = fin_qq;
    LAUNCH(S, cls, k_sweep_skew<OP>, dim3(nbl), dim3(192), a);
    S->ticket_base += (unsigned)nbl;
    if (chain && g_last + 1 < nb)
      COMM_CALL(S->comm.chain(S->comm.ctx, S->granules + (size_t)g_last * S->gran_stride * 2, row_bytes, r, next_rank));
  } else {
    SweepArgs a = make_sweep_args(S, OP, force);
    LAUNCH(S, cls, k_sweep_simple<OP>, dim3(1), dim3(1024), a);
  }
  return EULER_OK;
}

// element range of this rank (the whole array without a communicator)
#define LOC(ptr) ((ptr) + S->e_lo)
static inline int loc_red_blocks(const euler_sim* S) { return (int)eu_blocks(S->e_cnt, EU_RED_ELEMS, 2048); }

static int launch_dot(euler_sim* S, const double* a, const double* b, int fin_op, int force) {
  if (S->cfg.dot_mode == EULER_DOT_SEQUENTIAL && !S->has_comm) {
    LAUNCH(S, KC_DOT, k_dot_sequential, dim3(1), dim3(256), a, b, S->cellmask, S->geom, S->sc, fin_op, force);
  } else {
    const int edges = (a == S->z && S->has_comm && eu_p2p_has_neighbour_arrays(S)) ? ((S->band_lo > 0 ? 1 : 0) | (S->band_hi < S->geom.nbands ? 2 : 0)) : 0;
    if (edges)
      LAUNCH(S, KC_DOT, k_dot_partial<true>, dim3(loc_red_blocks(S)), dim3(RED_THREADS), LOC(a), LOC(b), LOC(S->cellmask), S->e_cnt,
             S->partial, S->sc, force, S->red_counter, fin_or_comm(S, fin_op), S->geom.TS, S->band_hi - S->band_lo, edges);
    else
      LAUNCH(S, KC_DOT, k_dot_partial<false>, dim3(loc_red_blocks(S)), dim3(RED_THREADS), LOC(a), LOC(b), LOC(S->cellmask), S->e_cnt,
             S->partial, S->sc, force, S->red_counter, fin_or_comm(S, fin_op), 0, 0, 0);
    if (S->has_comm) return comm_finish(S, fin_op, 0, force);
  }
  return EULER_OK;
}

// tile-local IC(0) in its production form: everything between two apply_a passes in one kernel (k_precond_tile)
static inline bool tile_fused(const euler_sim* S) { return eu_is_tile(S) && S->cfg.sweep_mode != EULER_SWEEP_SIMPLE; }
// A s' never goes to memory: k_search_apply leaves it out and the r update (k_precond_tile<16, true>) forms it again from s' (the same bits).
// Not where somebody else reads the array - the sequential replay of dot(s, A s) - nor where the rows across a slab boundary are not at hand:
// the mailbox configurations (comm_p2p.hip) and the reference's IC(0) on several ranks.  EULER_OPT_TILE_STORE_AS restores the stored form.
static inline int sa_run(const euler_sim* S);
static inline bool tile_recompute(const euler_sim* S) {
  if (S->opt[EULER_OPT_TILE_STORE_AS] || sa_run(S) != 8) return false;
  if (!tile_fused(S)) return !S->has_comm && S->cfg.dot_mode != EULER_DOT_SEQUENTIAL;      // (the r update of the other modes is this kernel's first half)
  if (S->tile_w != 16) return false;
  if (S->has_comm) return ghost_mode(S);
  return S->cfg.dot_mode != EULER_DOT_SEQUENTIAL;
}
// max |r| and dot(z,r) of all ranks after ONE exchange ("fuse the latter two into one ... message pair"): every rank
// folds the gathered pairs in rank order - identical bits everywhere - and applies the two scalar epilogues
__global__ void k_pair_fold(PcgScalars* sc, const double* __restrict__ pairs, int stride, int R, int rupd, int fin_dot, int force) {
  if (!force && pcg_idle(sc)) return;
  double vmax = 0.0, vsum = 0.0;
  for (int r = 0; r < R; ++r) { vmax = pairs[(size_t)stride * r] > vmax ? pairs[(size_t)stride * r] : vmax; vsum += pairs[(size_t)stride * r + 1]; }
  if (rupd) pcg_scalar_step(sc, FIN_RNORM, vmax);
  if (fin_dot >= 0 && !(rupd && sc->done)) pcg_scalar_step(sc, fin_dot, vsum);
}
static TileArgs make_tile_args(euler_sim* S, int force) {
  TileArgs a;
  a.g = S->geom; a.mask = S->cellmask; a.pre = S->precon; a.r = S->r; a.as = S->tile_as_override ? S->tile_as_override : S->q; a.z = S->z;
  a.band_lo = S->band_lo; a.nb_local = S->band_hi - S->band_lo;
  a.rupd = 0; a.sweeps = 1; a.fin_dot = -1;
  a.via = S->has_comm ? (S->p2p_on ? (int)FIN_VIA_P2P : (int)FIN_TO_COMM) : 0;
  a.part_max = S->partial; a.part_dot = S->partial2; a.counter = S->red_counter; a.sc = S->sc; a.force = force; a.alpha_arg = 0.0;
  a.pair_slot = S->pair_buf + 2 * (S->has_comm ? S->comm.rank : 0);
  a.list = !force ? S->chunk_list : nullptr;      // (tiles of 16 records only; forced single operations may run on masks no solve has listed)
  a.table = S->tile_table;
  a.zsend_lo = a.zsend_hi = nullptr; a.edge_lo = a.edge_hi = -1;
  // descending: k_search_apply walks the chunks upwards, so this pass starts on what the Infinity Cache still holds of it - and ends
  // where the next k_search_apply starts.  8192^2: 548 -> 536 us per iteration (EULER_OPT_TILE_REVERSE 0 restores the ascending order)
  a.reverse = S->opt[EULER_OPT_TILE_REVERSE] != 0;
  a.cpart = nullptr; a.cshift = 0; a.cmode = 0; a.cnx = a.cny = 0;
  a.gs_lo = a.gs_hi = nullptr;
  if (ghost_mode(S)) {
    if (S->band_lo > 0) { a.zsend_lo = xrow(S, XR_ZSEND_LO); a.edge_lo = S->band_lo; }
    if (S->band_hi < S->geom.nbands) { a.zsend_hi = xrow(S, XR_ZSEND_HI); a.edge_hi = S->band_hi - 1; }
  }
  return a;
}
static inline unsigned tile_blocks(const euler_sim* S) {
  const size_t tiles = (size_t)(S->band_hi - S->band_lo) * (S->geom.T / S->tile_w);
  return eu_blocks(tiles, PT_THREADS / 64, 2048);
}
static int launch_factor_tile(euler_sim* S, int force) {
  const TileArgs a = make_tile_args(S, force);
  switch (S->tile_w) {
    case 8: LAUNCH(S, KC_PRECON_FACTOR, k_factor_tile<8>, dim3(tile_blocks(S)), dim3(PT_THREADS), a); break;
    case 32: LAUNCH(S, KC_PRECON_FACTOR, k_factor_tile<32>, dim3(tile_blocks(S)), dim3(PT_THREADS), a); break;
    default: LAUNCH(S, KC_PRECON_FACTOR, k_factor_tile<16>, dim3(tile_blocks(S)), dim3(PT_THREADS), a); break;
  }
  return EULER_OK;
}
// [r -= alpha A s, max |r|,] z = M^-1 r, dot(z, r) with its scalar epilogue fin_dot (FIN_SIGMA_INIT / FIN_BETA / FIN_STORE_ONLY)
// r_only: the kernel's first half alone - r -= alpha A s and max |r| with its epilogue (`done`) - over 16-record chunks whatever the
// handle's tile width: how EVERY non-tile configuration (the reference's IC(0), Jacobi) updates r since round 3 (p rides in k_search_apply)
static int launch_precond_tile(euler_sim* S, int rupd, int sweeps, int fin_dot, int force, double alpha, bool r_only = false) {
  const bool seq = S->cfg.dot_mode == EULER_DOT_SEQUENTIAL && !S->has_comm;
  TileArgs a = make_tile_args(S, force);
  a.rupd = rupd; a.sweeps = sweeps; a.fin_dot = (seq || !sweeps) ? -1 : fin_dot; a.alpha_arg = alpha;
  // two-level mode: the tile pass also leaves P^T r per tile and only STORES its share of dot(z, r); k_coarse_solve adds the coarse
  // share and applies the epilogue
  const bool two_level = eu_is_two_level(S) && sweeps && !r_only && !force && a.list != nullptr;
  const int fin_real = a.fin_dot;
  if (two_level) {
    const bool mg = eu_is_mg(S);
    a.cpart = mg ? S->mg_part : S->cc_part; a.cshift = mg ? 0 : S->coarse_shift;
    a.cmode = mg ? 2 : 1; a.cnx = mg ? S->mg_nx[0] : 0; a.cny = mg ? S->mg_ny[0] : 0;
    if (a.fin_dot >= 0) a.fin_dot = FIN_STORE_ONLY;
  }
  // row slabs + coarse correction: the pair and this rank's rows of the level-0 right-hand side travel in ONE slot of ONE all-gather inside the G1 exchange
  double* xsmall = S->pair_buf;
  int nsmall = 2;
  bool split = false;      // multilevel mode: the cycle split by rows (k_mg.hip) - the edge rows of z go straight into its messages
  if (two_level && a.via == FIN_TO_COMM) {
    split = eu_is_mg(S) && ghost_mode(S) && eu_mg_split(S);
    nsmall = eu_coarse_comm_slots(S);
    if (nsmall < 0) return EULER_ENOMEM;
    xsmall = S->mg_xbuf;
    a.pair_slot = xsmall + (size_t)S->comm.rank * nsmall;
    if (split) {
      if (a.zsend_lo) a.zsend_lo = eu_mg_split_msg(S, 0);
      if (a.zsend_hi) a.zsend_hi = eu_mg_split_msg(S, 1);
    }
  }
  const int w = r_only ? 16 : S->tile_w, cls = r_only ? KC_UPDATE_PR : KC_PRECOND_TILE;
  const unsigned nblk = eu_blocks((size_t)(S->band_hi - S->band_lo) * (S->geom.T / w), PT_THREADS / 64, 2048);
  // (r_only: the other modes' r update - A s' would sit in q behind k_search_apply, in z behind the solve's first k_apply_a, which stores it)
  const bool recomp = rupd && !force && tile_recompute(S) && (r_only ? S->tile_as_override == S->q : !S->tile_as_override);
  if (recomp) {      // A s' is formed from the search direction (S->s behind k_search_apply's swap, or s_0 behind k_apply_a) and, on row slabs, the ghost rows of s' of its generation
    a.as = S->s;
    if (ghost_mode(S)) {
      if (S->band_lo > 0) a.gs_lo = xrow(S, XR_GS_LO0 + S->gs_cur);
      if (S->band_hi < S->geom.nbands) a.gs_hi = xrow(S, XR_GS_HI0 + S->gs_cur);
    }
  }
  switch (w) {
    case 8: LAUNCH(S, cls, k_precond_tile<8>, dim3(nblk), dim3(PT_THREADS), a); break;
    case 32: LAUNCH(S, cls, k_precond_tile<32>, dim3(nblk), dim3(PT_THREADS), a); break;
    default:
      if (a.cmode == 2) { if (recomp) LAUNCH(S, cls, (k_precond_tile<16, true, 2>), dim3(nblk), dim3(PT_THREADS), a); else LAUNCH(S, cls, (k_precond_tile<16, false, 2>), dim3(nblk), dim3(PT_THREADS), a); }
      else if (a.cmode == 1) { if (recomp) LAUNCH(S, cls, (k_precond_tile<16, true, 1>), dim3(nblk), dim3(PT_THREADS), a); else LAUNCH(S, cls, (k_precond_tile<16, false, 1>), dim3(nblk), dim3(PT_THREADS), a); }
      else if (recomp) LAUNCH(S, cls, (k_precond_tile<16, true>), dim3(nblk), dim3(PT_THREADS), a);
      else LAUNCH(S, cls, k_precond_tile<16>, dim3(nblk), dim3(PT_THREADS), a);
      break;
  }
  if (two_level && a.via != FIN_TO_COMM) { int rc = eu_launch_coarse_solve(S, fin_real, force); if (rc) return rc; }
  if (seq && sweeps && fin_dot >= 0)   // the reference's row-major dot(z, r); a no-op once max |r| <= tol
    LAUNCH(S, KC_DOT, k_dot_sequential, dim3(1), dim3(256), S->z, S->r, S->cellmask, S->geom, S->sc, fin_dot, force);
  if (a.via == FIN_TO_COMM) {          // no mailboxes: G1 - both results (and, in the solve, the edge rows of the new z) in ONE exchange, then the epilogues
    const int R = S->comm.nranks;
    const bool rows = ghost_mode(S) && !force;
    int rc = !two_level ? EULER_OK : split ? eu_mg_split_pre(S) : eu_launch_coarse_pre(S, force);
    if (rc) return rc;
    if (split) rc = comm_exchange(S, eu_mg_split_msg(S, 0), eu_mg_split_msg(S, 1), eu_mg_split_msg(S, 2), eu_mg_split_msg(S, 3), eu_mg_split_count(S), xsmall, nsmall);
    else rc = comm_exchange(S, xrow(S, XR_ZSEND_LO), xrow(S, XR_ZSEND_HI), xrow(S, XR_ZRECV_LO), xrow(S, XR_ZRECV_HI), rows ? S->X : 0, xsmall, nsmall);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pair_fold, dim3(1), dim3(1), 0, S->stream, S->sc, xsmall, nsmall, R, rupd, a.fin_dot, force);
    if (split) {      // a third exchange point: the ranks' shares of the correction's dot product (and of the gauge sums) live on their own rows
      if ((rc = eu_mg_split_mid(S, fin_real, force, xrow(S, XR_ZRECV_LO), xrow(S, XR_ZRECV_HI)))) return rc;
      if ((rc = comm_exchange(S, nullptr, nullptr, nullptr, nullptr, 0, eu_mg_split_gc(S), 1 + MG_NULL_MAX))) return rc;
      return eu_mg_split_fold(S, fin_real, force);
    }
    // coarse correction on row slabs: the tiles' shares of dot(z, r) are folded (stored, not applied); the V-cycle - its level-0 right-hand
    // side the sum of the ranks' shares, the rest replicated - adds its share and applies the epilogue, the same bits on every rank
    if (two_level) { rc = eu_launch_coarse_scatter(S); if (rc) return rc; rc = eu_launch_coarse_solve(S, fin_real, force); if (rc) return rc; }
  }
  return EULER_OK;
}

// ghost rows of TWO skewed vectors in one exchange (z and s before the fused search + apply_a pass: the neighbouring
// slabs' edge rows land in the adjacent bands' storage, where k_search_apply's lanes 0 / 63 look for them)
static int comm_halo_two(euler_sim* S, double* a, double* b) {
  if (S->p2p_on) { int rc = eu_p2p_halo_skewed(S, a); return rc ? rc : eu_p2p_halo_skewed(S, b); }
  const int X = S->X, nbk = (X + 255) / 256;
  double *send_lo = S->halo_buf, *send_hi = S->halo_buf + 2 * X, *recv_lo = S->halo_buf + 4 * X, *recv_hi = S->halo_buf + 6 * X;
  const int has_lo = S->band_lo > 0, has_hi = S->band_hi < S->geom.nbands;
  // one launch packs the four rows (2 vectors x lowest / highest own row), one unpacks the four that arrived
  hipLaunchKernelGGL(k_halo_rows2<true>, dim3(nbk, 4), dim3(256), 0, S->stream, a, b, send_lo, send_hi, S->geom, S->band_lo, S->band_hi, has_lo, has_hi);
  COMM_CALL(S->comm.halo(S->comm.ctx, send_lo, send_hi, recv_lo, recv_hi, 2 * X));
  hipLaunchKernelGGL(k_halo_rows2<false>, dim3(nbk, 4), dim3(256), 0, S->stream, a, b, recv_lo, recv_hi, S->geom, S->band_lo, S->band_hi, has_lo, has_hi);
  return EULER_OK;
}

// dot(z, r) rides in the forward sweep as dot(q, q) (SweepArgs::fin_qq): one rank, tree-dot mode, the band schedule
static inline bool dot_rides_in_sweep(const euler_sim* S) {
  return S->cfg.precond == EULER_PRECOND_IC0 && S->cfg.sweep_mode != EULER_SWEEP_SIMPLE && !S->has_comm && S->cfg.dot_mode != EULER_DOT_SEQUENTIAL;
}
static int launch_precondition(euler_sim* S, int force, int fin_dot = -1) {   // z = M^-1 r [and the scalar epilogue of dot(z, r)]
  if (S->cfg.precond == EULER_PRECOND_JACOBI) {
    LAUNCH(S, KC_JACOBI, k_jacobi, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), LOC(S->r), LOC(S->z), LOC(S->cellmask),
           S->e_cnt, S->sc, force);
    return fin_dot >= 0 ? launch_dot(S, S->z, S->r, fin_dot, force) : EULER_OK;
  }
  const bool ride = fin_dot >= 0 && dot_rides_in_sweep(S);
  int rc = launch_sweep<SW_FORWARD>(S, KC_FORWARD_SOLVE, force, ride ? fin_dot : -1);
  if (rc) return rc;
  if ((rc = launch_sweep<SW_BACKWARD>(S, KC_BACKWARD_SOLVE, force))) return rc;
  return fin_dot >= 0 && !ride ? launch_dot(S, S->z, S->r, fin_dot, force) : EULER_OK;
}

static int launch_apply_a_and_alpha(euler_sim* S, int force) {
  const bool seq = S->cfg.dot_mode == EULER_DOT_SEQUENTIAL && !S->has_comm;
  if (S->has_comm && ghost_mode(S) && !force) {
    // the first search direction is z_0, whose edge rows arrived with the exchange behind the solve's first k_precond_tile: they
    // become the ghost rows of s - in the adjacent bands' storage for this one pass (k_apply_a looks there), and generation 0 of
    // the compact rows k_search_apply keeps current from here on
    const int X = S->X, nbk = (X + 255) / 256;
    const bool coarse = eu_is_two_level(S);      // (the first search direction is z_0 + P y then: the ghost rows take the P y of their own cells)
    if (S->band_lo > 0) {
      HIPCHK(hipMemcpyAsync(xrow(S, XR_GS_LO0), xrow(S, XR_ZRECV_LO), (size_t)X * 8, hipMemcpyDeviceToDevice, S->stream));
      if (coarse) { int rc = eu_launch_coarse_add_row(S, xrow(S, XR_GS_LO0), 64 * S->band_lo - 1); if (rc) return rc; }
      hipLaunchKernelGGL(k_unpack_row, dim3(nbk), dim3(256), 0, S->stream, S->s, xrow(S, XR_GS_LO0), S->geom, S->band_lo - 1, 63);
    }
    if (S->band_hi < S->geom.nbands) {
      HIPCHK(hipMemcpyAsync(xrow(S, XR_GS_HI0), xrow(S, XR_ZRECV_HI), (size_t)X * 8, hipMemcpyDeviceToDevice, S->stream));
      if (coarse) { int rc = eu_launch_coarse_add_row(S, xrow(S, XR_GS_HI0), 64 * S->band_hi); if (rc) return rc; }
      hipLaunchKernelGGL(k_unpack_row, dim3(nbk), dim3(256), 0, S->stream, S->s, xrow(S, XR_GS_HI0), S->geom, S->band_hi, 0);
    }
    S->gs_cur = 0;
  } else if (S->has_comm) { int rc = comm_halo_s(S); if (rc) return rc; }
  SkewGeom gl = S->geom;
  gl.S = S->e_cnt;
  double* out = tile_fused(S) ? S->q : S->z;     // tile-local mode: A s always lands in q (k_precond_tile reads it there and writes z)
  LAUNCH(S, KC_APPLY_A, k_apply_a, dim3(loc_red_blocks(S)), dim3(RED_THREADS), LOC(S->s), LOC(out), LOC(S->cellmask), gl,
         S->partial, S->sc, force, S->red_counter, seq ? -1 : fin_or_comm(S, FIN_ALPHA));
  if (seq)
    LAUNCH(S, KC_DOT, k_dot_sequential, dim3(1), dim3(256), out, S->s, S->cellmask, S->geom, S->sc, (int)FIN_ALPHA, force);
  if (S->has_comm) return comm_finish(S, FIN_ALPHA, 0, force);
  return EULER_OK;
}

static inline unsigned sa_blocks(const euler_sim* S, int run) {   // one wave per run of pair-records, at most 2048 blocks (the partials)
  const size_t runs = (size_t)(S->band_hi - S->band_lo) * ((S->geom.TS / 2 + run - 1) / run);
  return eu_blocks(runs, SA_THREADS / 64, 2048);
}
static inline int sa_run(const euler_sim* S) {   // short runs while long ones would leave CUs without a wave
  if (S->opt[EULER_OPT_SA_RUN] != 8 && !S->has_comm && !eu_is_two_level(S)) return (int)S->opt[EULER_OPT_SA_RUN];      // (experiments: 16, 32)
  // measured (same box, tile-local mode): 8192^2 - 8: 346 us, 16: 358, 32: 376 (113 / 134 / 185 VGPRs: occupancy beats the window's
  // two extra pair loads per run, which hit L2); 16384^2, scanning every run's masks - 8: 1412 us, 32: 1389; with the list of active
  // chunks (runs of 8 only) - 8: 1177 us, 32: 1401
  return 8;
}
// p += alpha s, N iterations at a time (k_search_apply PMODE N): 8 - the search directions turn through a ring of eight arrays, six of them allocated when the
// first solve needs them (8192^2: 431 -> 421 us per iteration against 4, 473 -> 450 for 4 against 2); 2 (rounds 3-4: s and s2 alone) with the mailboxes, whose
// peers map exactly those two, and with the experimental run lengths.  EULER_OPT_P_STEPS 2 / 4 select the shorter rings (the same bits: the fmadds of main.c:753
// are applied in their order either way).
static inline int p_steps(const euler_sim* S) {
  if (S->p2p_on || sa_run(S) != 8) return 2;
  return (int)S->opt[EULER_OPT_P_STEPS];
}
// the ring of this solve: [0], [1] = s, s2 as the solve finds them, then the extra arrays (zeroed once; like s and s2 they are only ever written on fluid cells)
static int ring_begin(euler_sim* S) {
  const int n = p_steps(S);
  if (!S->s_base[0]) { S->s_base[0] = S->s; S->s_base[1] = S->s2; }      // the two arrays of the handle (whichever way round the solves so far left them)
  if (n > 2) { S->s = S->s_base[0]; S->s2 = S->s_base[1]; }             // (the last solve may have ended on one of the extra arrays)
  else if (S->s != S->s_base[0] && S->s != S->s_base[1]) { S->s = S->s_base[0]; S->s2 = S->s_base[1]; }
  else if (S->s2 != S->s_base[0] && S->s2 != S->s_base[1]) S->s2 = S->s == S->s_base[0] ? S->s_base[1] : S->s_base[0];
  S->s_ring[0] = S->s; S->s_ring[1] = S->s2;
  for (int k = 2; k < n; ++k) {
    if (!S->s_ring_alloc[k]) {
      const size_t elems = S->Sw + EU_SKEW_SLACK;
      if (hipMalloc(&S->s_ring_alloc[k], elems * sizeof(double)) != hipSuccess) {
        // no room for the ring (13 GB at 16384^2 on one GPU): the solve goes on with the handle's own two arrays - the same bits (tests/test_gpu_tile_precond.py: the
        // forms test), p is updated every second iteration instead of every eighth - rather than fail a substep whose marker stage has already run
        (void)hipGetLastError();
        S->s_ring_alloc[k] = nullptr;
        for (int j = 2; j < k; ++j) if (S->s_ring_alloc[j]) { (void)hipFree(S->s_ring_alloc[j]); S->s_ring_alloc[j] = nullptr; S->hbm_bytes -= elems * sizeof(double); }
        S->opt[EULER_OPT_P_STEPS] = 2;
        return ring_begin(S);
      }
      S->hbm_bytes += elems * sizeof(double);      // (euler_hbm_bytes: what the handle holds)
      HIPCHK(hipMemsetAsync(S->s_ring_alloc[k], 0, elems * sizeof(double), S->stream));
    }
    S->s_ring[k] = static_cast<double*>(S->s_ring_alloc[k]) + EU_SKEW_SLACK - S->skew_off;
  }
  S->s_ring_n = n;
  S->s_launched = 0;
  return EULER_OK;
}
// EULER_F_PCG_S: the search direction of the last iteration that RAN (launches behind convergence return at once but still turn the ring)
double* eu_current_search_direction(euler_sim* S) {
  if (S->s_ring_n < 2 || S->s_launched <= 0) return S->s;
  (void)hipStreamSynchronize(S->stream);      // sc_host: copied at the end of the solve
  const int ran = S->sc_host->nonzero ? S->sc_host->iters : 0;
  if (ran <= 0 || ran > S->s_launched) return S->s;
  return S->s_ring[(ran - 1) % S->s_ring_n];
}
// iterations >= 1 of a single-GPU solve: s' = z + beta s and A s' in one launch; returns with S->s = s' and A s' in S->q
static int launch_search_apply_and_alpha(euler_sim* S, int it) {
  const bool seq = S->cfg.dot_mode == EULER_DOT_SEQUENTIAL && !S->has_comm;
  SlabNeighbours nbr = {nullptr, nullptr, nullptr, nullptr, S->band_hi - S->band_lo, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const bool direct = S->has_comm && eu_p2p_has_neighbour_arrays(S);   // (opt-in) read the neighbouring slabs' z and s where they live
  const bool ghost = ghost_mode(S);
  if (ghost) {      // the neighbours' z rows came with G1; the ghost rows of s are kept here (two generations, like s / s2)
    const int c = S->gs_cur;
    if (S->band_lo > 0) { nbr.zrow_dn = xrow(S, XR_ZRECV_LO); nbr.srow_dn = xrow(S, XR_GS_LO0 + c); nbr.snew_dn = xrow(S, XR_GS_LO0 + (c ^ 1)); }
    if (S->band_hi < S->geom.nbands) { nbr.zrow_up = xrow(S, XR_ZRECV_HI); nbr.srow_up = xrow(S, XR_GS_HI0 + c); nbr.snew_up = xrow(S, XR_GS_HI0 + (c ^ 1)); }
    S->gs_cur = c ^ 1;
  } else if (direct) {   // addressed with this rank's offsets (the arrays are full-size everywhere)
    eu_p2p_neighbour_arrays(S, &nbr.z_dn, &nbr.s_dn, &nbr.z_up, &nbr.s_up);
    if (nbr.z_dn) { nbr.z_dn += S->e_lo; nbr.s_dn += S->e_lo; }
    if (nbr.z_up) { nbr.z_up += S->e_lo; nbr.s_up += S->e_lo; }
  } else if (S->has_comm) {   // the default: one exchange brings the neighbours' edge rows of z and s into the adjacent bands' storage
    int rc = comm_halo_two(S, S->z, S->s);
    if (rc) return rc;
  }
  SkewGeom gl = S->geom;
  gl.S = S->e_cnt;
  // p += alpha s rides along, two iterations' worth on every even iteration (k_search_apply PMODE) - in every configuration since
  // round 3: the parity mode's k_update_pr read and wrote p on every iteration for nothing but this
  const int steps = S->s_ring_n;
  const int pmode = (it >= steps && it % steps == 0) ? steps : 1;
  SaHist hist = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
  for (int j = 1; j + 1 < steps; ++j) hist.s[j - 1] = LOC(S->s_ring[j]);      // PMODE N (it a multiple of N): s_(it-N+j) sits in ring[j]
  const int run = sa_run(S);
  const int fin = S->has_comm ? fin_or_comm(S, FIN_ALPHA) : (seq ? -1 : (int)FIN_ALPHA);
  double* pp = LOC(S->p);
  const bool store = !tile_recompute(S);      // false: A s' is not stored (k_precond_tile<16, true> forms it again)
  const bool mg = eu_is_mg(S);
  const CoarseRef cref = {mg ? S->mg_x : S->cc_y, mg ? 0 : S->coarse_shift, mg ? S->mg_nx[0] : S->coarse_nx, mg ? S->mg_ny[0] : S->coarse_ny, S->band_lo};
#define SA_LAUNCH_CS(SLABF, PM, RUNV, CF, ST)                                                                                                \
  LAUNCH(S, KC_APPLY_A, (k_search_apply<SLABF, PM, RUNV, CF, ST>), dim3(sa_blocks(S, RUNV)), dim3(SA_THREADS), LOC(S->s), LOC(S->z), LOC(S->s2), \
         LOC(S->q), LOC(S->cellmask), gl, S->partial, S->sc, 0, S->red_counter, fin, nbr, pp, S->s2, S->s, (RUNV) == 8 ? S->chunk_list : (const unsigned int*)nullptr, cref, hist)
#define SA_LAUNCH_C(SLABF, PM, RUNV, CF) do { if ((RUNV) == 8 && !store) SA_LAUNCH_CS(SLABF, PM, RUNV, CF, ((RUNV) != 8)); else SA_LAUNCH_CS(SLABF, PM, RUNV, CF, true); } while (0)
#define SA_LAUNCH(SLABF, PM, RUNV) SA_LAUNCH_C(SLABF, PM, RUNV, 0)
#define SA_RUNS(SLABF, PM) do { if (run == 8) SA_LAUNCH(SLABF, PM, 8); else if (run == 16) SA_LAUNCH(SLABF, PM, 16); else SA_LAUNCH(SLABF, PM, 32); } while (0)
  if (eu_is_two_level(S) && tile_fused(S) && ghost) {      // coarse correction on row slabs (multilevel mode): the ghost rows of z get their P y here as well
    if (pmode == 8) SA_LAUNCH_C(2, 8, 8, 2); else if (pmode == 4) SA_LAUNCH_C(2, 4, 8, 2); else if (pmode == 2) SA_LAUNCH_C(2, 2, 8, 2); else SA_LAUNCH_C(2, 1, 8, 2);
  } else if (eu_is_two_level(S) && tile_fused(S)) {      // two-level / multilevel preconditioner on one GPU: runs of 8 (the list), z + P y
    if (mg) { if (pmode == 8) SA_LAUNCH_C(0, 8, 8, 2); else if (pmode == 4) SA_LAUNCH_C(0, 4, 8, 2); else if (pmode == 2) SA_LAUNCH_C(0, 2, 8, 2); else SA_LAUNCH_C(0, 1, 8, 2); }
    else { if (pmode == 8) SA_LAUNCH_C(0, 8, 8, 1); else if (pmode == 4) SA_LAUNCH_C(0, 4, 8, 1); else if (pmode == 2) SA_LAUNCH_C(0, 2, 8, 1); else SA_LAUNCH_C(0, 1, 8, 1); }
  } else if (ghost) {      // (tile-local mode; several ranks: runs of 8)
    if (pmode == 8) SA_LAUNCH(2, 8, 8); else if (pmode == 4) SA_LAUNCH(2, 4, 8); else if (pmode == 2) SA_LAUNCH(2, 2, 8); else SA_LAUNCH(2, 1, 8);
  } else if (direct) {
    if (pmode == 2) { if (run == 8) SA_LAUNCH(1, 2, 8); else SA_LAUNCH(1, 2, 32); }
    else if (pmode == 1) { if (run == 8) SA_LAUNCH(1, 1, 8); else SA_LAUNCH(1, 1, 32); }
    else { if (run == 8) SA_LAUNCH(1, 0, 8); else SA_LAUNCH(1, 0, 32); }
  } else {
    if (pmode == 8) SA_LAUNCH(0, 8, 8); else if (pmode == 4) SA_LAUNCH(0, 4, 8); else if (pmode == 2) SA_RUNS(0, 2); else if (pmode == 1) SA_RUNS(0, 1); else SA_RUNS(0, 0);
  }
#undef SA_RUNS
#undef SA_LAUNCH
#undef SA_LAUNCH_C
#undef SA_LAUNCH_CS
  S->s = S->s_ring[it % steps]; S->s2 = S->s_ring[(it + 1) % steps];      // (two arrays: the swap of rounds 1-4)
  S->s_launched = it + 1;
  if (seq)
    LAUNCH(S, KC_DOT, k_dot_sequential, dim3(1), dim3(256), S->q, S->s, S->cellmask, S->geom, S->sc, (int)FIN_ALPHA, 0);
  if (S->has_comm) return comm_finish(S, FIN_ALPHA, 0, 0);
  return EULER_OK;
}

int eu_launch_build_system(euler_sim* S, float dt);
int eu_launch_velocity_update(euler_sim* S, float dt, int finish);

__global__ void k_pcg_reset(PcgScalars* sc, double tol, int max_iters) {
  sc->sigma = sc->zs = sc->sigma_new = sc->alpha = sc->alpha_prev = sc->beta = sc->rnorm = 0.0;
  sc->tol = tol; sc->nonzero = 0; sc->done = 0; sc->iters = 0; sc->max_iters = max_iters;
}

// project() (main.c:709-806).  With a communicator every launch below covers this rank's bands
// only; the exchange points are: OR of all_zero(r), the band hand-off rows around each sweep
// (exact coupling), the ghost rows of s before apply_a, one scalar all-reduce per reduction, and
// the all-gather of p before the (replicated) velocity update.
int eu_launch_project(euler_sim* S, float dt) {
  S->s_launched = 0;      // (EULER_F_PCG_S: no multi-kernel iteration of this solve has run yet)
  S->p_pending = 0;       // (the assembly writes p afresh: whatever the last solve left unfinished in memory is gone with it)
  S->pcg_fields_resident = 0;
  const PcgScalars prev_solve = *S->sc_host;      // (the previous solve's final scalars: a resident launch that has to be redone must not leave its own in their place)
  // the previous solve's final scalars are in sc_host by now (copied at its end, synced since)
  S->solve_iters[S->solve_seq & 255] = S->sc_host->nonzero ? S->sc_host->iters : -1;
  if (S->solve_seq > 0) { S->res_last_chunks = S->sc_host->n_chunks; S->res_have_last = 1; }      // (the resident solver's guess for this solve, below)
  S->solve_seq += 1;
  // Convergence polls never drain the queue (round 6): the host enqueues a chunk of iterations, a 48-byte copy of the scalars and an event, and then waits for the
  // copy behind the PREVIOUS chunk - the GPU always has the next chunk queued behind the one whose outcome the host is waiting for.  What is queued behind
  // convergence returns at once (every per-iteration kernel reads sc->done first), so a late poll costs launches, not work.  The chunk length follows a
  // prediction: a solve that converged in k iterations is followed by one that needs about k (scenes change slowly) - chunks of pcg_poll_interval up to k - 2,
  // chunks of 2 from there on (rounds 1-5: a blocking poll every pcg_poll_interval iterations for converging solves: ~27 us of idle GPU each, and up to
  // pcg_poll_interval - 1 launches behind convergence that, in the multilevel mode's cycle, did their whole work).
  // (several ranks: nonzero / done / iters are the same on every rank, so all ranks cut the same chunks and stop behind the same one)
  const int pred_iters = (S->sc_host->nonzero && S->sc_host->done && S->sc_host->iters < S->cfg.max_iterations) ? S->sc_host->iters : 0;
  // With the peer-to-peer mailboxes a rank's kernels WAIT inside the kernel for their peers' (comm_p2p.hip).  Ranks that share one GPU (the tests) then need the queue to run
  // dry now and then - a process whose kernels spin while the other's are queued behind them makes a 2 s test take 200 - so converging solves keep rounds 1-5's blocking
  // poll there (solves that ran into the cap polled one chunk late before, and still do).
  const bool drain_poll = S->p2p_on && !(S->sc_host->nonzero && !S->sc_host->done && S->sc_host->iters >= S->cfg.max_iterations);
  S->prof_iter = -1;
  int rc;
  LAUNCH(S, KC_MISC, k_pcg_reset, dim3(1), dim3(1), S->sc, S->cfg.tol, S->cfg.max_iterations);
  eu_launch_build_system(S, dt);
  if (S->has_comm) {   // all_zero(r) over the whole grid
    hipLaunchKernelGGL(k_nonzero_to_comm, dim3(1), dim3(1), 0, S->stream, S->sc);
    if ((rc = comm_allreduce_scalar(S, 1))) return rc;
    hipLaunchKernelGGL(k_nonzero_from_comm, dim3(1), dim3(1), 0, S->stream, S->sc);
  }
  // Small grids (every chunk finds a wave on the chip at once): the whole solve as ONE persistent launch (k_resident.hip).  Two host round trips per solve -
  // the number of active chunks before the launch, its error word behind it - instead of one per eight iterations.
  const bool res_skip = S->res_skip_once != 0;
  S->res_skip_once = 0;
  if (!res_skip && eu_resident_eligible(S)) {
    // Whether this solve's active chunks fit is decided from the PREVIOUS solve's count (scenes change slowly; sc_host holds it: copied at the end of every solve) - no host
    // round trip in front of the launch: the kernel is launched over the device's whole capacity, the workgroups beyond (n_chunks + 3) / 4 leave at once, and a solve that
    // does not fit after all says so itself (error word 2) and is redone below.  The first solve of a handle asks the device.
    const unsigned int cap_chunks = 4u * (unsigned int)eu_resident_capacity(S, S->cfg.pcg_precision == EULER_PCG_F32);
    unsigned int guess = S->res_last_chunks;
    if (!S->res_have_last) {
      HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
      HIPCHK(hipStreamSynchronize(S->stream));
      guess = S->sc_host->n_chunks;
    }
    if (guess + guess / 16 + 8 <= cap_chunks || S->cfg.pcg_precision == EULER_PCG_F32) {
      S->prof_iter = -2;
      if ((rc = eu_launch_resident(S, cap_chunks))) return rc;      // (a no-op on the device when the right-hand side is all zero: main.c:742, p = 0 from the assembly)
      HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
      HIPCHK(hipStreamSynchronize(S->stream));
      S->res_last_chunks = S->sc_host->n_chunks; S->res_have_last = 1;
      const int err = *S->res_err;
      if (err == 0) {
        if (S->sc_host->nonzero) { S->res_solves += 1; S->pcg_fields_resident = 1; }      // (z, s, q never left the registers: EULER_F_PCG_Z / _S / _Q have nothing to show)
        if ((rc = eu_launch_velocity_update(S, dt, 0))) return rc;      // (the resident kernel wrote the final p)
        return EULER_OK;      // (sc_host is current)
      }
      *S->res_err = 0;
      if (S->cfg.pcg_precision == EULER_PCG_F32 && err == 2) { eu_set_error("EULER_PCG_F32: the solve's active chunks do not fit the resident solver"); return EULER_EHIP; }
      // (a time-out on an EULER_PCG_F32 handle: this system and every later one are solved in DOUBLE by the multi-kernel path - the substep's markers have moved already,
      // so failing here would leave the handle between two stages; euler_resident_info reports the fallback)
      if (err != 2) { S->res_fallbacks += 1; S->res_disabled = 1; }      // a wait ran out (the workgroups were not all resident - another process on the device?): the handle keeps to the multi-kernel path from here on
      // (error 2: more active chunks than fit - the scene grew; this system with the multi-kernel path: k_pcg_reset + the assembly again, written whole)
      S->solve_seq -= 1;
      S->lean_ok = 0;
      S->res_skip_once = 1;
      *S->sc_host = prev_solve;      // (solve_iters[] and the look-ahead of the redo are the previous SOLVE's, not the aborted launch's)
      return eu_launch_project(S, dt);
    }
    // (too many active chunks last time: the multi-kernel path below; its final copy of the scalars keeps the count current)
  }
  // if (!all_zero(r)) { ... }: every kernel below is a no-op when sc->nonzero == 0
  const bool tile = tile_fused(S);
  const bool two_level = tile && eu_is_two_level(S);
  if (two_level && S->has_comm && !(eu_is_mg(S) && ghost_mode(S) && S->slab_on)) {
    eu_set_error("coarse-correction preconditioners with several ranks: EULER_PRECOND_IC0_TILE_MG on row slabs without mailboxes only");
    return EULER_EINVAL;
  }
  if (two_level && (rc = eu_launch_coarse_setup(S))) return rc;   // P^T A P of this system, its factor and inverse (k_coarse.hip)
  if (two_level && (rc = eu_launch_coarse_consistent(S))) return rc;   // (water cut off from the air: r = b made compatible with the singular A)
  if (tile) {                                                     // E^-1 per tile, then z = M^-1 r and sigma = dot(z, r) in one pass
    if ((rc = launch_factor_tile(S, 0))) return rc;
    if ((rc = launch_precond_tile(S, 0, 1, FIN_SIGMA_INIT, 0, 0.0))) return rc;
  } else {
  if (S->cfg.precond != EULER_PRECOND_JACOBI && (rc = launch_sweep<SW_FACTOR>(S, KC_PRECON_FACTOR, 0))) return rc;   // once per solve
  if ((rc = launch_precondition(S, 0, FIN_SIGMA_INIT))) return rc;
  }
  if ((rc = ring_begin(S))) return rc;      // (S->s = ring[0] takes the first search direction)
  // multilevel mode on one GPU: the first search direction s = z + P y is formed by the first k_search_apply itself (beta is 0 behind k_pcg_reset, and 0 times what an earlier
  // solve left in the ring is 0) - a copy pass and a k_apply_a less per solve.  Only there: no recorded digest pins this mode's zeros' signs (z = -0 comes out as +0)
  const bool fold0 = two_level && eu_is_mg(S) && !S->has_comm && S->cfg.dot_mode != EULER_DOT_SEQUENTIAL && sa_run(S) == 8;
  if (fold0) {
    // (below) - beta = 0 times what an earlier solve left in ring[n - 1].  That product is only 0 while the array is finite: a solve that broke down (tol = 0 runs where
    // z . s reaches 0: alpha = 0 / 0) leaves Inf / NaN there, and 0 x Inf = NaN would poison every later solve - such a solve's ring is cleared first
    const PcgScalars& ps = prev_solve;
    if (ps.nonzero && !(std::isfinite(ps.alpha) && std::isfinite(ps.beta) && std::isfinite(ps.rnorm) && std::isfinite(ps.sigma) && std::isfinite(ps.zs)))
      for (int k = 0; k < S->s_ring_n; ++k) HIPCHK(hipMemsetAsync(S->s_ring[k] + S->skew_off, 0, S->Sw * sizeof(double), S->stream));
  }
  else if (two_level) { if ((rc = eu_launch_coarse_search_init(S))) return rc; }      // s = z + P y
  else
  LAUNCH(S, KC_UPDATE_SEARCH, k_update_search<true>, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), LOC(S->s), LOC(S->z),
         LOC(S->cellmask), S->e_cnt, S->sc, 0, 0.0);
  if (S->has_comm && eu_p2p_has_neighbour_arrays(S)) {   // the neighbours read these rows of s in the second iteration's k_search_apply
    const int y0 = S->band_lo > 0 ? 64 * S->band_lo : -1, y1 = S->band_hi < S->geom.nbands ? 64 * S->band_hi - 1 : -1;
    LAUNCH(S, KC_UPDATE_SEARCH, k_publish_edge_rows, dim3((S->X + 255) / 256, 2), dim3(256), S->s, S->geom, y0, y1, S->sc);
  }
  // ranks with their neighbours' z and s mapped run the fused kernel too (it reads across the slab boundary directly)
  // update_search of iteration k rides along with apply_a of iteration k + 1 in every configuration (several ranks: the ghost
  // rows of z and s are exchanged in front of it)
  // s' of non-fluid cells is never written: with two arrays the second one is cleared per solve as ever; a ring of four keeps what earlier solves left
  // off the fluid - nothing reads those elements unmasked, EULER_F_PCG_S shows them as +0 (s_stale) - rather than clear three arrays per solve
  // (s_stale with two arrays as well: the copy s = z above carries what z holds in tiles that are no longer active)
  if (S->s_ring_n == 2) HIPCHK(hipMemsetAsync(LOC(S->s2), 0, S->e_cnt * sizeof(double), S->stream));
  S->s_stale = 1;
  HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  const int poll = S->cfg.pcg_poll_interval > 0 ? S->cfg.pcg_poll_interval : 8;
  const int max_it = S->cfg.max_iterations;
  int it = 0, chunk = 0;
  bool stop = !S->sc_host->nonzero;   // all_zero(r): main.c:742
  S->s_none = stop ? 1 : 0;           // (no search direction exists: EULER_F_PCG_S reads as +0)
  while (it < max_it && !stop) {
    int chunk_end = it + poll;
    if (!drain_poll && pred_iters > 0 && it < pred_iters + poll && chunk_end > pred_iters - 2) {      // around the predicted end: short chunks (the first of them ends at pred - 2)
      const int lo = pred_iters - 2 < chunk_end ? pred_iters - 2 : chunk_end;
      chunk_end = lo > it + 2 ? lo : it + 2;
    }
    if (chunk_end > max_it) chunk_end = max_it;
    for (; it < chunk_end; ++it) {
      S->prof_iter = it;
      const bool fused = it > 0 || fold0;   // update_search of iteration it-1 rides along with this apply_a (A s' lands in q)
      if (it == 0 && fold0) { S->s = S->s_ring[S->s_ring_n - 1]; S->s2 = S->s_ring[0]; }      // (the pass writes ring[0], the first direction's place; what it reads of s counts 0 times)
      if ((rc = fused ? launch_search_apply_and_alpha(S, it) : launch_apply_a_and_alpha(S, 0))) return rc;
      if (tile) {
        // r -= alpha A s, max |r| (sets `done`), z = M^-1 r, beta = dot(z, r) / sigma: one pass; on the last iteration of the
        // budget only r and its norm (main.c:760-765 would be computed and never consumed)
        S->prof_iter = it;
        if ((rc = launch_precond_tile(S, 1, it + 1 < max_it, FIN_BETA, 0, 0.0))) return rc;
        continue;
      }
      // r -= alpha A s, max |r| (sets `done`): the first half of the tile pass, over the solve's active chunks (A s sits in q, or in z
      // behind the first apply_a); p += alpha s rides in k_search_apply / k_finish_p
      S->tile_as_override = fused ? S->q : S->z;
      rc = launch_precond_tile(S, 1, 0, -1, 0, 0.0, true);
      S->tile_as_override = nullptr;
      if (rc) return rc;
      if (it + 1 < max_it) {   // the tail of the last iteration (main.c:760-765) is never consumed
        // these belong to iteration `it` but only run if it did not converge: tag them it+1 so that
        // they count as active only when the device went on to iteration it+1
        S->prof_iter = it + 1;
        if ((rc = launch_precondition(S, 0, FIN_BETA))) return rc;
      }
    }
    if (it < max_it && drain_poll) {
      HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
      HIPCHK(hipStreamSynchronize(S->stream));
      stop = S->sc_host->done != 0;
    } else
    if (it < max_it) {   // poll the device-side convergence flag (identical on every rank): one chunk late, see above
      const int slot = chunk & 1;
      if (chunk > 0) {                                      // the poll behind the previous chunk
        HIPCHK(hipEventSynchronize(S->poll_event[slot ^ 1]));
        stop = S->poll_host[slot ^ 1].done != 0;
      }
      HIPCHK(hipMemcpyAsync(&S->poll_host[slot], S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
      HIPCHK(hipEventRecord(S->poll_event[slot], S->stream));
      ++chunk;
    }
  }
  S->prof_iter = -2;
  // the last p += alpha s (the others rode along with the apply_a passes): in memory where a communicator needs the finished rows (and in the two-pass form), else inside the
  // velocity update's pass (k_grid.hip k_velocity_update_para)
  const bool finish_in_update = !S->has_comm && !S->slab_on && S->opt[EULER_OPT_VELOCITY_TWO_PASS] == 0;
  if (!finish_in_update) { if ((rc = eu_launch_finish_p(S))) return rc; }
  if (S->s_ring_n > 2) { S->s = S->s_base[0]; S->s2 = S->s_base[1]; }      // between solves S->s / S->s2 are the handle's own two arrays (EULER_F_PCG_S finds the last direction in the ring)
  if (S->has_comm && S->slab_on) {
    // row slabs: the velocity update of the highest own row reads p one row up (main.c:800) - one ghost row from the rank above
    const int X = S->X, nbk = (X + 255) / 256;
    double *send_lo = S->halo_buf, *send_hi = S->halo_buf + X, *recv_lo = S->halo_buf + 2 * X, *recv_hi = S->halo_buf + 3 * X;
    if (S->band_lo > 0) hipLaunchKernelGGL(k_pack_row, dim3(nbk), dim3(256), 0, S->stream, S->p, send_lo, S->geom, S->band_lo, 0);
    COMM_CALL(S->bulk.halo(S->bulk.ctx, send_lo, send_hi, recv_lo, recv_hi, X));
    if (S->band_hi < S->geom.nbands) hipLaunchKernelGGL(k_unpack_row, dim3(nbk), dim3(256), 0, S->stream, S->p, recv_hi, S->geom, S->band_hi, 0);
  } else if (S->has_comm) {   // every rank needs the whole pressure field for the replicated velocity update
    const int n = S->comm.nranks, nb = S->geom.nbands;
    std::vector<int64_t> off(n), cnt(n);
    for (int k = 0; k < n; ++k) {
      const int64_t lo = (int64_t)nb * k / n, hi = (int64_t)nb * (k + 1) / n;
      off[k] = lo * S->geom.TS * 64 * 8;
      cnt[k] = (hi - lo) * S->geom.TS * 64 * 8;
    }
    COMM_CALL(S->comm.allgather(S->comm.ctx, S->p, off.data(), cnt.data()));
  }
  if ((rc = eu_launch_velocity_update(S, dt, finish_in_update ? 1 : 0))) return rc;
  HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
  return EULER_OK;
}

// single building blocks for kernel-level parity tests (euler_pcg_op)
int eu_launch_pcg_op(euler_sim* S, int op, float dt, double a, double* out) {
  S->s_none = 0;
  S->s_launched = 0;      // (a single operation works on S->s as it stands)
  const bool want_scalar = op == EULER_OP_DOT_ZR || op == EULER_OP_DOT_ZS || op == EULER_OP_INF_NORM_R;
  const size_t SS = S->geom.S;
  if (eu_is_two_level(S) && (op == EULER_OP_PRECON_FACTOR || op == EULER_OP_FORWARD_SOLVE || op == EULER_OP_BACKWARD_SOLVE)) {
    // the coarse level is assembled over a solve's chunk list and lives in the solve: a single forced operation would return the tile level alone
    eu_set_error("euler_pcg_op: the preconditioner of EULER_PRECOND_IC0_TILE2 / _MG is not available as a single operation (select EULER_PRECOND_IC0_TILE for its tile level)");
    return EULER_EINVAL;
  }
  switch (op) {
    case EULER_OP_BUILD_SYSTEM:
      LAUNCH(S, KC_MISC, k_pcg_reset, dim3(1), dim3(1), S->sc, S->cfg.tol, S->cfg.max_iterations);
      eu_launch_build_system(S, dt);
      break;
    case EULER_OP_PRECON_FACTOR: if (tile_fused(S)) launch_factor_tile(S, 1); else launch_sweep<SW_FACTOR>(S, KC_PRECON_FACTOR, 1); break;
    // tile-local mode has no forward solve of its own (q lives in registers): BACKWARD_SOLVE runs the whole Z = M^-1 R
    case EULER_OP_FORWARD_SOLVE: if (!tile_fused(S)) launch_sweep<SW_FORWARD>(S, KC_FORWARD_SOLVE, 1); break;
    case EULER_OP_BACKWARD_SOLVE:
      if (tile_fused(S)) launch_precond_tile(S, 0, 1, FIN_STORE_ONLY, 1, 0.0); else launch_sweep<SW_BACKWARD>(S, KC_BACKWARD_SOLVE, 1);
      break;
    case EULER_OP_APPLY_A:
      LAUNCH(S, KC_APPLY_A, k_apply_a, dim3(S->red_blocks), dim3(RED_THREADS), S->s, S->z, S->cellmask, S->geom,
             S->partial, S->sc, 1, S->red_counter, -1);
      break;
    case EULER_OP_DOT_ZR: launch_dot(S, S->z, S->r, FIN_STORE_ONLY, 1); break;
    case EULER_OP_DOT_ZS: launch_dot(S, S->z, S->s, FIN_STORE_ONLY, 1); break;
    case EULER_OP_INF_NORM_R:
      LAUNCH(S, KC_UPDATE_PR, k_inf_norm, dim3(S->red_blocks), dim3(RED_THREADS), S->r, S->cellmask, SS, S->partial);
      LAUNCH(S, KC_REDUCE_FINAL, k_reduce_final<true>, dim3(1), dim3(RED_THREADS), S->partial, S->red_blocks, S->sc,
             (int)FIN_STORE_ONLY, 1);
      break;
    case EULER_OP_UPDATE_PR:
      LAUNCH(S, KC_UPDATE_PR, k_update_pr, dim3(S->red_blocks), dim3(RED_THREADS), S->p, S->r, S->s, S->z, S->cellmask,
             SS, S->partial, S->sc, 1, a, S->red_counter, -1);
      break;
    case EULER_OP_UPDATE_SEARCH:
      LAUNCH(S, KC_UPDATE_SEARCH, k_update_search<false>, dim3(eu_blocks(SS, 256 * 4, 4096)), dim3(256), S->s, S->z,
             S->cellmask, SS, S->sc, 1, a);
      break;
    default: eu_set_error("unknown pcg op %d", op); return EULER_EINVAL;
  }
  HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  if (want_scalar && out) *out = S->sc_host->sigma_new;
  return EULER_OK;
}
