// k_coarse.hip — the coarse levels of the TWO-LEVEL and the MULTILEVEL preconditioner (EULER_PRECOND_IC0_TILE2 / _TILE_MG, include/euler.h; round 3):
//
//     z = M_tile^-1 r + P (P^T A P)^-1 P^T r            (two-level; the first part of this file)
//     z = M_tile^-1 r + P_0 V(P_0^T r)                  (multilevel: "Multilevel mode" below - a V-cycle over bilinear node grids 8, 16, 32, ... cells apart (k_mg.hip) whose top level is the
//                                                        dense level of the two-level mode; also on row slabs)
//
// M_tile = the tile-local IC(0) of k_pcg.hip (64-row x 16-record blocks, one pass over memory); P = piecewise constants over coarse
// cells of g x g grid cells (g = 64 m, a power of two; at most 256 coarse cells, 16 x 16 on a square grid) restricted to the fluid.
// No reference counterpart (the reference has ONE preconditioner, main.c:580-627): this is an extension of the roofline mode, restated
// in the oracle (eo_sim.coarse_m, coarse_correction) and compared with it to rounding.  Why: a block-local factor has no coupling
// beyond its block, and the first hundred iterations of a large solve - all the reference's cap ever allows, main.c:735 - live on the
// long-range part of A^-1 (the hydrostatic mode of a tank).  The coarse space has that part: docs/solver_two_level.md.
//
// What runs where:
//   per solve      k_coarse_assemble  P^T A P as integer sums of A's entries per pair of coarse cells (a 5-point stencil over the coarse
//                                     grid; integers: exact whatever the order of the atomics), walked tile by tile like k_factor_tile
//                  k_coarse_factor    its dense Cholesky factor, banded (half-bandwidth nx), one workgroup
//                  k_coarse_inverse   the explicit inverse, one thread per column, columns in LDS (n <= 256: 0.5 MB) - applying the
//                                     preconditioner is then a 256 x 256 matrix-vector product instead of two triangular solves on the critical path
//   per iteration  k_precond_tile     (k_pcg.hip) leaves, per tile, the sums of the new r over its fluid cells by coarse column: 3 doubles
//                  k_coarse_solve     a workgroup per coarse cell: r_c = sums of those partials in a fixed order; the last one to finish:
//                                     y = (P^T A P)^-1 r_c, dot(z, r) += y . r_c, and the scalar epilogue (sigma / beta) k_precond_tile leaves to it
//                  k_search_apply     (k_pcg.hip, COARSE) adds y of a cell's coarse cell to z wherever it forms s' = z + beta s
#include "euler_dev.h"
#include "k_mg.h"

#include <stdlib.h>

#define CC_MAX 256          // coarse cells
#define CC_THREADS 1024
#define CC_NULL_MAX 4       // fluid regions cut off from the air whose indicators are kept (k_coarse_nullfix)
#define CC_NULL_TOTAL (CC_NULL_MAX * CC_MAX + 1 + 2 * CC_NULL_MAX + 512 * 2 * CC_NULL_MAX + 1)      // = NS_TOTAL below
#define COMM_CALL(expr) do { if ((expr) != 0) { eu_set_error("communicator callback failed: %s", #expr); return EULER_ECOMM; } } while (0)

// the scalar epilogues of k_pcg.hip (same codes)
enum { CFIN_SIGMA_INIT = 0, CFIN_BETA = 3 };

void eu_coarse_release(euler_sim* S);
int eu_coarse_alloc(euler_sim* S, bool mg) {
  if (S->cc_null) return mg ? eu_mg_alloc(S) : EULER_OK;      // (the last allocation below: everything of the dense level is there)
  eu_coarse_release(S);                // (a failed earlier attempt may have left some of it)
  int m = 1, shift = 6;
  while (((S->X + 64 * m - 1) / (64 * m)) * ((S->geom.nbands + m - 1) / m) > CC_MAX) { m *= 2; shift += 1; }
  S->coarse_m = m; S->coarse_shift = shift;
  S->coarse_nx = (S->X + 64 * m - 1) / (64 * m);
  S->coarse_ny = (S->geom.nbands + m - 1) / m;
  S->coarse_n = S->coarse_nx * S->coarse_ny;
  const size_t n = (size_t)S->coarse_n, nn = n > MG_TOP_MAX ? n : MG_TOP_MAX;      // (the multilevel mode's dense level has its own size: at most MG_TOP_MAX nodes)
  HIPCHK(hipMalloc((void**)&S->cc_diag, 4 * nn * sizeof(int)));      // diagonal, right, up; pinned[n] (k_coarse_factor)
  S->cc_right = S->cc_diag + nn; S->cc_up = S->cc_diag + 2 * nn; S->cc_pinned = S->cc_diag + 3 * nn;
  HIPCHK(hipMalloc((void**)&S->cc_sten, 5 * nn * sizeof(double)));    // the two-level mode's stencil as doubles: d, e, n, ne, nw (k_coarse_factor's input)
  HIPCHK(hipMalloc((void**)&S->cc_fac, nn * nn * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_inv, nn * nn * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_part, (S->chunk_cap + 64) * 3 * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_y, (2 * CC_MAX + 1) * sizeof(double)));      // y [CC_MAX], r_c [CC_MAX], the ticket counter of k_coarse_solve
  HIPCHK(hipMemset(S->cc_part, 0, (S->chunk_cap + 64) * 3 * sizeof(double)));
  HIPCHK(hipMemset(S->cc_y, 0, (2 * CC_MAX + 1) * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_null, CC_NULL_TOTAL * sizeof(double)));      // indicators, their number, the sums / partials / ticket of k_null_sums
  HIPCHK(hipMemset(S->cc_null, 0, CC_NULL_TOTAL * sizeof(double)));
  if (mg) return eu_mg_alloc(S);
  return EULER_OK;
}

void eu_coarse_release(euler_sim* S) {
  if (S->cc_diag) (void)hipFree(S->cc_diag);
  for (double* d : {S->cc_fac, S->cc_inv, S->cc_part, S->cc_y, S->cc_null, S->cc_sten}) if (d) (void)hipFree(d);
  S->cc_diag = S->cc_right = S->cc_up = S->cc_pinned = nullptr;
  S->cc_fac = S->cc_inv = S->cc_part = S->cc_y = S->cc_null = S->cc_sten = nullptr;
  eu_mg_release(S);
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ---- P^T A P.  A tile = 16 records x 64 lanes of one band: cell (lane l, record t) sits in column x = t - l, row 64 band + l.  Its
// columns span at most three coarse columns (79 columns, g >= 64), counted from J0 = the coarse column of the tile's leftmost column.
// Per coarse cell c:  diag += a_diag of every fluid cell, - 2 per fluid-fluid edge inside c;  right[c] -= 1 per edge to coarse cell c + 1;
// up[c] -= 1 per edge to c + nx.  Every edge is visited once, from its left / lower cell (CM_RIGHT, CM_UP of the mask byte).
__global__ __launch_bounds__(256) void k_coarse_assemble(const uint8_t* __restrict__ mask, SkewGeom g, const unsigned int* __restrict__ list,
                                                         const PcgScalars* sc, int band_lo, int shift, int m, int nx, int* cd, int* cr, int* cu) {
  const int lane = threadIdx.x & 63;
  const int ntb = g.T / 16, todo = (int)sc->n_chunks;
  const int n_waves = gridDim.x * 4;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < todo; i += n_waves) {
    const int tile = (int)(list[i] & ~EU_CHUNK_INTERIOR);
    const int band = band_lo + tile / ntb, k = tile % ntb;
    const size_t base = ((size_t)band * g.TS + (size_t)k * 16) * 64 + 2 * lane;
    const int xl = k * 16 - 63;
    const int J0 = (xl > 0 ? xl : 0) >> shift;
    const int I = band / m;
    const bool top_of_coarse_row = lane == 63 && (band + 1) % m == 0;      // the cell above lies in the next coarse row
    int d[3] = {0, 0, 0}, r[3] = {0, 0, 0}, u[3] = {0, 0, 0};
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + base + P * 128);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned int cm = (mm >> (8 * h)) & 0xff;
        if (!(cm & CM_FLUID)) continue;
        const int x = k * 16 + 2 * P + h - lane;      // >= 1 for a fluid cell
        const int b = (x >> shift) - J0;
        int dd = (int)(cm >> CM_DIAG_SHIFT), rr = 0, uu = 0;
        if (cm & CM_RIGHT) { if (((x + 1) >> shift) - J0 == b) dd -= 2; else rr -= 1; }
        if (cm & CM_UP) { if (!top_of_coarse_row) dd -= 2; else uu -= 1; }
        if (b == 0) { d[0] += dd; r[0] += rr; u[0] += uu; }
        else if (b == 1) { d[1] += dd; r[1] += rr; u[1] += uu; }
        else { d[2] += dd; r[2] += rr; u[2] += uu; }
      }
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int sd = wave_sum_i(d[b]), sr = wave_sum_i(r[b]), su = wave_sum_i(u[b]);
      if (lane == 0 && J0 + b < nx) {
        const int c = I * nx + J0 + b;
        if (sd) atomicAdd(&cd[c], sd);
        if (sr) atomicAdd(&cr[c], sr);
        if (su) atomicAdd(&cu[c], su);
      }
    }
  }
}

// ---- the dense matrix from the stencil and its banded Cholesky factor (lower triangle), one workgroup.
// Column k: d = sqrt(a_kk); the bw entries below it are divided by d; the bw x bw window behind it takes the rank-1 update -
// an element per thread and pass (bw = nx = 16 on a square grid: one pass of 1024 threads).  An empty coarse cell gets a unit diagonal
// (its right-hand side is 0).  The band is worked on in LDS when it fits (element (i, j) at i * bw + j + bw: n (bw + 1) doubles, 34 KB for
// 16 x 16) and copied out to the dense array the inverse reads; a wide band (a flat grid with few coarse rows) is factored in place in
// global memory instead - same arithmetic, same order.
#define CC_LDS_BAND 8192      // doubles
// the stencil -> doubles (the two-level mode's integer sums; the multilevel mode hands its top level's nine-point stencil over directly)
__global__ __launch_bounds__(256) void k_coarse_sten(const int* __restrict__ cd, const int* __restrict__ cr, const int* __restrict__ cu, int n, double* __restrict__ st) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  st[c] = (double)cd[c]; st[n + c] = (double)cr[c]; st[2 * n + c] = (double)cu[c]; st[3 * n + c] = 0.0; st[4 * n + c] = 0.0;
}
// sd: the diagonal; se / sn / sne / snw: the couplings of node c to c + 1, c + nx, c + nx + 1, c + nx - 1 (the last two 0 for a five-point stencil)
__global__ __launch_bounds__(CC_THREADS) void k_coarse_factor(const double* __restrict__ sd, const double* __restrict__ se, const double* __restrict__ sn, const double* __restrict__ sne,
                                                              const double* __restrict__ snw, int n, int nx, int bw, double* __restrict__ A, const PcgScalars* sc, int* __restrict__ pinned) {
  if (!sc->nonzero) return;
  __shared__ double s_band[CC_LDS_BAND];
  __shared__ double s_d;
  const int tid = threadIdx.x;
  const bool in_lds = n * (bw + 1) + bw <= CC_LDS_BAND;
  double* M = in_lds ? s_band : A;
  const int rs = in_lds ? bw : n, off = in_lds ? bw : 0;      // element (i, j), i - bw <= j <= i, at M[i * rs + j + off]
  for (int e = tid; e < n * n; e += CC_THREADS) A[e] = 0.0;
  if (in_lds) for (int e = tid; e < CC_LDS_BAND; e += CC_THREADS) s_band[e] = 0.0;
  for (int c = tid; c < n; c += CC_THREADS) pinned[c] = 0;
  __syncthreads();
  for (int c = tid; c < n; c += CC_THREADS) {
    const int J = c % nx;
    M[c * rs + c + off] = sd[c] != 0.0 ? sd[c] : 1.0;
    if (J + 1 < nx && c + 1 < n) M[(c + 1) * rs + c + off] = se[c];
    if (c + nx < n) M[(c + nx) * rs + c + off] = sn[c];
    if (bw > nx && J + 1 < nx && c + nx + 1 < n) M[(c + nx + 1) * rs + c + off] = sne[c];
    if (bw > nx && J > 0 && c + nx - 1 < n) M[(c + nx - 1) * rs + c + off] = snw[c];
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const int w = k + bw < n - 1 ? bw : n - 1 - k;      // rows below the diagonal inside the band
    if (tid == 0) {
      // a fluid region cut off from the air (closed box, enclosed pool) makes A - and P^T A P - singular: the last pivot of such a component is rounding noise instead
      // of 0.  Like the reference's own factor (main.c:595) the pivot falls back to the matrix' diagonal then: that coarse cell is pinned, the operator stays positive definite
      double piv = M[k * rs + k + off];
      const double a_kk = sd[k] != 0.0 ? sd[k] : 1.0;
      if (!(piv > 1e-8 * a_kk)) { piv = a_kk; if (sd[k] != 0.0) pinned[k] = 1; }      // (k_coarse_nullfix takes the pinned cell's component out of the inverse again)
      const double d = sqrt(piv); M[k * rs + k + off] = d; s_d = d;
    }
    __syncthreads();
    if (tid < w) M[(k + 1 + tid) * rs + k + off] /= s_d;
    __syncthreads();
    for (int idx = tid; idx < w * w; idx += CC_THREADS) {
      const int ti = idx / w, tj = idx % w;
      if (tj <= ti) M[(k + 1 + ti) * rs + (k + 1 + tj) + off] -= M[(k + 1 + ti) * rs + k + off] * M[(k + 1 + tj) * rs + k + off];
    }
    __syncthreads();
  }
  if (in_lds)
    for (int e = tid; e < n * (bw + 1); e += CC_THREADS) {
      const int i = e / (bw + 1), j = i - bw + e % (bw + 1);
      if (j >= 0) A[(size_t)i * n + j] = s_band[i * rs + j + off];
    }
}

// ---- the explicit inverse: wave c solves L L^T x = e_c by two banded substitutions - lanes over the band (one product per lane and
// step for bw <= 64), a shuffle fold, x in LDS - and leaves it as row c of inv (coalesced; the inverse is symmetric).  Four columns per
// workgroup; the factor's band is staged in LDS first when it fits (as in k_coarse_factor: 34 KB for 16 x 16 coarse cells), so a step
// of the dependent chain costs LDS latency, not L2's (measured: 1110 us as a thread per column out of global memory -> ~60 us).
#define CC_INV_BAND 6144      // doubles
__device__ __forceinline__ double wave_sum_all(double v) {      // the same value in every lane (butterfly)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void k_coarse_inverse(const double* __restrict__ L, int n, int bw, double* __restrict__ inv, const PcgScalars* sc) {
  if (!sc->nonzero) return;
  __shared__ double s_band[CC_INV_BAND];
  __shared__ double s_x[4][CC_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = blockIdx.x * 4 + wave;
  const bool in_lds = n * (bw + 1) + bw <= CC_INV_BAND;
  if (in_lds) {
    for (int e = tid; e < n * (bw + 1); e += 256) {
      const int i = e / (bw + 1), j = i - bw + e % (bw + 1);
      s_band[i * bw + j + bw] = j >= 0 ? L[(size_t)i * n + j] : 0.0;
    }
    __syncthreads();
  }
  if (c >= n) return;      // (no workgroup barrier below)
  const double* M = in_lds ? s_band : L;
  const int rs = in_lds ? bw : n, off = in_lds ? bw : 0;      // element (i, j), i - bw <= j <= i, at M[i * rs + j + off]
  double* x = s_x[wave];
  for (int i = lane; i < c; i += 64) x[i] = 0.0;
  __builtin_amdgcn_wave_barrier();
  for (int i = c; i < n; ++i) {      // L w = e_c  (w_i = 0 for i < c)
    const int j0 = i - bw > c ? i - bw : c;
    double t = 0.0;
    for (int j = j0 + lane; j < i; j += 64) t += M[i * rs + j + off] * x[j];
    t = wave_sum_all(t);
    const double xi = ((i == c ? 1.0 : 0.0) - t) / M[i * rs + i + off];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) x[i] = xi;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = n - 1; i >= 0; --i) {      // L^T x = w
    const int j1 = i + bw < n - 1 ? i + bw : n - 1;
    double t = 0.0;
    for (int j = i + 1 + lane; j <= j1; j += 64) t += M[j * rs + i + off] * x[j];
    t = wave_sum_all(t);
    const double xi = (x[i] - t) / M[i * rs + i + off];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) x[i] = xi;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < n; i += 64) inv[(size_t)c * n + i] = x[i];
}

// ---- fluid cut off from the air: P^T A P has the indicator n of every such component in its null space, the factor pinned one cell of it (k_coarse_factor) and the inverse
// S of the pinned matrix treats the component lopsidedly - PCG then stalls on the part of r along n that no A s can touch.  With n = a_kk S e_k (the pinned system's answer
// to the pin itself: exactly the indicator) the inverse becomes (I - n n^T / n.n) S (I - n n^T / n.n): the pseudo-inverse of P^T A P - zero along n, S elsewhere.
__global__ __launch_bounds__(CC_THREADS) void k_coarse_nullfix(double* __restrict__ inv, const double* __restrict__ sd, const int* __restrict__ pinned, int n, const PcgScalars* sc,
                                                               double* __restrict__ nullv) {
  const int tid = threadIdx.x;
  if (tid == 0) nullv[CC_NULL_MAX * CC_MAX] = 0.0;
  if (!sc->nonzero) return;
  __shared__ double s_n[CC_MAX], s_w[CC_MAX], s_red[CC_THREADS / 64];
  __shared__ double s_nn, s_nw;
  int found = 0;
  for (int k = 0; k < n; ++k) {
    if (!pinned[k]) continue;      // (uniform)
    const double a_kk = sd[k];
    if (tid < n) s_n[tid] = a_kk * inv[(size_t)tid * n + k];
    __syncthreads();
    if (found < CC_NULL_MAX) {      // kept for eu_launch_coarse_consistent
      if (tid < CC_MAX) nullv[found * CC_MAX + tid] = tid < n ? s_n[tid] : 0.0;
      if (tid == 0) nullv[CC_NULL_MAX * CC_MAX] = (double)(found + 1);
    }
    ++found;
    double v = tid < n ? s_n[tid] * s_n[tid] : 0.0;
    v = eu_wave_sum(v);
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int q = 0; q < CC_THREADS / 64; ++q) t += s_red[q]; s_nn = t; }
    __syncthreads();
    if (tid < n) {      // w = S n
      double t = 0.0;
      for (int j = 0; j < n; ++j) t += inv[(size_t)j * n + tid] * s_n[j];
      s_w[tid] = t;
    }
    __syncthreads();
    v = tid < n ? s_n[tid] * s_w[tid] : 0.0;
    v = eu_wave_sum(v);
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int q = 0; q < CC_THREADS / 64; ++q) t += s_red[q]; s_nw = t; }
    __syncthreads();
    const double nn = s_nn, c = s_nw / (nn * nn);
    for (int e = tid; e < n * n; e += CC_THREADS) {
      const int i = e / n, j = e % n;
      inv[e] = inv[e] - (s_n[i] * s_w[j] + s_w[i] * s_n[j]) / nn + s_n[i] * s_n[j] * c;
    }
    __syncthreads();
  }
}

// ---- water cut off from the air, continued: b - float divergences - is compatible with the singular A only to rounding (n.b ~ 1e-4 over such a region, not 0), and CG on a
// singular, slightly inconsistent system wanders once it gets close (seen: 2000 iterations instead of 100).  The part of r = b along every region's indicator - a few 1e-9 per cell,
// far below the tolerance, and nothing any A s could touch - is taken out before the solve starts: eps_q = n_q . r and n_q . n_q for all (up to four) indicators in one pass over
// this rank's cells (per-block partials, the last block folds them in block order), on row slabs the sums of all ranks (ONE all-reduce of 8 doubles: the indicators, like the whole
// dense level, are the same on every rank), then r -= sum_q n_q eps_q / nn_q.  Both launches are unconditional and return at once on the device when no region is cut off
// (the usual case): no host round trip per solve (round 3: a one-workgroup kernel behind a hipStreamSynchronize, one GPU only).
#define NULL_BLOCKS 512
// cc_null layout: [CC_NULL_MAX][CC_MAX] indicators, [CC_NULL_MAX * CC_MAX] their number, then NS_SUMS: 8 sums {eps_q, nn_q}, NS_PART: per-block partials, NS_TICKET
#define NS_SUMS (CC_NULL_MAX * CC_MAX + 1)
#define NS_PART (NS_SUMS + 2 * CC_NULL_MAX)
#define NS_TICKET (NS_PART + NULL_BLOCKS * 2 * CC_NULL_MAX)
#define NS_TOTAL (NS_TICKET + 1)
static_assert(NS_TOTAL == CC_NULL_TOTAL, "cc_null layout");
// MgNull: the multilevel mode's indicators live on its node grids - mg0 != null: indicator q on level 0 at mg0 + q * stride, sampled through P_0 (bilinear)
struct MgNull { const double* mg0; size_t stride; int nx0, ny0; };
__device__ __forceinline__ double null_at(const double* __restrict__ nullv, const MgNull& M, int q, int shift, int nx, int x, int y) {
  if (M.mg0) return mg_interp0(M.mg0 + (size_t)q * M.stride, M.nx0, M.ny0, x, y);
  return nullv[q * CC_MAX + (size_t)(y >> shift) * nx + (x >> shift)];
}
__global__ __launch_bounds__(256) void k_null_sums(const double* __restrict__ r, const uint8_t* __restrict__ mask, SkewGeom g, size_t e_lo, size_t e_cnt,
                                                   double* __restrict__ nullv, int shift, int nx, const PcgScalars* sc, MgNull M) {
  const int count = (int)nullv[CC_NULL_MAX * CC_MAX];
  if (!sc->nonzero || count <= 0) {      // (row slabs all-reduce the sums whatever they hold: keep them finite)
    if (blockIdx.x == 0 && threadIdx.x < 2 * CC_NULL_MAX) nullv[NS_SUMS + threadIdx.x] = 0.0;
    return;
  }
  double acc[2 * CC_NULL_MAX];
#pragma unroll
  for (int k = 0; k < 2 * CC_NULL_MAX; ++k) acc[k] = 0.0;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < e_cnt; k += (size_t)gridDim.x * 256) {
    const size_t e = e_lo + k;
    if (!(mask[e] & CM_FLUID)) continue;
    int band, t, l;
    skew_decode(g, e, band, t, l);
    const double rv = r[e];
#pragma unroll
    for (int q = 0; q < CC_NULL_MAX; ++q)
      if (q < count) { const double w = null_at(nullv, M, q, shift, nx, t - l, band * 64 + l); acc[2 * q] += rv * w; acc[2 * q + 1] += w * w; }
  }
  __shared__ double s_red[4][2 * CC_NULL_MAX];
  __shared__ int am_last;
#pragma unroll
  for (int k = 0; k < 2 * CC_NULL_MAX; ++k) {
    const double v = eu_wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  double* part = nullv + NS_PART;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(nullv + NS_TICKET);
  if (threadIdx.x < 2 * CC_NULL_MAX) {
    const int k = threadIdx.x;
    const double t = (s_red[0][k] + s_red[1][k]) + (s_red[2][k] + s_red[3][k]);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&part[(size_t)blockIdx.x * 2 * CC_NULL_MAX + k]), (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) am_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!am_last) return;
  if (threadIdx.x < 2 * CC_NULL_MAX) {      // block order: the same bits whichever block comes last
    double t = 0.0;
    for (unsigned int b2 = 0; b2 < gridDim.x; ++b2)
      t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&part[(size_t)b2 * 2 * CC_NULL_MAX + threadIdx.x]), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
    nullv[NS_SUMS + threadIdx.x] = t;
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_null_apply(double* __restrict__ r, const uint8_t* __restrict__ mask, SkewGeom g, size_t e_lo, size_t e_cnt,
                                                    const double* __restrict__ nullv, int shift, int nx, const PcgScalars* sc, MgNull M) {
  const int count = (int)nullv[CC_NULL_MAX * CC_MAX];
  if (!sc->nonzero || count <= 0) return;
  double f[CC_NULL_MAX];
#pragma unroll
  for (int q = 0; q < CC_NULL_MAX; ++q) { const double nn = nullv[NS_SUMS + 2 * q + 1]; f[q] = (q < count && nn > 1e-12) ? nullv[NS_SUMS + 2 * q] / nn : 0.0; }      // (1e-12: a dependent node's "indicator" is rounding noise on the cells - oracle: eo_project)
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < e_cnt; k += (size_t)gridDim.x * 256) {
    const size_t e = e_lo + k;
    if (!(mask[e] & CM_FLUID)) continue;
    int band, t, l;
    skew_decode(g, e, band, t, l);
    double v = r[e];
#pragma unroll
    for (int q = 0; q < CC_NULL_MAX; ++q)
      if (q < count) v = v - null_at(nullv, M, q, shift, nx, t - l, band * 64 + l) * f[q];
    r[e] = v;
  }
}
// multilevel mode: an indicator known on the dense level (cc_null) on every level below, down to level 0: n_l = P n_(l+1) on the nodes that carry fluid (weights 1, 1/2)
__global__ __launch_bounds__(256) void k_mg_null_prolong(const double* __restrict__ nullv, double* __restrict__ pool, size_t stride, const double* __restrict__ d, size_t off, int nx, int ny,
                                                         size_t coff, int cnx, int cny, int from_top, const PcgScalars* sc) {
  const int count = (int)nullv[CC_NULL_MAX * CC_MAX];
  if (!sc->nonzero || count <= 0) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nx * ny) return;
  const int i = c / nx, j = c % nx, I = i >> 1, J = j >> 1;
  const bool oy = (i & 1) && I + 1 <= cny - 1, ox = (j & 1) && J + 1 <= cnx - 1;
  const int I1 = oy ? I + 1 : I, J1 = ox ? J + 1 : J;
  const double fy = oy ? 0.5 : 0.0, fx = ox ? 0.5 : 0.0;
  for (int q = 0; q < CC_NULL_MAX && q < count; ++q) {
    const double* e = from_top ? nullv + q * CC_MAX : pool + (size_t)q * stride + coff;
    const double lo = (1.0 - fx) * e[I * cnx + J] + fx * e[I * cnx + J1], hi = (1.0 - fx) * e[I1 * cnx + J] + fx * e[I1 * cnx + J1];
    pool[(size_t)q * stride + off + c] = d[c] != 0.0 ? (1.0 - fy) * lo + fy * hi : 0.0;
  }
}
__global__ __launch_bounds__(256) void k_mg_null_top(const double* __restrict__ nullv, double* __restrict__ pool, size_t stride, size_t off, int n, const PcgScalars* sc) {      // (a hierarchy of one level: level 0 is the dense level)
  const int count = (int)nullv[CC_NULL_MAX * CC_MAX];
  if (!sc->nonzero || count <= 0) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < n) for (int q = 0; q < CC_NULL_MAX && q < count; ++q) pool[(size_t)q * stride + off + c] = nullv[q * CC_MAX + c];
}
int eu_launch_coarse_consistent(euler_sim* S) {
  const unsigned nblk = eu_blocks(S->e_cnt, 256 * 8, NULL_BLOCKS);
  MgNull M = {nullptr, 0, 0, 0};
  if (eu_is_mg(S)) {
    const int top = S->mg_levels - 1;
    if (top == 0) LAUNCH(S, KC_PRECON_FACTOR, k_mg_null_top, dim3((S->mg_nx[0] * S->mg_ny[0] + 255) / 256), dim3(256), S->cc_null, S->mg_null0, S->mg_cells, (size_t)0, S->mg_nx[0] * S->mg_ny[0], S->sc);
    for (int l = top - 1; l >= 0; --l)
      LAUNCH(S, KC_PRECON_FACTOR, k_mg_null_prolong, dim3((S->mg_nx[l] * S->mg_ny[l] + 255) / 256), dim3(256), S->cc_null, S->mg_null0, S->mg_cells, S->mg_a + 9 * S->mg_off[l] + 4 * (size_t)S->mg_nx[l] * S->mg_ny[l],
             S->mg_off[l], S->mg_nx[l], S->mg_ny[l], S->mg_off[l + 1], S->mg_nx[l + 1], S->mg_ny[l + 1], l + 1 == top ? 1 : 0, S->sc);
    M = MgNull{S->mg_null0, S->mg_cells, S->mg_nx[0], S->mg_ny[0]};
    int rc = eu_mg_null_setup(S);
    if (rc) return rc;
  }
  LAUNCH(S, KC_PRECON_FACTOR, k_null_sums, dim3(nblk), dim3(256), S->r, S->cellmask, S->geom, S->e_lo, S->e_cnt, S->cc_null, S->coarse_shift, S->coarse_nx, S->sc, M);
  if (S->has_comm) COMM_CALL(S->bulk.allreduce(S->bulk.ctx, S->cc_null + NS_SUMS, 2 * CC_NULL_MAX, 0));      // (zeros when nothing is cut off: every rank calls it, every solve)
  LAUNCH(S, KC_PRECON_FACTOR, k_null_apply, dim3(nblk), dim3(256), S->r, S->cellmask, S->geom, S->e_lo, S->e_cnt, S->cc_null, S->coarse_shift, S->coarse_nx, S->sc, M);
  return EULER_OK;
}

// ---- row slabs, multilevel mode.  The level-0 right-hand side of a node row collects tiles of the bands either side of it, so a rank forms, from ITS tiles, the node rows
// [RPB band_lo - 1, RPB band_hi + 1) - its own and one either side - into its slot of the G1 exchange's all-gather (behind {max |r|, dot(z, r)}); k_mg_scatter0 then adds, in rank
// order, what the ranks hold of every row: the same bits on every rank, and the V-cycle runs replicated from there.
int eu_coarse_comm_slots(euler_sim* S) {
  int rows = 0;
  for (int r = 0; r < S->bulk.nranks && r < 64; ++r) rows = MG_RPB * (S->part_hi[r] - S->part_lo[r]) + 2 > rows ? MG_RPB * (S->part_hi[r] - S->part_lo[r]) + 2 : rows;
  const int slot = eu_mg_split(S) ? eu_mg_split_nsmall(S) : 2 + rows * S->mg_nx[0];      // (the split cycle: a window of the gather level instead of level 0's rows)
  if (slot != S->mg_xslot || !S->mg_xbuf) {
    if (S->mg_xbuf) { if (hipStreamSynchronize(S->stream) != hipSuccess || hipFree(S->mg_xbuf) != hipSuccess) return -1; S->mg_xbuf = nullptr; }
    if (hipMalloc((void**)&S->mg_xbuf, (size_t)slot * S->bulk.nranks * sizeof(double)) != hipSuccess) { eu_set_error("hipMalloc of the multilevel exchange buffer failed"); return -1; }
    if (hipMemsetAsync(S->mg_xbuf, 0, (size_t)slot * S->bulk.nranks * sizeof(double), S->stream) != hipSuccess) return -1;
    S->mg_xslot = slot;
  }
  return slot;
}
struct MgParts { int n; int lo[64], hi[64]; };      // node rows [lo, hi) of level 0 that rank r's slot holds
__global__ __launch_bounds__(256) void k_mg_scatter0(const double* __restrict__ xbuf, int slot, MgParts P, double* __restrict__ rhs0, int nx0, int n0) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n0) return;
  const int I = c / nx0, J = c % nx0;
  double t = 0.0;
  for (int r = 0; r < P.n; ++r)
    if (I >= P.lo[r] && I < P.hi[r]) t = t + xbuf[(size_t)r * slot + 2 + (size_t)(I - P.lo[r]) * nx0 + J];
  rhs0[c] = t;
}
int eu_launch_coarse_pre(euler_sim* S, int force) { return eu_mg_slab_rows(S, force); }
int eu_launch_coarse_scatter(euler_sim* S) {
  MgParts P;
  P.n = S->bulk.nranks < 64 ? S->bulk.nranks : 64;
  const int ny0 = S->mg_ny[0];
  for (int r = 0; r < P.n; ++r) { P.lo[r] = MG_RPB * S->part_lo[r] - 1 < 0 ? 0 : MG_RPB * S->part_lo[r] - 1; P.hi[r] = MG_RPB * S->part_hi[r] + 1 > ny0 ? ny0 : MG_RPB * S->part_hi[r] + 1; }
  const int n0 = S->mg_nx[0] * ny0;
  hipLaunchKernelGGL(k_mg_scatter0, dim3((n0 + 255) / 256), dim3(256), 0, S->stream, S->mg_xbuf, S->mg_xslot, P, S->mg_rhs, S->mg_nx[0], n0);
  return EULER_OK;
}

int eu_launch_coarse_setup(euler_sim* S) {
  const unsigned nblk = eu_blocks(S->chunk_cap, 4, 2048);
  int n, nx, bw;
  const double* st[5];
  if (eu_is_mg(S)) {      // every level from the one below; the dense level is the hierarchy's top
    int rc = eu_mg_setup(S);
    if (rc) return rc;
    const int top = S->mg_levels - 1;
    nx = S->mg_nx[top]; n = nx * S->mg_ny[top];
    bw = S->mg_ny[top] > 1 ? nx + 1 : 1;
    const double* a = S->mg_a + 9 * S->mg_off[top];
    st[0] = a + 4 * (size_t)n; st[1] = a + 5 * (size_t)n; st[2] = a + 7 * (size_t)n; st[3] = a + 8 * (size_t)n; st[4] = a + 6 * (size_t)n;
  } else {
    n = S->coarse_n; nx = S->coarse_nx;
    HIPCHK(hipMemsetAsync(S->cc_diag, 0, 3 * (size_t)(n > MG_TOP_MAX ? n : MG_TOP_MAX) * sizeof(int), S->stream));
    HIPCHK(hipMemsetAsync(S->cc_part, 0, (S->chunk_cap + 64) * 3 * sizeof(double), S->stream));      // (tiles outside this solve's list contribute nothing)
    LAUNCH(S, KC_PRECON_FACTOR, k_coarse_assemble, dim3(nblk), dim3(256), S->cellmask, S->geom, S->chunk_list, S->sc, S->band_lo, S->coarse_shift,
           S->coarse_m, S->coarse_nx, S->cc_diag, S->cc_right, S->cc_up);
    LAUNCH(S, KC_PRECON_FACTOR, k_coarse_sten, dim3((n + 255) / 256), dim3(256), S->cc_diag, S->cc_right, S->cc_up, n, S->cc_sten);
    bw = S->coarse_ny > 1 ? S->coarse_nx : 1;      // half-bandwidth of P^T A P in row-major order of the coarse cells
    for (int k = 0; k < 5; ++k) st[k] = S->cc_sten + (size_t)k * n;
  }
  S->cc_n = n;
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_factor, dim3(1), dim3(CC_THREADS), st[0], st[1], st[2], st[3], st[4], n, nx, bw, S->cc_fac, S->sc, S->cc_pinned);
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_inverse, dim3((n + 3) / 4), dim3(256), S->cc_fac, n, bw, S->cc_inv, S->sc);
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_nullfix, dim3(1), dim3(CC_THREADS), S->cc_inv, st[0], S->cc_pinned, n, S->sc, S->cc_null);
  return EULER_OK;
}

// ---- per iteration: y = (P^T A P)^-1 P^T r and the epilogue of dot(z, r).  One workgroup per coarse cell gathers the tiles' partial
// sums of its cell (the bands of the coarse row x the tiles whose 79 columns touch the coarse column: ~300 entries for 16 x 16 coarse
// cells on 8192^2, 1.5 MB over all - one workgroup alone took 99 us for that) in a fixed assignment of entries to threads and a fixed
// fold, publishes r_c[c] and takes a ticket; the workgroup that draws the last ticket forms y = inv r_c (four threads per row, 0.5 MB
// out of L2), dot(z, r) += y . r_c and the scalar epilogue.  The hand-off is the one of block_finish (k_pcg.hip): 8-byte agent-scope
// atomics on both sides.  Nothing depends on which workgroup is last.
__device__ __forceinline__ double cc_block_sum(double v, double* s_red) {      // valid in thread 0; fixed order
  v = eu_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int k = 0; k < CC_THREADS / 64; ++k) t += s_red[k];
  return t;
}

__global__ __launch_bounds__(CC_THREADS) void k_coarse_solve(const double* __restrict__ part, const double* __restrict__ inv, double* __restrict__ y,
                                                             double* rc, unsigned int* counter, PcgScalars* sc, int fin_op, int force, int n, int nx,
                                                             int m, int shift, int ntb, int band_lo, int band_hi) {
  if (!force && (sc->done || !sc->nonzero)) return;      // (`done` may have been raised by this very iteration's max |r|)
  __shared__ double s_rc[CC_MAX], s_q[4][CC_MAX], s_red[CC_THREADS / 64];
  __shared__ int am_last;
  const int tid = threadIdx.x, c = blockIdx.x;
  {
    const int I = c / nx, J = c % nx, g = 1 << shift;
    const int klo = (J * g) >> 4;
    int khi = ((J + 1) * g - 1 + 63) >> 4;
    if (khi > ntb - 1) khi = ntb - 1;
    const int nk = khi - klo + 1;
    const int b0 = I * m > band_lo ? I * m : band_lo, b1 = (I + 1) * m < band_hi ? (I + 1) * m : band_hi;
    const int total = b1 > b0 ? (b1 - b0) * nk : 0;
    double sum = 0.0;
    for (int e = tid; e < total; e += CC_THREADS) {
      const int band = b0 + e / nk, k = klo + e % nk;
      const int xl = k * 16 - 63;
      const int b = J - ((xl > 0 ? xl : 0) >> shift);
      if (b >= 0 && b < 3) sum += part[((size_t)(band - band_lo) * ntb + k) * 3 + b];
    }
    sum = cc_block_sum(sum, s_red);
    if (tid == 0) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(&rc[c]), (unsigned long long)__double_as_longlong(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      am_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!am_last) return;
  }
  if (tid < CC_MAX)
    s_rc[tid] = tid < n ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&rc[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
  __syncthreads();
  const int row = tid & (CC_MAX - 1), q = tid >> 8;      // rows of the (symmetric) inverse by columns: coalesced over `row`; j = q, q + 4, ...
  double acc = 0.0;
  if (row < n)      // (16 loads in flight per thread: this workgroup alone pulls the 0.5 MB, latency-bound unless the loads overlap)
    for (int j = q; j < n; j += 64) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = j + 4 * u < n ? inv[(size_t)(j + 4 * u) * n + row] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u] * s_rc[(j + 4 * u) & (CC_MAX - 1)];
    }
  s_q[q][row] = acc;
  __syncthreads();
  double dv = 0.0;
  if (tid < n) {
    const double yv = (s_q[0][tid] + s_q[1][tid]) + (s_q[2][tid] + s_q[3][tid]);
    y[tid] = yv;
    dv = yv * s_rc[tid];
  }
  const double t = cc_block_sum(dv, s_red);
  if (tid == 0) {
    const double v = sc->sigma_new + t;      // k_precond_tile left dot(z_tile, r) there (FIN_STORE_ONLY)
    if (fin_op == CFIN_SIGMA_INIT) sc->sigma = v;                                                     // main.c:748
    else if (fin_op == CFIN_BETA) { sc->sigma_new = v; sc->beta = v / sc->sigma; sc->sigma = v; }     // main.c:762-765
    else sc->sigma_new = v;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next iteration
  }
}

int eu_launch_coarse_solve(euler_sim* S, int fin_op, int force) {
  if (eu_is_mg(S)) return eu_mg_solve(S, fin_op, force);
  LAUNCH(S, KC_COARSE_CYCLE, k_coarse_solve, dim3(S->coarse_n), dim3(CC_THREADS), S->cc_part, S->cc_inv, S->cc_y, S->cc_y + CC_MAX,
         reinterpret_cast<unsigned int*>(S->cc_y + 2 * CC_MAX), S->sc, fin_op, force, S->coarse_n, S->coarse_nx, S->coarse_m, S->coarse_shift,
         S->geom.T / 16, S->band_lo, S->band_hi);
  return EULER_OK;
}

// ---- the first search direction of a solve: s = z + P y (the memcpy at main.c:746, with the coarse part of z added on fluid cells), this rank's bands
__global__ __launch_bounds__(256) void k_coarse_search_init(double* __restrict__ s, const double* __restrict__ z, const uint8_t* __restrict__ mask,
                                                            const double* __restrict__ y, SkewGeom g, int shift, int nx, size_t e_lo, size_t e_cnt, const PcgScalars* sc) {
  if (sc->done || !sc->nonzero) return;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < e_cnt; k += (size_t)gridDim.x * blockDim.x) {
    const size_t e = e_lo + k;
    double v = z[e];
    if (mask[e] & CM_FLUID) {
      int band, t, l;
      skew_decode(g, e, band, t, l);
      v = v + y[(size_t)((band * 64 + l) >> shift) * nx + ((t - l) >> shift)];
    }
    s[e] = v;
  }
}

int eu_launch_coarse_search_init(euler_sim* S) {
  if (eu_is_mg(S)) return eu_mg_search_init(S);
  LAUNCH(S, KC_UPDATE_SEARCH, k_coarse_search_init, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), S->s, S->z, S->cellmask, S->cc_y, S->geom,
         S->coarse_shift, S->coarse_nx, S->e_lo, S->e_cnt, S->sc);
  return EULER_OK;
}

// ---- row slabs: a neighbour's edge row of z_0 (a compact row, one value per column) becomes a ghost row of the first search direction: + P y of its cells
__global__ __launch_bounds__(256) void k_coarse_add_row(double* __restrict__ row, const double* __restrict__ y, int X, int yrow, int shift, int nx, const PcgScalars* sc) {
  if (sc->done || !sc->nonzero) return;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < X) row[x] = row[x] + y[(size_t)(yrow >> shift) * nx + (x >> shift)];
}
int eu_launch_coarse_add_row(euler_sim* S, double* row, int yrow) {
  if (eu_is_mg(S)) return eu_mg_add_row(S, row, yrow);
  LAUNCH(S, KC_UPDATE_SEARCH, k_coarse_add_row, dim3((S->X + 255) / 256), dim3(256), row, S->cc_y, S->X, yrow, S->coarse_shift, S->coarse_nx, S->sc);
  return EULER_OK;
}
