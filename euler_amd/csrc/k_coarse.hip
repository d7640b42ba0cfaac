// k_coarse.hip — the coarse levels of the TWO-LEVEL and the MULTILEVEL preconditioner (EULER_PRECOND_IC0_TILE2 / _TILE_MG, include/euler.h; round 3):
//
//     z = M_tile^-1 r + P (P^T A P)^-1 P^T r            (two-level; the first part of this file)
//     z = M_tile^-1 r + P_0 V(P_0^T r)                  (multilevel: "Multilevel mode" below - a V-cycle over aggregates of 16, 32, ... cells whose top level is the
//                                                        dense level of the two-level mode; also on row slabs)
//
// M_tile = the tile-local IC(0) of k_pcg.hip (64-row x 16-record blocks, one pass over memory); P = piecewise constants over coarse
// cells of g x g grid cells (g = 64 m, a power of two; at most 256 coarse cells, 16 x 16 on a square grid) restricted to the fluid.
// No reference counterpart (the reference has ONE preconditioner, main.c:580-627): this is an extension of the roofline mode, restated
// in the oracle (eo_sim.coarse_m, coarse_correction) and compared with it to rounding.  Why: a block-local factor has no coupling
// beyond its block, and the first hundred iterations of a large solve - all the reference's cap ever allows, main.c:735 - live on the
// long-range part of A^-1 (the hydrostatic mode of a tank).  The coarse space has that part: DESIGN.md 5c.
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

#include <stdlib.h>

#define CC_MAX 256          // coarse cells
#define CC_THREADS 1024
#define CC_NULL_MAX 4       // fluid regions cut off from the air whose indicators are kept (k_coarse_nullfix)
#define CC_NULL_TOTAL (CC_NULL_MAX * CC_MAX + 1 + 2 * CC_NULL_MAX + 512 * 2 * CC_NULL_MAX + 1)      // = NS_TOTAL below
#define MG_DOT_BLOCKS 512    // most blocks of the level-0 kernel that closes the V-cycle (its dot partials)
// Round 4 scanned both on the GPU (tools/r04/mg_scan.py, mg_scan16k.py; iterations per solve to 1e-6 at 0.8 / 1.5, the values of round 3, and at 1.0 / 1.7: 8192^2 half
// tank 155 -> 128, 2048^2 dam break 43.2 -> 40.3, 2048^2 waterfall 129.5 -> 121.7; 1.8 - 2.0 are a little better still on deep water (4096^2 waterfall 165 -> 147 at
// 1.8) and worse on the thin sheet of a dam break's first frames after impact (16384^2: 90 -> 101 iterations over those frames): 1.7 sits between.  omega = 1 is the largest admissible value: the cycle is symmetric positive SEMI-definite for
// omega <= 2 / lambda_max(D^-1 A) (-> 1 on a large grid), and the tile-local part of the preconditioner is positive definite, so the sum stays SPD; at 1.1 PCG needs
// four times the iterations and at 1.2 it does not converge any more.
#ifndef MG_OMEGA
#define MG_OMEGA 1.0         // Jacobi (undamped)
#endif
#ifndef MG_KAPPA
#define MG_KAPPA 1.7         // scaling of the coarse-grid correction (plain aggregation under-corrects)
#endif

#define COMM_CALL(expr) do { if ((expr) != 0) { eu_set_error("communicator callback failed: %s", #expr); return EULER_ECOMM; } } while (0)

// the scalar epilogues of k_pcg.hip (same codes)
enum { CFIN_SIGMA_INIT = 0, CFIN_BETA = 3 };

void eu_coarse_release(euler_sim* S);
int eu_coarse_alloc(euler_sim* S) {
  if (S->mg_dot) return EULER_OK;      // (the last allocation below: everything is there)
  eu_coarse_release(S);                // (a failed earlier attempt may have left some of it)
  int m = 1, shift = 6;
  while (((S->X + 64 * m - 1) / (64 * m)) * ((S->geom.nbands + m - 1) / m) > CC_MAX) { m *= 2; shift += 1; }
  S->coarse_m = m; S->coarse_shift = shift;
  S->coarse_nx = (S->X + 64 * m - 1) / (64 * m);
  S->coarse_ny = (S->geom.nbands + m - 1) / m;
  S->coarse_n = S->coarse_nx * S->coarse_ny;
  const size_t n = (size_t)S->coarse_n;
  HIPCHK(hipMalloc((void**)&S->cc_diag, 4 * n * sizeof(int)));      // diagonal, right, up; pinned[n] (k_coarse_factor)
  S->cc_right = S->cc_diag + n; S->cc_up = S->cc_diag + 2 * n;
  HIPCHK(hipMalloc((void**)&S->cc_fac, n * n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_inv, n * n * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_part, (S->chunk_cap + 64) * 3 * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_y, (2 * CC_MAX + 1) * sizeof(double)));      // y [CC_MAX], r_c [CC_MAX], the ticket counter of k_coarse_solve
  HIPCHK(hipMemset(S->cc_part, 0, (S->chunk_cap + 64) * 3 * sizeof(double)));
  HIPCHK(hipMemset(S->cc_y, 0, (2 * CC_MAX + 1) * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->cc_null, CC_NULL_TOTAL * sizeof(double)));      // indicators, their number, the sums / partials / ticket of k_null_sums
  HIPCHK(hipMemset(S->cc_null, 0, CC_NULL_TOTAL * sizeof(double)));
  // the multilevel hierarchy below the dense level: aggregates of 16, 32, ... , 32 m grid cells (k_mg_* below)
  S->mg_levels = 0; S->mg_cells = 0;
  for (int g = 16; g < 64 * m; g *= 2) {
    const int l = S->mg_levels++;
    S->mg_nx[l] = (S->X + g - 1) / g; S->mg_ny[l] = (S->geom.nbands * 64 + g - 1) / g;      // (rows up to the last band's end: aggregates above Y stay empty)
    S->mg_off[l] = S->mg_cells; S->mg_cells += (size_t)S->mg_nx[l] * S->mg_ny[l];
  }
  HIPCHK(hipMalloc((void**)&S->mg_d, 3 * S->mg_cells * sizeof(int)));
  S->mg_rt = S->mg_d + S->mg_cells; S->mg_up = S->mg_d + 2 * S->mg_cells;
  HIPCHK(hipMalloc((void**)&S->mg_rhs, 3 * S->mg_cells * sizeof(double)));      // rhs, x, x1 (the Jacobi step) per level
  S->mg_x = S->mg_rhs + S->mg_cells;
  HIPCHK(hipMalloc((void**)&S->mg_part, (S->chunk_cap + 64) * 8 * sizeof(double)));
  HIPCHK(hipMalloc((void**)&S->mg_dot, (MG_DOT_BLOCKS + 1) * sizeof(double)));
  HIPCHK(hipMemset(S->mg_rhs, 0, 3 * S->mg_cells * sizeof(double)));
  HIPCHK(hipMemset(S->mg_part, 0, (S->chunk_cap + 64) * 8 * sizeof(double)));
  HIPCHK(hipMemset(S->mg_dot, 0, (MG_DOT_BLOCKS + 1) * sizeof(double)));
  return EULER_OK;
}

void eu_coarse_release(euler_sim* S) {
  if (S->cc_diag) (void)hipFree(S->cc_diag);
  for (double* d : {S->cc_fac, S->cc_inv, S->cc_part, S->cc_y, S->mg_rhs, S->mg_part, S->mg_dot, S->mg_xbuf, S->cc_null}) if (d) (void)hipFree(d);
  S->cc_null = nullptr;
  S->mg_xbuf = nullptr; S->mg_xslot = 0;
  if (S->mg_d) (void)hipFree(S->mg_d);
  S->cc_diag = S->cc_right = S->cc_up = nullptr;
  S->cc_fac = S->cc_inv = S->cc_part = S->cc_y = nullptr;
  S->mg_d = S->mg_rt = S->mg_up = nullptr; S->mg_rhs = S->mg_x = S->mg_part = S->mg_dot = nullptr;
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
__global__ __launch_bounds__(CC_THREADS) void k_coarse_factor(const int* __restrict__ cd, const int* __restrict__ cr, const int* __restrict__ cu,
                                                              int n, int nx, int bw, double* __restrict__ A, const PcgScalars* sc, int* __restrict__ pinned) {
  if (!sc->nonzero) return;
  __shared__ double s_band[CC_LDS_BAND];
  __shared__ double s_d;
  const int tid = threadIdx.x;
  const bool in_lds = n * (bw + 1) + bw <= CC_LDS_BAND;
  double* M = in_lds ? s_band : A;
  const int rs = in_lds ? bw : n, off = in_lds ? bw : 0;      // element (i, j), i - bw <= j <= i, at M[i * rs + j + off]
  for (int e = tid; e < n * n; e += CC_THREADS) A[e] = 0.0;
  if (in_lds) for (int e = tid; e < CC_LDS_BAND; e += CC_THREADS) s_band[e] = 0.0;
  __syncthreads();
  for (int c = tid; c < n; c += CC_THREADS) {
    M[c * rs + c + off] = cd[c] != 0 ? (double)cd[c] : 1.0;
    if ((c + 1) % nx != 0 && c + 1 < n) M[(c + 1) * rs + c + off] = (double)cr[c];
    if (c + nx < n) M[(c + nx) * rs + c + off] = (double)cu[c];
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const int w = k + bw < n - 1 ? bw : n - 1 - k;      // rows below the diagonal inside the band
    if (tid == 0) {
      // a fluid region cut off from the air (closed box, enclosed pool) makes A - and P^T A P - singular: the last pivot of such a component is rounding noise instead
      // of 0.  Like the reference's own factor (main.c:595) the pivot falls back to the matrix' diagonal then: that coarse cell is pinned, the operator stays positive definite
      double piv = M[k * rs + k + off];
      const double a_kk = cd[k] != 0 ? (double)cd[k] : 1.0;
      if (!(piv > 1e-8 * a_kk)) { piv = a_kk; if (cd[k] != 0) pinned[k] = 1; }      // (k_coarse_nullfix takes the pinned cell's component out of the inverse again)
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

// ==========================================================================================
// Multilevel mode (EULER_PRECOND_IC0_TILE_MG): z = M_tile^-1 r + P_0 V(P_0^T r).  Level l = aggregates of (16 << l)^2 grid cells; A_l as a
// 5-point stencil with integer entries (d, rt = coupling to the aggregate on the right, up = to the one above); the level above the last is
// the dense level of the two-level mode.  The CPU restatement the tests check it against (mg_build / mg_vcycle) uses the same formulas in the same order.
//
// A tile against the aggregates of 16: lane l = 16 j + i sits in aggregate row 4 band + j; its 16 records are the columns 16 (k - j) - i .. + 15:
// the first i of them lie in aggregate column k - j - 1 ("left"), the other 16 - i in column k - j ("right") - for all 16 lanes of the group alike,
// so a tile contributes to 4 x 2 aggregates and an aggregate (I, J) collects the right part of tile k = J + j and the left part of tile k = J + j + 1
// (j = I & 3) of its band.
__device__ __forceinline__ int group_sum_i(int v) {      // over the 16 lanes of a group
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void k_mg_assemble0(const uint8_t* __restrict__ mask, SkewGeom g, const unsigned int* __restrict__ list, const PcgScalars* sc,
                                                      int band_lo, int nx0, int* cd, int* cr, int* cu) {
  const int lane = threadIdx.x & 63, j = lane >> 4, i = lane & 15;
  const int ntb = g.T / 16, todo = (int)sc->n_chunks;
  const int n_waves = gridDim.x * 4;
  for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < todo; w += n_waves) {
    const int tile = (int)(list[w] & ~EU_CHUNK_INTERIOR);
    const int band = band_lo + tile / ntb, k = tile % ntb;
    const size_t base = ((size_t)band * g.TS + (size_t)k * 16) * 64 + 2 * lane;
    int d[2] = {0, 0}, r[2] = {0, 0}, u[2] = {0, 0};
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      const unsigned int mm = *reinterpret_cast<const unsigned short*>(mask + base + P * 128);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned int cm = (mm >> (8 * h)) & 0xff;
        if (!(cm & CM_FLUID)) continue;
        const int jj = 2 * P + h;                 // column 16 (k - j) + jj - i
        const int side = jj >= i;
        int dd = (int)(cm >> CM_DIAG_SHIFT), rr = 0, uu = 0;
        if (cm & CM_RIGHT) { if (((jj - i + 1) & 15) != 0) dd -= 2; else rr -= 1; }      // the cell to the right opens the next aggregate column
        if (cm & CM_UP) { if (i != 15) dd -= 2; else uu -= 1; }                             // the cell above opens the next aggregate row
        d[side] += dd; r[side] += rr; u[side] += uu;
      }
    }
#pragma unroll
    for (int sd = 0; sd < 2; ++sd) {
      const int td = group_sum_i(d[sd]), tr = group_sum_i(r[sd]), tu = group_sum_i(u[sd]);
      const int J = k - j - 1 + sd;
      if (i == 0 && J >= 0 && J < nx0) {
        const size_t c = (size_t)(4 * band + j) * nx0 + J;
        if (td) atomicAdd(&cd[c], td);
        if (tr) atomicAdd(&cr[c], tr);
        if (tu) atomicAdd(&cu[c], tu);
      }
    }
  }
}

// A_(l+1) = P^T A_l P for 2 x 2 aggregation: the couplings inside a parent count twice on its diagonal, those leaving it add up
__global__ __launch_bounds__(256) void k_mg_coarsen(const int* __restrict__ fd, const int* __restrict__ fr, const int* __restrict__ fu, int fnx, int fny,
                                                    int* __restrict__ cd, int* __restrict__ cr, int* __restrict__ cu, int cnx, int cny, const PcgScalars* sc) {
  if (!sc->nonzero) return;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= cnx * cny) return;
  const int I = p / cnx, J = p % cnx;
  int d = 0, r = 0, u = 0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int ci = 2 * I + a, cj = 2 * J + b;
      if (ci >= fny || cj >= fnx) continue;
      const size_t c = (size_t)ci * fnx + cj;
      d += fd[c];
      if (b) r += fr[c]; else d += 2 * fr[c];
      if (a) u += fu[c]; else d += 2 * fu[c];
    }
  cd[p] = d; cr[p] = r; cu[p] = u;
}

struct MgLevel { const int *d, *rt, *up; const double *rhs, *x1; int nx, ny; };      // x1 = the Jacobi step from zero, omega rhs / d (0 on empty aggregates), left by the kernel that wrote rhs
// v of a cell: x1, optionally + kappa e[parent] (0 on empty aggregates)
template <bool WITH_E>
__device__ __forceinline__ double mg_val(const MgLevel& L, const double* __restrict__ e, int enx, int I, int J) {
  const size_t c = (size_t)I * L.nx + J;
  if (!WITH_E) return L.x1[c];
  return L.d[c] ? L.x1[c] + MG_KAPPA * e[(size_t)(I >> 1) * enx + (J >> 1)] : 0.0;
}
// (A_l v)[c]: diagonal, right, left, up, down - the oracle's order (mg_apply)
template <bool WITH_E>
__device__ __forceinline__ double mg_apply(const MgLevel& L, const double* __restrict__ e, int enx, int I, int J, double vc) {
  const size_t c = (size_t)I * L.nx + J;
  double t = (double)L.d[c] * vc;
  if (J + 1 < L.nx) t = t + (double)L.rt[c] * mg_val<WITH_E>(L, e, enx, I, J + 1);
  if (J > 0) t = t + (double)L.rt[c - 1] * mg_val<WITH_E>(L, e, enx, I, J - 1);
  if (I + 1 < L.ny) t = t + (double)L.up[c] * mg_val<WITH_E>(L, e, enx, I + 1, J);
  if (I > 0) t = t + (double)L.up[c - L.nx] * mg_val<WITH_E>(L, e, enx, I - 1, J);
  return t;
}

// level-0 right-hand side from the tiles' partial sums (right part of tile J + j, then left part of tile J + j + 1), and its Jacobi step.
// Row slabs: rows [row0, row1) only, into this rank's slot of the exchange buffer (x10 null: k_mg_scatter0 forms the Jacobi step once every rank's rows are there)
__global__ __launch_bounds__(256) void k_mg_gather0(const double* __restrict__ part, const int* __restrict__ d0, double* __restrict__ dst, double* __restrict__ x10,
                                                    int nx0, int row0, int row1, int ntb, int band_lo, int band_hi, const PcgScalars* sc, int force) {
  (void)force; (void)sc;      // (no early exit on `done`: see launch_mg_cycle)
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nx0 * (row1 - row0)) return;
  const int I = row0 + k / nx0, J = k % nx0, band = I >> 2, j = I & 3;
  double t = 0.0;
  if (band >= band_lo && band < band_hi) {
    const size_t row = (size_t)(band - band_lo) * ntb;
    const int k0 = J + j, k1 = J + j + 1;
    if (k0 < ntb) t = part[(row + k0) * 8 + j * 2 + 1];
    if (k1 < ntb) t = t + part[(row + k1) * 8 + j * 2];
  }
  dst[k] = t;
  if (x10) { const int d = d0[(size_t)I * nx0 + J]; x10[k] = d ? MG_OMEGA * t / (double)d : 0.0; }
}

// row slabs: every rank's rows of rhs_0 arrived in its slot of the exchange buffer ({max |r|, dot} first); rhs_0 whole, and its Jacobi step
struct MgParts { int n; int lo[64], hi[64]; };      // aggregate rows [lo, hi) per rank
__global__ __launch_bounds__(256) void k_mg_scatter0(const double* __restrict__ xbuf, int slot, MgParts P, const int* __restrict__ d0, double* __restrict__ rhs0, double* __restrict__ x10,
                                                     int nx0, int n0, const PcgScalars* sc, int force) {
  (void)force; (void)sc;      // (no early exit on `done`: see launch_mg_cycle)
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n0) return;
  const int I = c / nx0, J = c % nx0;
  double t = 0.0;
  for (int r = 0; r < P.n; ++r)
    if (I >= P.lo[r] && I < P.hi[r]) { t = xbuf[(size_t)r * slot + 2 + (size_t)(I - P.lo[r]) * nx0 + J]; break; }
  rhs0[c] = t;
  const int d = d0[c];
  x10[c] = d ? MG_OMEGA * t / (double)d : 0.0;
}

// down: the right-hand side of level l + 1 = the restricted residual of level l after its Jacobi step, children in the order (2I, 2J),
// (2I, 2J + 1), (2I + 1, 2J), (2I + 1, 2J + 1).  A thread per CHILD (four neighbouring lanes per parent; a thread per parent left one wave
// per SIMD with 40 dependent loads each: 25 us at level 0 of 8192^2); the first of the four adds them up in that order and leaves the
// parent's Jacobi step as well.
__global__ __launch_bounds__(256) void k_mg_down(MgLevel L, double* __restrict__ crhs, double* __restrict__ cx1, const int* __restrict__ cd, int cnx, int cny,
                                                 const PcgScalars* sc, int force) {
  (void)force; (void)sc;      // (no early exit on `done`: see launch_mg_cycle)
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int p = tid >> 2, q = tid & 3;
  double res = 0.0;
  if (p < cnx * cny) {
    const int ci = 2 * (p / cnx) + (q >> 1), cj = 2 * (p % cnx) + (q & 1);
    if (ci < L.ny && cj < L.nx) {      // (every load below is issued before the first result is needed: no load waits for another's value)
      const size_t c = (size_t)ci * L.nx + cj;
      const int dc = L.d[c];
      const double t = L.rhs[c] - mg_apply<false>(L, nullptr, 0, ci, cj, L.x1[c]);
      res = dc ? t : 0.0;
    }
  }
  const int base = (threadIdx.x & 63) & ~3;
  const double r0 = __shfl(res, base, 64), r1 = __shfl(res, base + 1, 64), r2 = __shfl(res, base + 2, 64), r3 = __shfl(res, base + 3, 64);
  if (q == 0 && p < cnx * cny) {
    double t = 0.0;
    t = t + r0; t = t + r1; t = t + r2; t = t + r3;
    crhs[p] = t;
    if (cx1) { const int d = cd[p]; cx1[p] = d ? MG_OMEGA * t / (double)d : 0.0; }
  }
}

// the dense level: y = (P^T A P)^-1 rhs, a workgroup per row (one workgroup alone took 15 us for the 0.5 MB)
__global__ __launch_bounds__(256) void k_mg_top(const double* __restrict__ rhs, const double* __restrict__ inv, double* __restrict__ y, int n,
                                                const PcgScalars* sc, int force) {
  (void)force; (void)sc;      // (no early exit on `done`: see launch_mg_cycle)
  __shared__ double s_red[4];
  const int row = blockIdx.x, j = threadIdx.x;
  double v = j < n ? inv[(size_t)row * n + j] * rhs[j] : 0.0;      // (the inverse is symmetric: row = column)
  v = eu_wave_sum(v);
  if ((j & 63) == 0) s_red[j >> 6] = v;
  __syncthreads();
  if (j == 0) y[row] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// up: x = x2 + omega (rhs - A x2) / d with x2 = the Jacobi step + kappa e[parent], e = the level above's result; a thread per cell.
// LAST (level 0): also the share of dot(z, r) the correction adds, x . rhs, folded per block and by the block that draws the last ticket
// (block_finish's hand-off), with the scalar epilogue k_precond_tile left open.
template <bool LAST>
__global__ __launch_bounds__(256) void k_mg_up(MgLevel L, const double* __restrict__ e, int enx, double* __restrict__ x, PcgScalars* sc, int fin_op, int force,
                                               double* dot_part, unsigned int* counter) {
  const bool idle = !force && (sc->done || !sc->nonzero);      // read first, consulted last: the load overlaps the level's own loads
  double dv = 0.0;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < L.nx * L.ny; c += gridDim.x * 256) {
    const int I = c / L.nx, J = c % L.nx;
    const int d = L.d[c];
    const double x2 = mg_val<true>(L, e, enx, I, J);
    const double t = x2 + MG_OMEGA * (L.rhs[c] - mg_apply<true>(L, e, enx, I, J, x2)) / (double)(d ? d : 1);      // (loads issued together; an empty aggregate selects 0)
    const double xv = d ? t : 0.0;
    x[c] = xv;
    if (LAST) dv += xv * L.rhs[c];
  }
  if (!LAST || idle) return;      // (after convergence the level arrays are dead; the scalar epilogue below must not run)
  __shared__ double s_red[4];
  __shared__ int am_last;
  dv = eu_wave_sum(dv);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = dv;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&dot_part[blockIdx.x]), (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    am_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!am_last) return;
  double t = 0.0;
  for (unsigned int k = threadIdx.x; k < gridDim.x; k += 256)
    t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&dot_part[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  t = eu_wave_sum(t);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = sc->sigma_new + ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));      // k_precond_tile left dot(z_tile, r) there (FIN_STORE_ONLY)
    if (fin_op == CFIN_SIGMA_INIT) sc->sigma = v;                                                     // main.c:748
    else if (fin_op == CFIN_BETA) { sc->sigma_new = v; sc->beta = v / sc->sigma; sc->sigma = v; }     // main.c:762-765
    else sc->sigma_new = v;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#define CLAUNCH(S, KERNEL, GRID, BLOCK, ...) hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, (S)->stream, __VA_ARGS__)
// The kernels of the cycle do NOT start with the usual `if (done) return`: that test is a memory round trip (~1.5 us) in front of launches that take ~3 us
// themselves.  Past convergence they recompute level arrays nobody reads again (the fine-grid kernels do return at once, and k_mg_up<true> keeps its
// scalar epilogue behind the flag, which it loads first and consults last).
static int launch_mg_cycle(euler_sim* S, int fin_op, int force) {
  const int nl = S->mg_levels;
  eu_prof_begin(S, KC_COARSE_CYCLE);      // ONE event pair around the whole cycle (a pair per 5 us launch would time the events)
  double* x1 = S->mg_x + S->mg_cells;      // the third pool
  auto level = [&](int l) { const size_t o = S->mg_off[l]; return MgLevel{S->mg_d + o, S->mg_rt + o, S->mg_up + o, S->mg_rhs + o, x1 + o, S->mg_nx[l], S->mg_ny[l]}; };
  const int n0 = S->mg_nx[0] * S->mg_ny[0];
  if (S->has_comm) {      // row slabs: the rows travelled inside the G1 exchange (eu_launch_coarse_pre); the V-cycle itself runs replicated (identical bits everywhere)
    MgParts P;
    P.n = S->bulk.nranks < 64 ? S->bulk.nranks : 64;
    for (int r = 0; r < P.n; ++r) { P.lo[r] = 4 * S->part_lo[r]; P.hi[r] = 4 * S->part_hi[r]; }
    CLAUNCH(S, k_mg_scatter0, dim3((n0 + 255) / 256), dim3(256), S->mg_xbuf, S->mg_xslot, P, S->mg_d, S->mg_rhs, x1, S->mg_nx[0], n0, S->sc, force);
  } else
    CLAUNCH(S, k_mg_gather0, dim3((n0 + 255) / 256), dim3(256), S->mg_part, S->mg_d, S->mg_rhs, x1, S->mg_nx[0], 0, S->mg_ny[0], S->geom.T / 16,
           S->band_lo, S->band_hi, S->sc, force);
  double* top_rhs = S->cc_y + CC_MAX;
  for (int l = 0; l < nl; ++l) {
    const bool top = l + 1 == nl;
    const int cnx = top ? S->coarse_nx : S->mg_nx[l + 1], cny = top ? S->coarse_ny : S->mg_ny[l + 1];
    const size_t co = top ? 0 : S->mg_off[l + 1];
    CLAUNCH(S, k_mg_down, dim3(((size_t)cnx * cny * 4 + 255) / 256), dim3(256), level(l), top ? top_rhs : S->mg_rhs + co, top ? (double*)nullptr : x1 + co,
           top ? (const int*)nullptr : S->mg_d + co, cnx, cny, S->sc, force);
  }
  CLAUNCH(S, k_mg_top, dim3(S->coarse_n), dim3(256), top_rhs, S->cc_inv, S->cc_y, S->coarse_n, S->sc, force);
  for (int l = nl - 1; l >= 0; --l) {
    const bool top = l + 1 == nl;
    const double* e = top ? S->cc_y : S->mg_x + S->mg_off[l + 1];
    const int enx = top ? S->coarse_nx : S->mg_nx[l + 1];
    const size_t cells = (size_t)S->mg_nx[l] * S->mg_ny[l];
    if (l > 0) CLAUNCH(S, k_mg_up<false>, dim3((unsigned)((cells + 255) / 256)), dim3(256), level(l), e, enx, S->mg_x + S->mg_off[l], S->sc, fin_op, force, (double*)nullptr, (unsigned int*)nullptr);
    else {
      const unsigned nb = (unsigned)((cells + 255) / 256) < MG_DOT_BLOCKS ? (unsigned)((cells + 255) / 256) : MG_DOT_BLOCKS;
      CLAUNCH(S, k_mg_up<true>, dim3(nb), dim3(256), level(l), e, enx, S->mg_x, S->sc, fin_op, force, S->mg_dot, reinterpret_cast<unsigned int*>(S->mg_dot + MG_DOT_BLOCKS));
    }
  }
  eu_prof_end(S, KC_COARSE_CYCLE);
  return EULER_OK;
}

// ---- fluid cut off from the air: P^T A P has the indicator n of every such component in its null space, the factor pinned one cell of it (k_coarse_factor) and the inverse
// S of the pinned matrix treats the component lopsidedly - PCG then stalls on the part of r along n that no A s can touch.  With n = a_kk S e_k (the pinned system's answer
// to the pin itself: exactly the indicator) the inverse becomes (I - n n^T / n.n) S (I - n n^T / n.n): the pseudo-inverse of P^T A P - zero along n, S elsewhere.
__global__ __launch_bounds__(CC_THREADS) void k_coarse_nullfix(double* __restrict__ inv, const int* __restrict__ cd, const int* __restrict__ pinned, int n, const PcgScalars* sc,
                                                               double* __restrict__ nullv) {
  const int tid = threadIdx.x;
  if (tid == 0) nullv[CC_NULL_MAX * CC_MAX] = 0.0;
  if (!sc->nonzero) return;
  __shared__ double s_n[CC_MAX], s_w[CC_MAX], s_red[CC_THREADS / 64];
  __shared__ double s_nn, s_nw;
  int found = 0;
  for (int k = 0; k < n; ++k) {
    if (!pinned[k]) continue;      // (uniform)
    const double a_kk = (double)cd[k];
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
__global__ __launch_bounds__(256) void k_null_sums(const double* __restrict__ r, const uint8_t* __restrict__ mask, SkewGeom g, size_t e_lo, size_t e_cnt,
                                                   double* __restrict__ nullv, int shift, int nx, const PcgScalars* sc) {
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
    const size_t c = (size_t)((band * 64 + l) >> shift) * nx + ((t - l) >> shift);
    const double rv = r[e];
#pragma unroll
    for (int q = 0; q < CC_NULL_MAX; ++q)
      if (q < count) { const double w = nullv[q * CC_MAX + c]; acc[2 * q] += rv * w; acc[2 * q + 1] += w * w; }
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
                                                    const double* __restrict__ nullv, int shift, int nx, const PcgScalars* sc) {
  const int count = (int)nullv[CC_NULL_MAX * CC_MAX];
  if (!sc->nonzero || count <= 0) return;
  double f[CC_NULL_MAX];
#pragma unroll
  for (int q = 0; q < CC_NULL_MAX; ++q) { const double nn = nullv[NS_SUMS + 2 * q + 1]; f[q] = (q < count && nn > 0.0) ? nullv[NS_SUMS + 2 * q] / nn : 0.0; }
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < e_cnt; k += (size_t)gridDim.x * 256) {
    const size_t e = e_lo + k;
    if (!(mask[e] & CM_FLUID)) continue;
    int band, t, l;
    skew_decode(g, e, band, t, l);
    const size_t c = (size_t)((band * 64 + l) >> shift) * nx + ((t - l) >> shift);
    double v = r[e];
#pragma unroll
    for (int q = 0; q < CC_NULL_MAX; ++q)
      if (q < count) v = v - nullv[q * CC_MAX + c] * f[q];
    r[e] = v;
  }
}
int eu_launch_coarse_consistent(euler_sim* S) {
  const unsigned nblk = eu_blocks(S->e_cnt, 256 * 8, NULL_BLOCKS);
  LAUNCH(S, KC_PRECON_FACTOR, k_null_sums, dim3(nblk), dim3(256), S->r, S->cellmask, S->geom, S->e_lo, S->e_cnt, S->cc_null, S->coarse_shift, S->coarse_nx, S->sc);
  if (S->has_comm) COMM_CALL(S->bulk.allreduce(S->bulk.ctx, S->cc_null + NS_SUMS, 2 * CC_NULL_MAX, 0));      // (zeros when nothing is cut off: every rank calls it, every solve)
  LAUNCH(S, KC_PRECON_FACTOR, k_null_apply, dim3(nblk), dim3(256), S->r, S->cellmask, S->geom, S->e_lo, S->e_cnt, S->cc_null, S->coarse_shift, S->coarse_nx, S->sc);
  return EULER_OK;
}

int eu_coarse_comm_slots(euler_sim* S) {
  int rows = 0;
  for (int r = 0; r < S->bulk.nranks && r < 64; ++r) rows = 4 * (S->part_hi[r] - S->part_lo[r]) > rows ? 4 * (S->part_hi[r] - S->part_lo[r]) : rows;
  const int slot = 2 + rows * S->mg_nx[0];
  if (slot != S->mg_xslot || !S->mg_xbuf) {
    if (S->mg_xbuf) { if (hipStreamSynchronize(S->stream) != hipSuccess || hipFree(S->mg_xbuf) != hipSuccess) return -1; S->mg_xbuf = nullptr; }
    if (hipMalloc((void**)&S->mg_xbuf, (size_t)slot * S->bulk.nranks * sizeof(double)) != hipSuccess) { eu_set_error("hipMalloc of the multilevel exchange buffer failed"); return -1; }
    if (hipMemsetAsync(S->mg_xbuf, 0, (size_t)slot * S->bulk.nranks * sizeof(double), S->stream) != hipSuccess) return -1;
    S->mg_xslot = slot;
  }
  return slot;
}
int eu_launch_coarse_pre(euler_sim* S, int force) {
  const int row0 = 4 * S->band_lo, row1 = 4 * S->band_hi;
  const int cells = (row1 - row0) * S->mg_nx[0];
  if (cells > 0)
    LAUNCH(S, KC_COARSE_CYCLE, k_mg_gather0, dim3((cells + 255) / 256), dim3(256), S->mg_part, S->mg_d, S->mg_xbuf + (size_t)S->bulk.rank * S->mg_xslot + 2, (double*)nullptr,
           S->mg_nx[0], row0, row1, S->geom.T / 16, S->band_lo, S->band_hi, S->sc, force);
  return EULER_OK;
}

int eu_launch_coarse_setup(euler_sim* S) {
  const int n = S->coarse_n;
  HIPCHK(hipMemsetAsync(S->cc_diag, 0, 4 * (size_t)n * sizeof(int), S->stream));
  const unsigned nblk = eu_blocks(S->chunk_cap, 4, 2048);
  if (eu_is_mg(S) && S->mg_levels > 0) {      // level 0 from the cells, every further level (and the dense one) from the level below
    HIPCHK(hipMemsetAsync(S->mg_d, 0, 3 * S->mg_cells * sizeof(int), S->stream));
    HIPCHK(hipMemsetAsync(S->mg_part, 0, (S->chunk_cap + 64) * 8 * sizeof(double), S->stream));      // (tiles outside this solve's list contribute nothing)
    LAUNCH(S, KC_PRECON_FACTOR, k_mg_assemble0, dim3(nblk), dim3(256), S->cellmask, S->geom, S->chunk_list, S->sc, S->band_lo, S->mg_nx[0], S->mg_d, S->mg_rt, S->mg_up);
    if (S->has_comm) {      // row slabs: an aggregate of 16 rows belongs to one rank (slabs are cut at band boundaries) - every rank contributes its rows of A_0, then all of them build the same hierarchy
      int64_t off[64], cnt[64];
      for (int r = 0; r < S->bulk.nranks && r < 64; ++r) { off[r] = (int64_t)4 * S->part_lo[r] * S->mg_nx[0] * sizeof(int); cnt[r] = (int64_t)4 * (S->part_hi[r] - S->part_lo[r]) * S->mg_nx[0] * sizeof(int); }
      COMM_CALL(S->bulk.allgather(S->bulk.ctx, S->mg_d, off, cnt));
      COMM_CALL(S->bulk.allgather(S->bulk.ctx, S->mg_rt, off, cnt));
      COMM_CALL(S->bulk.allgather(S->bulk.ctx, S->mg_up, off, cnt));
    }
    for (int l = 1; l <= S->mg_levels; ++l) {
      const bool top = l == S->mg_levels;
      const int cnx = top ? S->coarse_nx : S->mg_nx[l], cny = top ? S->coarse_ny : S->mg_ny[l];
      const size_t fo = S->mg_off[l - 1], co = top ? 0 : S->mg_off[l];
      LAUNCH(S, KC_PRECON_FACTOR, k_mg_coarsen, dim3(((size_t)cnx * cny + 255) / 256), dim3(256), S->mg_d + fo, S->mg_rt + fo, S->mg_up + fo, S->mg_nx[l - 1], S->mg_ny[l - 1],
             top ? S->cc_diag : S->mg_d + co, top ? S->cc_right : S->mg_rt + co, top ? S->cc_up : S->mg_up + co, cnx, cny, S->sc);
    }
  } else {
  HIPCHK(hipMemsetAsync(S->cc_part, 0, (S->chunk_cap + 64) * 3 * sizeof(double), S->stream));      // (tiles outside this solve's list contribute nothing)
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_assemble, dim3(nblk), dim3(256), S->cellmask, S->geom, S->chunk_list, S->sc, S->band_lo, S->coarse_shift,
         S->coarse_m, S->coarse_nx, S->cc_diag, S->cc_right, S->cc_up);
  }
  const int bw = S->coarse_ny > 1 ? S->coarse_nx : 1;      // half-bandwidth of P^T A P in row-major order of the coarse cells
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_factor, dim3(1), dim3(CC_THREADS), S->cc_diag, S->cc_right, S->cc_up, n, S->coarse_nx, bw, S->cc_fac, S->sc, S->cc_diag + 3 * (size_t)n);
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_inverse, dim3((n + 3) / 4), dim3(256), S->cc_fac, n, bw, S->cc_inv, S->sc);
  LAUNCH(S, KC_PRECON_FACTOR, k_coarse_nullfix, dim3(1), dim3(CC_THREADS), S->cc_inv, S->cc_diag, S->cc_diag + 3 * (size_t)n, n, S->sc, S->cc_null);
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
  if (eu_is_mg(S) && S->mg_levels > 0) return launch_mg_cycle(S, fin_op, force);
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
  const bool mg = eu_is_mg(S) && S->mg_levels > 0;
  LAUNCH(S, KC_UPDATE_SEARCH, k_coarse_search_init, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), S->s, S->z, S->cellmask, mg ? S->mg_x : S->cc_y, S->geom,
         mg ? 4 : S->coarse_shift, mg ? S->mg_nx[0] : S->coarse_nx, S->e_lo, S->e_cnt, S->sc);
  return EULER_OK;
}

// ---- row slabs: a neighbour's edge row of z_0 (a compact row, one value per column) becomes a ghost row of the first search direction: + P y of its cells
__global__ __launch_bounds__(256) void k_coarse_add_row(double* __restrict__ row, const double* __restrict__ y, int X, int yrow, int shift, int nx, const PcgScalars* sc) {
  if (sc->done || !sc->nonzero) return;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < X) row[x] = row[x] + y[(size_t)(yrow >> shift) * nx + (x >> shift)];
}
int eu_launch_coarse_add_row(euler_sim* S, double* row, int yrow) {
  const bool mg = eu_is_mg(S) && S->mg_levels > 0;
  LAUNCH(S, KC_UPDATE_SEARCH, k_coarse_add_row, dim3((S->X + 255) / 256), dim3(256), row, mg ? S->mg_x : S->cc_y, S->X, yrow, mg ? 4 : S->coarse_shift, mg ? S->mg_nx[0] : S->coarse_nx, S->sc);
  return EULER_OK;
}
