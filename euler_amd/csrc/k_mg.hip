// k_mg.hip — the coarse part of the MULTILEVEL preconditioner (EULER_PRECOND_IC0_TILE_MG, include/euler.h; round 5):
//
//     z = M_tile^-1 r + P_0 V(P_0^T r)
//
// M_tile = the tile-local IC(0) of k_pcg.hip.  Level 0 = a grid of NODES, node (I, J) at the centre of grid cell (8 J + 4, 8 I + 4) (MG_G0 = 8, k_mg.h: two node columns per
// tile width, eight node rows per band); P_0 = bilinear interpolation from the four nodes around a cell (weights in eighths), restricted to the fluid, constant beyond the
// outermost nodes.  Level l + 1 = every other node of level l in both directions (its node J sits ON node 2 J), bilinear again: weights 1, 1/2 - full weighting.  Because nodes
// sit AT cell centres / on finer nodes, a hat is 0 at the neighbouring nodes and the Galerkin operators A_0 = P_0^T A P_0, A_(l+1) = P^T A_l P are exact NINE-POINT
// stencils (a[k][c]: the entry that couples node c = (I, J) to node (I + k / 3 - 1, J + k % 3 - 1); A_0's entries are multiples of 2^-12: integer sums, exact in any order).
// V = one symmetric V-cycle: damped Jacobi (per-node omega, k_mg.h) from a zero guess, restricted residual, recursion, correction, Jacobi again; the top level (<= 64 nodes) is
// the dense pseudo-inverse of k_coarse.hip (factor, inverse, null-space fix for water cut off from the air).
//
// History: rounds 3-4 used piecewise constants over 16 x 16 blocks (aggregation: 5-point stencils with integer entries, the correction scaled by 1.7) and ONE launch per
// level and direction (12 launches, 63 us per iteration at 8192^2).  Bilinear spaces on nodes 16 cells apart halved the iteration count, nodes 8 cells apart nearly halved it
// again (tank at rest to 1e-6: 104 -> 52 -> 30; tools/r05/mg_proto.py) for a level 0 of four times the nodes.  The cycle runs in FOUR to SIX launches whatever the depth:
//
//   k_mg_down1<true>  (level 0 of more than MG_SMALL_LEVEL0 nodes) gathers the level-0 right-hand side from the tiles' partial sums (k_precond_tile leaves MG_PART = 72
//                     doubles per tile: [group of 8 lanes][row slot][column slot], k_mg.h) and takes it down ONE level on workgroups of 256 threads
//   k_mg_down<..>     the next (up to three) level transitions per launch; every workgroup owns a tile of the OUTPUT level and recomputes the halo it needs of the levels
//                     below in LDS (Jacobi step, residual, restriction: a level transition is three LDS phases, no launch); the workgroup that draws the last ticket of the
//                     launch that reaches the entry level then runs every level of <= 1024 nodes - down, the dense top, up again - alone, out of LDS and registers
//   k_mg_up           from that level's result back to level 1 (level 0 on small grids): every workgroup owns a 32 x 32 tile and recomputes the halos of the coarser levels
//   k_mg_up0          level 1 -> level 0 on small workgroups; its last workgroup folds x_0 . rhs_0 into dot(z, r) and applies the scalar epilogue
// Every one of them leaves at once behind convergence (sc->done; round 6).
//
// The CPU restatement the tests check all of this against: the test oracle's mg_build / mg_vcycle (same formulas; sums in another order: agreement to rounding).
// No reference counterpart (the reference has ONE preconditioner, main.c:580-627).
#include "euler_dev.h"
#include "k_mg.h"

#define COMM_CALL(expr) do { if ((expr) != 0) { eu_set_error("communicator callback failed: %s", #expr); return EULER_ECOMM; } } while (0)

enum { MFIN_SIGMA_INIT = 0, MFIN_BETA = 3 };      // the scalar epilogues of k_pcg.hip (same codes)

// ------------------------------------------------------------------------------------------ hierarchy
struct MgHier {
  int nl;                          // levels, the dense top included
  int nx[MG_MAXLEV], ny[MG_MAXLEV];
  unsigned int off[MG_MAXLEV];     // first node of level l in the pooled arrays
  const double* a;                 // stencils: level l at a + 9 * off[l], entry k of node c at [k * n_l + c]
  double* rhs;
  double* x;
  const double* wd;                // omega / diagonal per node (0 where the node carries no fluid): the Jacobi steps multiply
  const uint8_t* inner0;           // level 0: 1 where a node's stencil is the one of deep water (`ic`), 2 where it is all zeros (no fluid under the node): the two level-0 kernels then load a byte instead of nine doubles
  double ic[9], icwd;
};
static MgHier mg_hier(const euler_sim* S) {
  MgHier H;
  H.nl = S->mg_levels;
  for (int l = 0; l < MG_MAXLEV; ++l) { H.nx[l] = l < H.nl ? S->mg_nx[l] : 0; H.ny[l] = l < H.nl ? S->mg_ny[l] : 0; H.off[l] = l < H.nl ? (unsigned int)S->mg_off[l] : 0u; }
  H.a = S->mg_a; H.rhs = S->mg_rhs; H.x = S->mg_x; H.wd = S->mg_wd;
  H.inner0 = S->mg_inner0;
  for (int k = 0; k < 9; ++k) H.ic[k] = S->mg_ic[k];
  {      // deep water's omega_i / d (k_mg_wd's expressions: a node flagged "inner" must get the bits its own entries would give)
    const double d = S->mg_ic[4];
    double off = 0.0;
    for (int k = 0; k < 9; ++k) if (k != 4) off = off + fabs(S->mg_ic[k]);
    double om = d != 0.0 ? MG_THETA / (1.0 + off / d) : 0.0;
    if (om > MG_OMEGA) om = MG_OMEGA;
    H.icwd = d != 0.0 ? om / d : 0.0;
  }
  return H;
}
__device__ __forceinline__ const double* mg_sten(const MgHier& H, int l) { return H.a + 9 * (size_t)H.off[l]; }

void eu_mg_release(euler_sim* S);
// (a failed allocation gives back what the attempt already holds: the next attempt starts from nothing - ADVICE r5)
#define MGCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { eu_mg_release(S); return eu_hip_fail(_e, #call, __FILE__, __LINE__); } } while (0)
int eu_mg_alloc(euler_sim* S) {
  if (S->mg_dot) return EULER_OK;
  int nx = (S->X + MG_G0 - 1) / MG_G0, ny = MG_RPB * S->geom.nbands, l = 0;
  S->mg_cells = 0;
  for (;; ++l) {
    if (l >= MG_MAXLEV) { eu_set_error("multilevel preconditioner: more than %d levels", MG_MAXLEV); return EULER_EINVAL; }
    S->mg_nx[l] = nx; S->mg_ny[l] = ny; S->mg_off[l] = S->mg_cells; S->mg_cells += (size_t)nx * ny;
    if (nx * ny <= MG_TOP_MAX) break;
    nx = (nx + 1) / 2; ny = (ny + 1) / 2;
  }
  S->mg_levels = l + 1;
  const size_t n0 = (size_t)S->mg_nx[0] * S->mg_ny[0];
  {      // the limits of the cycle's launches, checked where the mode is SELECTED (euler_create, euler_set_precond) - not inside a solve, behind the substep's marker stages
    int lC = 0;      // (mg_entry_level)
    while (lC < S->mg_levels - 1 && (size_t)S->mg_nx[lC] * S->mg_ny[lC] > 1024 /* MG_TAIL_MAX */) ++lC;
    const size_t tiles0 = (size_t)((S->mg_nx[0] + 31) / 32) * ((S->mg_ny[0] + 31) / 32);
    if (S->mg_levels - 1 - lC > 4 /* MG_TAIL_LEVELS */) { eu_set_error("multilevel preconditioner: %d levels behind the entry level (at most 4): the grid is too large for this mode", S->mg_levels - 1 - lC); S->mg_levels = 0; return EULER_EINVAL; }
    if (tiles0 > MG_DOT_BLOCKS) { eu_set_error("multilevel preconditioner: %zu tiles of level 0 (at most %d): the grid is too large for this mode", tiles0, MG_DOT_BLOCKS); S->mg_levels = 0; return EULER_EINVAL; }
  }
  MGCHK(hipMalloc((void**)&S->mg_a, 9 * S->mg_cells * sizeof(double)));
  MGCHK(hipMalloc((void**)&S->mg_a0i, 9 * n0 * sizeof(unsigned long long)));
  MGCHK(hipMalloc((void**)&S->mg_rhs, 3 * S->mg_cells * sizeof(double)));
  S->mg_x = S->mg_rhs + S->mg_cells; S->mg_wd = S->mg_rhs + 2 * S->mg_cells;
  MGCHK(hipMalloc((void**)&S->mg_part, (S->chunk_cap + 64) * MG_PART * sizeof(double)));
  MGCHK(hipMalloc((void**)&S->mg_null0, (size_t)MG_NULL_MAX * S->mg_cells * sizeof(double)));      // the indicators of cut-off regions on every level (k_mg_null_prolong)
  MGCHK(hipMemset(S->mg_a, 0, 9 * S->mg_cells * sizeof(double)));
  MGCHK(hipMemset(S->mg_rhs, 0, 3 * S->mg_cells * sizeof(double)));
  MGCHK(hipMemset(S->mg_part, 0, (S->chunk_cap + 64) * MG_PART * sizeof(double)));
  MGCHK(hipMemset(S->mg_null0, 0, (size_t)MG_NULL_MAX * S->mg_cells * sizeof(double)));
  MGCHK(hipMalloc((void**)&S->mg_inner0, n0));
  MGCHK(hipMemset(S->mg_inner0, 0, n0));
  {      // level 0's stencil under deep water: A_0 = K (x) M + M (x) K with the 1-D mass and stiffness sums of the hats (multiples of 1 / G0^2: exact)
    double M[3] = {0, 0, 0}, K[3] = {-1.0 / MG_G0, 2.0 / MG_G0, -1.0 / MG_G0};
    for (int k = -(MG_G0 - 1); k <= MG_G0 - 1; ++k) { const double w = 1.0 - (k < 0 ? -k : k) / (double)MG_G0; M[1] += w * w; }
    for (int k = 0; k < MG_G0; ++k) M[0] += (1.0 - k / (double)MG_G0) * (k / (double)MG_G0);
    M[2] = M[0];
    for (int q = 0; q < 9; ++q) S->mg_ic[q] = K[q % 3] * M[q / 3] + M[q % 3] * K[q / 3];
  }
  MGCHK(hipMalloc((void**)&S->mg_m0, (MG_NULL_MAX * n0 + MG_NULL_MAX) * sizeof(double)));
  MGCHK(hipMemset(S->mg_m0, 0, (MG_NULL_MAX * n0 + MG_NULL_MAX) * sizeof(double)));
  MGCHK(hipMalloc((void**)&S->mg_dot, ((1 + MG_NULL_MAX) * MG_DOT_BLOCKS + 2) * sizeof(double)));      // per-workgroup partials of x_0 . rhs_0 and of the gauge sums, then the tickets of k_mg_up and k_mg_down
  MGCHK(hipMemset(S->mg_dot, 0, ((1 + MG_NULL_MAX) * MG_DOT_BLOCKS + 2) * sizeof(double)));
  S->hbm_bytes += (9 * S->mg_cells + 3 * S->mg_cells + (size_t)MG_NULL_MAX * S->mg_cells) * sizeof(double) + 9 * n0 * 8 + (S->chunk_cap + 64) * MG_PART * sizeof(double);
  return EULER_OK;
}
#undef MGCHK
void eu_mg_release(euler_sim* S) {
  eu_mg_split_release(S);
  for (double* d : {S->mg_a, S->mg_rhs, S->mg_part, S->mg_dot, S->mg_xbuf, S->mg_null0, S->mg_m0}) if (d) (void)hipFree(d);
  if (S->mg_a0i) (void)hipFree(S->mg_a0i);
  if (S->mg_inner0) (void)hipFree(S->mg_inner0);
  S->mg_inner0 = nullptr;
  S->mg_wd = nullptr;
  S->mg_a = S->mg_rhs = S->mg_x = S->mg_part = S->mg_dot = S->mg_xbuf = S->mg_null0 = S->mg_m0 = nullptr;
  S->mg_a0i = nullptr; S->mg_xslot = 0;
}

// ------------------------------------------------------------------------------------------ per solve: the operators
// A_0 = P_0^T A P_0 from the tiles' masks.  With c' = a_diag - (fluid neighbours) (the air neighbours of a cell) and p_i = P_0^T e_i (a cell's four weights),
//     A = sum_i c'_i e_i e_i^T + sum_edges (e_i - e_j)(e_i - e_j)^T      =>      A_0 = sum_i c'_i p_i p_i^T + sum_edges (p_i - p_j)(p_i - p_j)^T
// and since the weights are linear between two nodes, p_i - p_j of a horizontal edge is (1 / MG_G0)(e_J - e_(J+1)) in x times the row weights wy - the same for every
// edge of that row between the nodes J and J + 1 (0 beyond the outermost nodes) - and likewise for vertical edges.  So a lane (one row, 16 consecutive columns: at
// most MG_NSEG = three node intervals) only counts: per interval, its horizontal edges H, sum c' wx wx^T and sum over its vertical edges of wx wx^T (three integers each, in
// 1 / MG_G0^2), and adds row weights x those to the stencil entries - 64 integer adds per lane into a window in LDS, flushed by 64-bit atomics (units of 1 / MG_G0^4 = 2^-12: exact).
#define MG_WIN_R (MG_RPB + 2)                 // node rows a band's 64 rows touch
#define MG_WIN_C (80 / MG_G0 + 2)             // node columns a tile's 79 columns touch (8: 12, 16: 7)
#define MG_WIN (MG_WIN_R * MG_WIN_C * 9)
// one tile's sums into the wave's window.  SYNTH: every cell deep inside the water (CM_INTERIOR) - no mask is read
template <bool SYNTH>
__device__ __forceinline__ void mg_tile_window(int* win, const uint8_t* __restrict__ mask, const SkewGeom& g, int lane, int band, int k, int nx0, int ny0) {
  const size_t base = ((size_t)band * g.TS + (size_t)k * 16) * 64 + 2 * lane;
  // the lane's row against the node rows (weights in 1 / G0)
  const int uy = 64 * band + lane - MG_G0 / 2, I0 = uy >> MG_LOG;
  int wy0 = MG_G0 - (uy & (MG_G0 - 1)), wy1 = uy & (MG_G0 - 1);
  if (I0 < 0) { wy0 = 0; wy1 = MG_G0; }
  if (I0 >= ny0 - 1) { wy0 = MG_G0; wy1 = 0; }
  const bool vert_ok = I0 >= 0 && I0 <= ny0 - 2;
  const int Jb = (16 * k - lane - MG_G0 / 2) >> MG_LOG;      // node interval of the lane's first column
  int H[MG_NSEG], C[MG_NSEG][3], V[MG_NSEG][3];
#pragma unroll
  for (int m = 0; m < MG_NSEG; ++m) { H[m] = 0; C[m][0] = C[m][1] = C[m][2] = 0; V[m][0] = V[m][1] = V[m][2] = 0; }
#pragma unroll
  for (int P = 0; P < 8; ++P) {
    const unsigned int mm = SYNTH ? (unsigned int)(CM_INTERIOR | (CM_INTERIOR << 8)) : (unsigned int)*reinterpret_cast<const unsigned short*>(mask + base + P * 128);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned int cm = (mm >> (8 * h)) & 0xff;
      if (!(cm & CM_FLUID)) continue;
      const int ux = 16 * k + 2 * P + h - lane - MG_G0 / 2, J0 = ux >> MG_LOG;
      int w0 = MG_G0 - (ux & (MG_G0 - 1)), w1 = ux & (MG_G0 - 1);
      if (J0 < 0) { w0 = 0; w1 = MG_G0; }
      if (J0 >= nx0 - 1) { w0 = MG_G0; w1 = 0; }
      const int cp = (int)(cm >> CM_DIAG_SHIFT) - __popc(cm & (CM_RIGHT | CM_UP | CM_LEFT | CM_DOWN));
      const int hh = ((cm & CM_RIGHT) && J0 >= 0 && J0 <= nx0 - 2) ? 1 : 0;
      const int vv = (cm & CM_UP) ? 1 : 0;
      const int q00 = w0 * w0, q01 = w0 * w1, q11 = w1 * w1;
#pragma unroll
      for (int m = 0; m < MG_NSEG; ++m)
        if (J0 - Jb == m) { H[m] += hh; C[m][0] += cp * q00; C[m][1] += cp * q01; C[m][2] += cp * q11; V[m][0] += vv * q00; V[m][1] += vv * q01; V[m][2] += vv * q11; }
    }
  }
  const int rI0 = I0 - (MG_RPB * band - 1);
  const int cbase = (16 * k - 63 - MG_G0 / 2) >> MG_LOG;      // the window's first node column
#pragma unroll
  for (int sg = 0; sg < MG_NSEG; ++sg) {
    const int cJ0 = Jb + sg - cbase;
    const int Mx[2][2] = {{C[sg][0] + H[sg], C[sg][1] - H[sg]}, {C[sg][1] - H[sg], C[sg][2] + H[sg]}};
    const int Vx[2][2] = {{V[sg][0], V[sg][1]}, {V[sg][1], V[sg][2]}};
    const int wy[2] = {wy0, wy1};
#pragma unroll
    for (int ra = 0; ra < 2; ++ra)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            int val = wy[ra] * wy[rb] * Mx[ca][cb];
            if (vert_ok) val += (ra == rb ? 1 : -1) * Vx[ca][cb];
            if (val != 0) atomicAdd(&win[((rI0 + ra) * MG_WIN_C + cJ0 + ca) * 9 + (rb - ra + 1) * 3 + (cb - ca + 1)], val);
          }
  }
}
__global__ __launch_bounds__(256) void k_mg_assemble0(const uint8_t* __restrict__ mask, SkewGeom g, const unsigned int* __restrict__ list, const PcgScalars* sc,
                                                      int band_lo, int nx0, int ny0, unsigned long long* __restrict__ a0i) {
  __shared__ int s_win[4][MG_WIN];
  __shared__ int s_deep[MG_WIN];      // the window of a tile deep inside the water, away from the outermost nodes: the same integers whatever the tile (round 6)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* win = s_win[wv];
  for (int e = lane; e < MG_WIN; e += 64) win[e] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int ntb = g.T / 16, todo = (int)sc->n_chunks;
  const int n_waves = gridDim.x * 4;
  const size_t n0 = (size_t)nx0 * ny0;
  // A tile's rows sit at 64 band + lane, its columns at 16 k + ... - lane: against nodes 8 cells apart every tile meets them at the same offsets, so an INTERIOR tile (the
  // chunk list's top bit: every cell CM_INTERIOR) whose cells lie between the outermost nodes adds the same 1080 integers as any other - formed ONCE per workgroup here, by
  // the very code below on a mask of CM_INTERIOR (band 2, tile 8 of a level that is wide enough for them), and flushed without reading a mask or touching the window
  // (most of a tank: 16384^2 dam break 563 -> 475 us per solve - what is left are the ~700 memory-side adds of a tile's window).
  if (wv == 0) {
    mg_tile_window<true>(win, mask, g, lane, 2, 8, 1 << 20, 1 << 20);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < MG_WIN; e += 64) { s_deep[e] = win[e]; win[e] = 0; }
  }
  __syncthreads();
  for (int w = blockIdx.x * 4 + wv; w < todo; w += n_waves) {
    const unsigned int entry = list[w];
    const int tile = (int)(entry & ~EU_CHUNK_INTERIOR);
    const int band = band_lo + tile / ntb, k = tile % ntb;
    // no row of the tile below node row 0 or above the last but one, no column left of node column 0 or right of the last but one (the clamps of mg_tile_window never act)
    const bool deep = (entry & EU_CHUNK_INTERIOR) != 0 && band >= 1 && MG_RPB * band + MG_RPB - 1 <= ny0 - 2 && 16 * k - 63 - MG_G0 / 2 >= 0 && ((16 * k + 15 - MG_G0 / 2) >> MG_LOG) <= nx0 - 2;
    if (!deep) {
      mg_tile_window<false>(win, mask, g, lane, band, k, nx0, ny0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const int cbase = (16 * k - 63 - MG_G0 / 2) >> MG_LOG;      // the window's first node column
    for (int e = lane; e < MG_WIN; e += 64) {
      const int v = deep ? s_deep[e] : win[e];
      if (v == 0) continue;
      if (!deep) win[e] = 0;
      const int kk = e % 9, cJ = (e / 9) % MG_WIN_C, rI = e / (9 * MG_WIN_C);
      const int I = MG_RPB * band - 1 + rI, J = cbase + cJ;
      if (I >= 0 && I < ny0 && J >= 0 && J < nx0) atomicAdd(&a0i[(size_t)kk * n0 + (size_t)I * nx0 + J], (unsigned long long)(long long)v);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
// (the nine planes' entries [first, first + count) each: a rank of a split cycle converts the rows its tiles reach)
__global__ __launch_bounds__(256) void k_mg_convert0(const unsigned long long* __restrict__ a0i, double* __restrict__ a, size_t n0, size_t first, size_t count) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * count) return;
  const size_t e = (i / count) * n0 + first + i % count;
  a[e] = (double)(long long)a0i[e] * (1.0 / ((double)MG_G0 * MG_G0 * MG_G0 * MG_G0));
}

// weight of node j of a finer level (fn nodes) on node Jc of the next one (cn nodes), whose node Jc sits on the finer level's node 2 Jc
__device__ __forceinline__ double mg_w1(int Jc, int j, int fn, int cn) {
  if (j < 0 || j >= fn || Jc < 0 || Jc >= cn) return 0.0;
  if (j == 2 * Jc) return 1.0;
  if (j == 2 * Jc - 1) return 0.5;
  if (j == 2 * Jc + 1) return Jc + 1 <= cn - 1 ? 0.5 : 1.0;      // (the last fine node, odd, without a node to its right: constant)
  return 0.0;
}
// A_(l+1) = P^T A_l P, per node of level l + 1 and stencil entry: entry q of node (I, J) = sum over the fine nodes m under (I, J) and the fine nodes n under the
// coarse neighbour (I, J) + q of w(m) A_l[m, n] w(n) - at most 3 x 3 x 3 x 3 terms, walked in a fixed order (rows [r0, r1) of the coarse level).
// A workgroup takes MGC_TI x MGC_TJ coarse nodes: the nine entries of the (2 TI + 1) x (2 TJ + 1) fine nodes under them come into LDS row by row (coalesced; the one-thread-per-
// entry form of rounds 4-5 read every fine entry up to nine times at a stride of two nodes: 654 us per solve at 16384^2, six times its bytes' worth), the sums are formed
// from there - the same terms in the same order, a thread per (node, entry) - and leave through LDS as rows of one entry.
#define MGC_TI 8
#define MGC_TJ 16
#define MGC_PH (2 * MGC_TI + 1)
#define MGC_PW (2 * MGC_TJ + 1)
__global__ __launch_bounds__(256) void k_mg_coarsen(const double* __restrict__ af, int fnx, int fny, double* __restrict__ ac, int cnx, int cny, const PcgScalars* sc, int r0, int r1) {
  if (!sc->nonzero) return;
  __shared__ double sa[9][MGC_PH * MGC_PW];      // 40 KB
  __shared__ double so[9][MGC_TI * MGC_TJ];      // 9 KB: [entry][node of the tile]
  const int tilesx = (cnx + MGC_TJ - 1) / MGC_TJ;
  const int I0 = r0 + (int)(blockIdx.x / tilesx) * MGC_TI, J0 = (int)(blockIdx.x % tilesx) * MGC_TJ;
  const int fi0 = 2 * I0 - 1, fj0 = 2 * J0 - 1;
  const size_t fn = (size_t)fnx * fny, cn = (size_t)cnx * cny;
  for (int k = threadIdx.x; k < 9 * MGC_PH * MGC_PW; k += 256) {
    const int e = k / (MGC_PH * MGC_PW), c = k % (MGC_PH * MGC_PW), mi = fi0 + c / MGC_PW, mj = fj0 + c % MGC_PW;
    sa[e][c] = (mi >= 0 && mi < fny && mj >= 0 && mj < fnx) ? af[(size_t)e * fn + (size_t)mi * fnx + mj] : 0.0;      // (beyond the level: +0, and its weight is 0)
  }
  __syncthreads();
#pragma unroll 1
  for (int k = threadIdx.x; k < 9 * MGC_TI * MGC_TJ; k += 256) {
    const int node = k / 9, q = k % 9;
    const int I = I0 + node / MGC_TJ, J = J0 + node % MGC_TJ, I2 = I + q / 3 - 1, J2 = J + q % 3 - 1;
    double acc = 0.0;
    if (I < r1 && J < cnx && I2 >= 0 && I2 < cny && J2 >= 0 && J2 < cnx) {
      // The weights once per thread - of the fine rows 2 I + dy on node I (wmy) and, for dy + ey = -2 .. 2, on the neighbour I2 (wny); the same along x; wn = wny wnx - instead
      // of one mg_w1 per term, and EVERY term added: the oracle's mg_build skips the terms whose weight is 0, which are +-0 here (the entries are finite) and leave a sum that
      // started from +0 as it is - the same bits without 81 branches around 81 LDS reads (the pass was bound by exactly that: 654 -> 290 us per solve at 16384^2).
      double wmy[3], wmx[3], wny[5], wnx[5];
#pragma unroll
      for (int d = 0; d < 3; ++d) { wmy[d] = mg_w1(I, 2 * I + d - 1, fny, cny); wmx[d] = mg_w1(J, 2 * J + d - 1, fnx, cnx); }
#pragma unroll
      for (int d = 0; d < 5; ++d) { wny[d] = mg_w1(I2, 2 * I + d - 2, fny, cny); wnx[d] = mg_w1(J2, 2 * J + d - 2, fnx, cnx); }
      const int m0 = (2 * I - fi0) * MGC_PW + (2 * J - fj0);
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const double wM = wmy[dy + 1] * wmx[dx + 1];
          const int m = m0 + dy * MGC_PW + dx;
          double a[9];
#pragma unroll
          for (int e = 0; e < 9; ++e) a[e] = sa[e][m];
#pragma unroll
          for (int ey = -1; ey <= 1; ++ey)
#pragma unroll
            for (int ex = -1; ex <= 1; ++ex) {
              const double wN = wny[dy + ey + 2] * wnx[dx + ex + 2];
              acc += wM * a[(ey + 1) * 3 + ex + 1] * wN;
            }
        }
    }
    so[q][node] = acc;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 9 * MGC_TI * MGC_TJ; k += 256) {      // rows of MGC_TJ consecutive nodes of one entry
    const int q = k / (MGC_TI * MGC_TJ), node = k % (MGC_TI * MGC_TJ), I = I0 + node / MGC_TJ, J = J0 + node % MGC_TJ;
    if (I < r1 && J < cnx) ac[(size_t)q * cn + (size_t)I * cnx + J] = so[q][node];
  }
}
// (launch: one workgroup per tile of the coarse rows [r0, r1))
static inline unsigned mg_coarsen_blocks(int cnx, int r0, int r1) { return (unsigned)(((cnx + MGC_TJ - 1) / MGC_TJ) * ((r1 - r0 + MGC_TI - 1) / MGC_TI)); }

// ------------------------------------------------------------------------------------------ per iteration: the cycle
struct MgRect { int i0, i1, j0, j1; };      // rows [i0, i1) x columns [j0, j1)
__device__ __forceinline__ int mg_rw(const MgRect& r) { return r.j1 - r.j0; }
__device__ __forceinline__ int mg_rn(const MgRect& r) { return (r.i1 - r.i0) * (r.j1 - r.j0); }
__device__ __forceinline__ MgRect mg_grow(const MgRect& r, int ny, int nx) {
  return MgRect{r.i0 > 0 ? r.i0 - 1 : 0, r.i1 < ny ? r.i1 + 1 : ny, r.j0 > 0 ? r.j0 - 1 : 0, r.j1 < nx ? r.j1 + 1 : nx};
}
// the nodes of level l whose residual feeds the nodes `c` of level l + 1
__device__ __forceinline__ MgRect mg_fine_of(const MgRect& c, int ny, int nx) {
  MgRect f = {2 * c.i0 - 1, 2 * c.i1, 2 * c.j0 - 1, 2 * c.j1};
  if (f.i0 < 0) f.i0 = 0;
  if (f.j0 < 0) f.j0 = 0;
  if (f.i1 > ny) f.i1 = ny;
  if (f.j1 > nx) f.j1 = nx;
  return f;
}
// the nodes of level l that the owner of the nodes `c` of level l + 1 writes (a partition of level l)
__device__ __forceinline__ MgRect mg_owned_of(const MgRect& c, int cny, int cnx, int ny, int nx) {
  return MgRect{2 * c.i0, c.i1 == cny ? ny : 2 * c.i1, 2 * c.j0, c.j1 == cnx ? nx : 2 * c.j1};
}
__device__ __forceinline__ bool mg_in(const MgRect& r, int i, int j) { return i >= r.i0 && i < r.i1 && j >= r.j0 && j < r.j1; }

// Level 0's right-hand side from the tiles' sums (k_precond_tile: [band][group][tile][row slot][column slot], k_mg.h).  Node row I collects row slot 0 of the groups with
// I0 = I and row slot 1 of those with I0 = I - 1; I0 = 8 b + G - 1, so one group per node row - two HALF groups (G = 8 of band b - 1, G = 0 of band b) where the row lies
// across a band boundary; node column J lies in the slots of exactly two tiles of a group, k = (J + G - 1) >> 1 (slot 2 or 3) and k + 1 (slot 0 or 1).  Four loads per node
// (eight across a band boundary), every one issued before the first add; a fixed order.
__device__ __forceinline__ double mg_gather0(const double* __restrict__ part, int I, int J, int ntb, int band_lo, int band_hi) {
  double v[8];
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
    const int I0 = I - rs;
    const int b1 = (I0 + 1) >> MG_LOG, G1 = I0 + 1 - MG_RPB * b1;      // (I0 >= -1)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = h == 0 ? b1 : b1 - 1, G = h == 0 ? G1 : 8;
      const bool grp = (h == 0 || G1 == 0) && b >= band_lo && b < band_hi;
      const int ka = (J + G - 1) >> 1;
      const double* row = part + (((size_t)(grp ? b - band_lo : 0) * MG_NGRP + (size_t)G) * ntb) * (2 * MG_NSLOT) + rs * MG_NSLOT;      // [band][group][tile][row slot][column slot]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = ka + u, q = J - (2 * k - G - 1);      // u = 0: slot 2 or 3; u = 1: slot 0 or 1
        const bool ok = grp && k >= 0 && k < ntb;
        v[rs * 4 + h * 2 + u] = ok ? row[(size_t)k * (2 * MG_NSLOT) + q] : 0.0;
      }
    }
  }
  double t = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u) t = t + v[u];
  return t;
}

struct MgDownArgs {
  MgHier H;
  int lA, lB;                 // from the right-hand side of level lA to that of level lB (lA <= lB)
  int tile;                   // edge of a workgroup's tile of level lB
  int cap0, cap1;             // doubles per LDS patch of the levels lA, lA + 2 / lA + 1, lA + 3
  int tail;                   // the workgroup that draws the last ticket goes on with the levels >= lB (all of <= 1024 nodes), the dense top and the way back up to lB
  const double* part;         // GATHER: the tiles' partial sums
  int ntb, band_lo, band_hi;
  const double* inv;          // the dense top level's inverse [n_top][n_top]
  unsigned int* ticket;
  const PcgScalars* sc;
  // row slabs, split cycle (eu_mg_split_*): the launch covers the tile rows [tile_row0, tile_row0 + gridDim.x / tiles_x) of the output level only, and the entry level's
  // right-hand side counts as 0 outside the rows [clip_lo, clip_hi) - this rank's share lives there, the memory beyond holds other iterations' leftovers
  int tile_row0, clip_lo, clip_hi;
  int force;                  // 0: a launch behind convergence (sc->done) or of a solve that never started (!sc->nonzero) returns at once - every workgroup alike, no ticket is drawn
};

#define MG_DOWN_THREADS 1024
#define MG_SPLIT 704          // threads [0, MG_SPLIT) serve a launch's second level (<= 25 x 25 nodes + a row of slack), the others its third (<= 11 x 11)
#define MG_TAIL_MAX 1024      // nodes of the first level the tail takes
#define MG_TAIL_LEVELS 4      // stencil levels of the tail at most (1024 -> 272 -> 72 -> top would be 3)
static_assert(MG_TAIL_MAX == 1024 && MG_TAIL_LEVELS == 4, "eu_mg_alloc checks these limits with literals");

__device__ __forceinline__ void mg_st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double mg_ld_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// (A_l v)[c] over a patch of v in LDS: entries k = 0 .. 8 in this order, neighbours outside the grid skipped
__device__ __forceinline__ double mg_apply_patch(const double* __restrict__ a, size_t n, size_t c, int i, int j, int ny, int nx, const double* v, const MgRect& R) {
  const int w = mg_rw(R);
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int i2 = i + k / 3 - 1, j2 = j + k % 3 - 1;
    if (i2 < 0 || i2 >= ny || j2 < 0 || j2 >= nx) continue;
    t = t + a[(size_t)k * n + c] * v[(i2 - R.i0) * w + (j2 - R.j0)];
  }
  return t;
}

// the tail: levels lB .. top by ONE workgroup of 1024 threads, a thread per node (tid < n_l); vectors in LDS, stencils in registers
__device__ void mg_tail(const MgDownArgs& A, double* lds) {
  const MgHier& H = A.H;
  const int tid = threadIdx.x, top = H.nl - 1, nt = top - A.lB;      // nt stencil levels, then the dense one
  const int ntop = H.nx[top] * H.ny[top];
  // LDS: per stencil level of the tail two vectors (x1 -> x, t / x2), then the dense level's right-hand side and result, then its inverse
  double* vec[MG_TAIL_LEVELS];
  double* p = lds;
#pragma unroll
  for (int q = 0; q < MG_TAIL_LEVELS; ++q) { vec[q] = p; if (q < nt) p += 2 * (size_t)H.nx[A.lB + q] * H.ny[A.lB + q]; }
  double* top_rhs = p;
  double* top_y = p + ntop;
  double* s_inv = p + 2 * ntop;
  // everything this workgroup needs from memory, in one batch: the entry level's right-hand side (published by all workgroups), the stencils, the inverse
  double rhs[MG_TAIL_LEVELS], a[MG_TAIL_LEVELS][9], wd[MG_TAIL_LEVELS];
#pragma unroll
  for (int q = 0; q < MG_TAIL_LEVELS; ++q) {
    rhs[q] = 0.0; wd[q] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) a[q][k] = 0.0;
    if (q < nt) {
      const int l = A.lB + q;
      const size_t n = (size_t)H.nx[l] * H.ny[l];
      if ((size_t)tid < n) {
        const double* st = mg_sten(H, l);
#pragma unroll
        for (int k = 0; k < 9; ++k) a[q][k] = st[(size_t)k * n + tid];
        wd[q] = H.wd[H.off[l] + tid];
      }
    }
  }
  {
    const size_t n = (size_t)H.nx[A.lB] * H.ny[A.lB];
    if ((size_t)tid < n) rhs[0] = mg_ld_agent(H.rhs + H.off[A.lB] + tid);
  }
  for (int e = tid; e < ntop * ntop; e += MG_DOWN_THREADS) s_inv[e] = A.inv[e];
  // ---- down
#pragma unroll
  for (int q = 0; q < MG_TAIL_LEVELS; ++q) {
    if (q >= nt) break;
    const int l = A.lB + q, nx = H.nx[l], ny = H.ny[l], n = nx * ny;
    const int cnx = H.nx[l + 1], cny = H.ny[l + 1], cn = cnx * cny;
    double* x1 = vec[q];
    double* tt = vec[q] + n;
    if (tid < n) x1[tid] = wd[q] * rhs[q];
    __syncthreads();
    if (tid < n) {
      const int i = tid / nx, j = tid % nx;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int i2 = i + k / 3 - 1, j2 = j + k % 3 - 1;
        if (i2 < 0 || i2 >= ny || j2 < 0 || j2 >= nx) continue;
        t = t + a[q][k] * x1[i2 * nx + j2];
      }
      tt[tid] = wd[q] != 0.0 ? rhs[q] - t : 0.0;
    }
    __syncthreads();
    if (tid < cn) {      // full weighting: rows outer, columns inner
      const int I = tid / cnx, J = tid % cnx;
      double t = 0.0;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const double w = mg_w1(I, 2 * I + dy, ny, cny) * mg_w1(J, 2 * J + dx, nx, cnx);
          if (w != 0.0) t = t + w * tt[(2 * I + dy) * nx + 2 * J + dx];
        }
      if (q + 1 < MG_TAIL_LEVELS && q + 1 < nt) rhs[q + 1] = t;
      if (q + 1 == nt) top_rhs[tid] = t;      // the dense level's right-hand side
    }
  }
  if (nt == 0 && tid < ntop) top_rhs[tid] = rhs[0];      // (the entry level IS the dense level)
  __syncthreads();
  // ---- the dense level: y = inv rhs, 16 threads per row
  {
    double* rt = top_rhs;
    double* yt = top_y;
    const int row = tid >> 4, part = tid & 15;
    double v = 0.0;
    if (row < ntop) for (int c = part; c < ntop; c += 16) v += s_inv[row * ntop + c] * rt[c];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (row < ntop && part == 0) yt[row] = v;
    __syncthreads();
    if (nt == 0 && tid < ntop) H.x[H.off[top] + tid] = yt[tid];
  }
  // ---- up
#pragma unroll
  for (int q = MG_TAIL_LEVELS - 1; q >= 0; --q) {
    if (q >= nt) continue;
    const int l = A.lB + q, nx = H.nx[l], ny = H.ny[l], n = nx * ny;
    const int cnx = H.nx[l + 1], cny = H.ny[l + 1];
    double* x1 = vec[q];            // becomes x
    double* x2 = vec[q] + n;
    const double* e = q + 1 == nt ? top_y : vec[q + 1 < MG_TAIL_LEVELS ? q + 1 : q];
    if (tid < n) {
      const int i = tid / nx, j = tid % nx;
      const int I = i >> 1, J = j >> 1;
      const int I1 = (i & 1) && I + 1 <= cny - 1 ? I + 1 : I, J1 = (j & 1) && J + 1 <= cnx - 1 ? J + 1 : J;
      const double fy = (i & 1) && I + 1 <= cny - 1 ? 0.5 : 0.0, fx = (j & 1) && J + 1 <= cnx - 1 ? 0.5 : 0.0;
      const double lo = (1.0 - fx) * e[I * cnx + J] + fx * e[I * cnx + J1];
      const double hi = (1.0 - fx) * e[I1 * cnx + J] + fx * e[I1 * cnx + J1];
      x2[tid] = wd[q] != 0.0 ? x1[tid] + ((1.0 - fy) * lo + fy * hi) : 0.0;
    }
    __syncthreads();
    double xv = 0.0;
    if (tid < n) {
      const int i = tid / nx, j = tid % nx;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int i2 = i + k / 3 - 1, j2 = j + k % 3 - 1;
        if (i2 < 0 || i2 >= ny || j2 < 0 || j2 >= nx) continue;
        t = t + a[q][k] * x2[i2 * nx + j2];
      }
      xv = wd[q] != 0.0 ? x2[tid] + wd[q] * (rhs[q] - t) : 0.0;
    }
    __syncthreads();
    if (tid < n) { x1[tid] = xv; if (q == 0) H.x[H.off[l] + tid] = xv; }
    __syncthreads();
  }
}

// One level of the way down inside a workgroup: the residual behind the Jacobi step (in place of the right-hand side patch), then full weighting into the next level's
// patch.  The stencils of the NS nodes a thread handles sit in registers (loaded when the kernel starts, together with everything else that does not depend on the
// level above: the phases wait for LDS only); the threads [toff, toff + ...) do the work (the small levels of a launch use disjoint thread ranges, so a thread holds the
// first level's slots and at most one more).
template <int NS>
__device__ __forceinline__ void mg_prefetch_sten(const MgHier& H, int l, const MgRect& T, int toff, double (&a)[NS][9]) {
  const int tid = (int)threadIdx.x - toff, nx = H.nx[l], tw = mg_rw(T), tn = mg_rn(T);
  const size_t n = (size_t)nx * H.ny[l];
  const double* st = mg_sten(H, l);
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int e = tid + u * MG_DOWN_THREADS;
    const bool on = tid >= 0 && e < tn;
    const size_t c = on ? (size_t)(T.i0 + e / tw) * nx + T.j0 + e % tw : 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) a[u][k] = on ? st[(size_t)k * n + c] : 0.0;
  }
}
template <int NS>
__device__ __forceinline__ void mg_residual_patch(const MgHier& H, int l, const MgRect& R, const MgRect& T, int toff, const double (&a)[NS][9], double* prhs, const double* px1) {
  const int tid = (int)threadIdx.x - toff, nx = H.nx[l], ny = H.ny[l], w = mg_rw(R), tw = mg_rw(T), tn = mg_rn(T);
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int e = tid + u * MG_DOWN_THREADS;
    if (tid < 0 || e >= tn) continue;
    const int i = T.i0 + e / tw, j = T.j0 + e % tw;
    const int pe = (i - R.i0) * w + (j - R.j0);
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i2 = i + k / 3 - 1, j2 = j + k % 3 - 1;
      if (i2 < 0 || i2 >= ny || j2 < 0 || j2 >= nx) continue;
      t = t + a[u][k] * px1[pe + (k / 3 - 1) * w + (k % 3 - 1)];
    }
    prhs[pe] = a[u][4] != 0.0 ? prhs[pe] - t : 0.0;
  }
}

template <bool GATHER>
__global__ __launch_bounds__(MG_DOWN_THREADS) void k_mg_down(MgDownArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_last;
  if (!A.force && (A.sc->done || !A.sc->nonzero)) return;      // (the host polls late: launches queued behind convergence do no work - main.c:757 has left the loop)
  const MgHier& H = A.H;
  const int tid = threadIdx.x;
  const int tiles_x = (H.nx[A.lB] + A.tile - 1) / A.tile;
  const int ti = A.tile_row0 + blockIdx.x / tiles_x, tj = blockIdx.x % tiles_x;
  MgRect own = {ti * A.tile, (ti + 1) * A.tile, tj * A.tile, (tj + 1) * A.tile};
  if (own.i1 > H.ny[A.lB]) own.i1 = H.ny[A.lB];
  if (own.j1 > H.nx[A.lB]) own.j1 = H.nx[A.lB];
  // the rectangles of level l: R = the right-hand side this workgroup needs, T = where it needs the residual, O = what it writes out
  auto rects = [&](int l, MgRect& R, MgRect& T, MgRect& O) {
    MgRect r = own, o = own, t = own;
    for (int k = A.lB - 1; k >= l; --k) {
      o = mg_owned_of(o, H.ny[k + 1], H.nx[k + 1], H.ny[k], H.nx[k]);
      t = mg_fine_of(r, H.ny[k], H.nx[k]);
      r = mg_grow(t, H.ny[k], H.nx[k]);
    }
    R = r; T = t; O = o;
  };
  const int nt = A.lB - A.lA;      // transitions of this launch: 0 .. 3
  double* buf[2] = {lds, lds + 2 * (size_t)A.cap0};
  int cap[2] = {A.cap0, A.cap1};
  // ---- everything that does not depend on another level, in one batch: the stencils of the nodes this thread handles (level lA: up to three per thread; lA + 1 and lA + 2:
  // one, on the threads below / from MG_SPLIT), omega / diagonal of the right-hand-side nodes, the entry level's right-hand side itself
  double a0[3][9], an[1][9], w0[3], wn = 0.0;      // (an: level lA + 1's stencil on the threads below MG_SPLIT, level lA + 2's on the others)
  MgRect R0, T0, O0;
  rects(A.lA, R0, T0, O0);
  if (nt >= 1) mg_prefetch_sten<3>(H, A.lA, T0, 0, a0);
#pragma unroll
  for (int k = 0; k < 9; ++k) an[0][k] = 0.0;
  if (nt >= 2 && tid < MG_SPLIT) { MgRect R, T, O; rects(A.lA + 1, R, T, O); mg_prefetch_sten<1>(H, A.lA + 1, T, 0, an); }
  if (nt >= 3 && tid >= MG_SPLIT) { MgRect R, T, O; rects(A.lA + 2, R, T, O); mg_prefetch_sten<1>(H, A.lA + 2, T, MG_SPLIT, an); }
  {      // omega / diagonal of the level this thread writes a right-hand side of in the restriction phases: lA + 1 on the threads below MG_SPLIT, lA + 2 from there
    if (nt >= 2 && tid < MG_SPLIT) { MgRect R, T, O; rects(A.lA + 1, R, T, O); if (tid < mg_rn(R)) wn = H.wd[H.off[A.lA + 1] + (size_t)(R.i0 + tid / mg_rw(R)) * H.nx[A.lA + 1] + R.j0 + tid % mg_rw(R)]; }
    if (nt >= 3 && tid >= MG_SPLIT) { MgRect R, T, O; rects(A.lA + 2, R, T, O); const int e = tid - MG_SPLIT; if (e < mg_rn(R)) wn = H.wd[H.off[A.lA + 2] + (size_t)(R.i0 + e / mg_rw(R)) * H.nx[A.lA + 2] + R.j0 + e % mg_rw(R)]; }
  }
  {
    const int l = A.lA, nx = H.nx[l], w = mg_rw(R0), rn = mg_rn(R0);
    double v0[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + u * MG_DOWN_THREADS;
      v0[u] = 0.0; w0[u] = 0.0;
      if (e >= rn) continue;
      const int i = R0.i0 + e / w, j = R0.j0 + e % w;
      const size_t c = (size_t)i * nx + j;
      w0[u] = H.wd[H.off[l] + c];
      v0[u] = GATHER ? mg_gather0(A.part, i, j, A.ntb, A.band_lo, A.band_hi) : ((i >= A.clip_lo && i < A.clip_hi) ? H.rhs[H.off[l] + c] : 0.0);
    }
    double* prhs = buf[0];
    double* px1 = buf[0] + cap[0];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + u * MG_DOWN_THREADS;
      if (e >= rn) continue;
      const int i = R0.i0 + e / w, j = R0.j0 + e % w;
      if (GATHER && mg_in(O0, i, j)) {
        const size_t c = (size_t)i * nx + j;
        if (A.tail && nt == 0) mg_st_agent(H.rhs + H.off[l] + c, v0[u]);
        else H.rhs[H.off[l] + c] = v0[u];
      }
      prhs[e] = v0[u];
      px1[e] = w0[u] * v0[u];
    }
  }
  __syncthreads();
  // ---- level transitions lA -> lA + 1 -> ... -> lB
  for (int l = A.lA; l < A.lB; ++l) {
    const int rel = l - A.lA, cur = rel & 1;
    MgRect R, T, O, Rc, Tc, Oc;
    rects(l, R, T, O);
    rects(l + 1, Rc, Tc, Oc);
    const int nx = H.nx[l], ny = H.ny[l];
    const int cnx = H.nx[l + 1], cny = H.ny[l + 1];
    double* prhs = buf[cur];
    double* px1 = buf[cur] + cap[cur];
    double* crhs = buf[cur ^ 1];
    double* cx1 = buf[cur ^ 1] + cap[cur ^ 1];
    const int w = mg_rw(R);
    // residual behind the Jacobi step, in place of the right-hand side
    if (rel == 0) mg_residual_patch<3>(H, l, R, T, 0, a0, prhs, px1);
    else if (rel == 1) mg_residual_patch<1>(H, l, R, T, 0, an, prhs, px1);
    else mg_residual_patch<1>(H, l, R, T, MG_SPLIT, an, prhs, px1);
    __syncthreads();
    // full weighting -> the next level's right-hand side (and its Jacobi step): the threads below MG_SPLIT for lA + 1 (and for a launch's last level), from MG_SPLIT for lA + 2
    const int toff = rel == 1 && l + 1 < A.lB ? MG_SPLIT : 0;
    const int cw = mg_rw(Rc), cn = mg_rn(Rc);
    for (int e = tid - toff; e >= 0 && e < cn; e += MG_DOWN_THREADS) {
      const int I = Rc.i0 + e / cw, J = Rc.j0 + e % cw;
      const size_t c = (size_t)I * cnx + J;
      double t = 0.0;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const double wgt = mg_w1(I, 2 * I + dy, ny, cny) * mg_w1(J, 2 * J + dx, nx, cnx);
          if (wgt != 0.0) t = t + wgt * prhs[(2 * I + dy - R.i0) * w + (2 * J + dx - R.j0)];
        }
      if (mg_in(Oc, I, J)) {
        if (A.tail && l + 1 == A.lB) mg_st_agent(H.rhs + H.off[l + 1] + c, t);
        else H.rhs[H.off[l + 1] + c] = t;
      }
      if (l + 1 < A.lB) { crhs[e] = t; cx1[e] = wn * t; }      // (one node per thread on these levels: wn is this node's omega / diagonal)
    }
    __syncthreads();
  }
  if (!A.tail) return;
  // ---- the last workgroup to get here takes the small levels
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned int t = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1;
    if (s_last) __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  mg_tail(A, lds);
}

// up: x_l = x2 + omega (rhs - A x2) / d with x2 = the Jacobi step + P x_(l+1), for the levels below the tail's entry level down to 0.  A workgroup owns a tile of
// level 0 and recomputes what it needs of the coarser levels: x_l on `X_l`, x2 on X_l grown by one, x_(l+1) on the nodes around that.
struct MgUpArgs {
  MgHier H;
  int lC;                     // the level whose result is in H.x already (the tail's entry level; 0: nothing to do but the dot product)
  int l0;                     // the finest level this launch computes (k_mg_up: > 0 when level 0 is left to k_mg_up0; then no dot product, no epilogue)
  int tile, cap;              // edge of a workgroup's tile of level l0; doubles per LDS patch
  int row_lo, row_hi;         // node rows of level 0 whose x . rhs this rank adds to dot(z, r) (row slabs: the own rows)
  PcgScalars* sc;
  int fin_op, force;
  double* dot_part;           // [MG_DOT_BLOCKS] partials of x_0 . rhs_0, then [MG_NULL_MAX][MG_DOT_BLOCKS] partials of m_0 . x_0 (the gauge, below)
  unsigned int* ticket;
  const double* nullv;        // k_coarse.hip's cc_null: [.. + MG_NULL_MAX * 256] = the number of fluid regions cut off from the air
  const double* n0;           // [MG_NULL_MAX][nstride]: their indicators on the levels (level 0 first)
  const double* m0;           // [MG_NULL_MAX][n_0]: P_0^T of their indicators on the cells; [MG_NULL_MAX * n_0 + q] = m_0 . n_0
  size_t nstride;
  int tile_row0;              // row slabs, split cycle: the launch covers the tile rows from here on (of level l0)
  double* slot;               // ... and k_mg_up0 leaves {x_0 . rhs_0 over the own rows, the gauge sums} HERE instead of applying them: they are summed over the ranks first (k_mg_split_fold)
};
#ifndef MG_UP_THREADS
#define MG_UP_THREADS 1024
#endif
__device__ __forceinline__ MgRect mg_coarse_around(const MgRect& f, int cny, int cnx) {
  MgRect c = {f.i0 >> 1, ((f.i1 - 1) >> 1) + 2, f.j0 >> 1, ((f.j1 - 1) >> 1) + 2};
  if (c.i1 > cny) c.i1 = cny;
  if (c.j1 > cnx) c.j1 = cnx;
  return c;
}
__global__ __launch_bounds__(MG_UP_THREADS) void k_mg_up(MgUpArgs A) {
  extern __shared__ double lds[];
  __shared__ double s_red[MG_UP_THREADS / 64];
  __shared__ int s_last;
  const MgHier& H = A.H;
  const int tid = threadIdx.x;
  if (!A.force && (A.sc->done || !A.sc->nonzero)) return;      // behind convergence the level arrays are dead and the scalar epilogue must not run: every workgroup leaves at once (round 6; rounds 3-5 did the whole pass and skipped the epilogue)
  // Water cut off from the air (rare): its pressure is determined up to a constant, PCG delivers the one with n . M p = 0 (M the preconditioner, n the region's indicator)
  // and the reference's clamp (main.c:773-779) makes that constant observable.  The tile-local factor alone gives, like the reference's own, very nearly "mean 0 over the
  // region"; the dense level's pseudo-inverse keeps that, the Jacobi steps of the levels in between do not - so the correction is made mean-free over the CELLS of every
  // such region: x_0 -= n_0 (m_0 . x_0) / (m_0 . n_0).  The workgroups leave partials of m_0 . x_0, the last one folds them and corrects x_0 (oracle: mg_gauge).
  const int n_null = (int)A.nullv[MG_NULL_MAX * 256];
  const size_t n0n = (size_t)H.nx[0] * H.ny[0];
  double gv[MG_NULL_MAX] = {0.0, 0.0, 0.0, 0.0};
  const int L0 = A.l0;
  const int tiles_x = (H.nx[L0] + A.tile - 1) / A.tile;
  const int ti = A.tile_row0 + blockIdx.x / tiles_x, tj = blockIdx.x % tiles_x;
  MgRect own = {ti * A.tile, (ti + 1) * A.tile, tj * A.tile, (tj + 1) * A.tile};
  if (own.i1 > H.ny[L0]) own.i1 = H.ny[L0];
  if (own.j1 > H.nx[L0]) own.j1 = H.nx[L0];
  auto rects = [&](int l, MgRect& X, MgRect& X2) {      // X_l: where x_l is needed; X2_l = X_l grown by one
    MgRect x = own, x2 = mg_grow(own, H.ny[L0], H.nx[L0]);
    for (int k = L0 + 1; k <= l; ++k) { x = mg_coarse_around(x2, H.ny[k], H.nx[k]); x2 = mg_grow(x, H.ny[k], H.nx[k]); }
    X = x; X2 = x2;
  };
  double* bx[2] = {lds, lds + A.cap};      // x of the level above / of this level, alternating
  double* b2 = lds + 2 * (size_t)A.cap;    // x2 of this level
  // ---- everything that does not depend on the level above, in one batch.  Level 0: two nodes of X2 and one of X per thread; the coarser levels are small - the thread
  // ranges [off_l, off_l + |X2_l|) are disjoint, so a thread serves level 0 and at most one more (`myl`)
  double w2a[2] = {0.0, 0.0}, r2a[2] = {0.0, 0.0}, a0[9], w0 = 0.0, r0 = 0.0;      // level 0
  double w2m = 0.0, r2m = 0.0, am[9], wm = 0.0, rm = 0.0;                              // level myl
  int myl = -1, mye = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) { a0[k] = 0.0; am[k] = 0.0; }
  if (A.lC > L0) {
    MgRect X, X2;
    rects(L0, X, X2);
    const int nx = H.nx[L0];
    const size_t n = (size_t)nx * H.ny[L0];
    const int w2 = mg_rw(X2), w = mg_rw(X);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * MG_UP_THREADS;
      if (e < mg_rn(X2)) { const size_t c = H.off[L0] + (size_t)(X2.i0 + e / w2) * nx + X2.j0 + e % w2; w2a[u] = H.wd[c]; r2a[u] = H.rhs[c]; }
    }
    if (tid < mg_rn(X)) {
      const size_t c = (size_t)(X.i0 + tid / w) * nx + X.j0 + tid % w;
      const double* st = mg_sten(H, L0);
#pragma unroll
      for (int k = 0; k < 9; ++k) a0[k] = st[(size_t)k * n + c];
      w0 = H.wd[H.off[L0] + c]; r0 = H.rhs[H.off[L0] + c];
    }
    int off = 0;
    for (int l = L0 + 1; l < A.lC; ++l) {
      rects(l, X, X2);
      const int e = tid - off, n2 = mg_rn(X2);
      if (e >= 0 && e < n2) {
        myl = l; mye = e;
        const int lnx = H.nx[l];
        const size_t ln = (size_t)lnx * H.ny[l];
        const int lw2 = mg_rw(X2), lw = mg_rw(X);
        const size_t c2 = (size_t)(X2.i0 + e / lw2) * lnx + X2.j0 + e % lw2;
        w2m = H.wd[H.off[l] + c2]; r2m = H.rhs[H.off[l] + c2];
        if (e < mg_rn(X)) {
          const size_t c = (size_t)(X.i0 + e / lw) * lnx + X.j0 + e % lw;
          const double* st = mg_sten(H, l);
#pragma unroll
          for (int k = 0; k < 9; ++k) am[k] = st[(size_t)k * ln + c];
          wm = H.wd[H.off[l] + c]; rm = H.rhs[H.off[l] + c];
        }
      }
      off += n2;
    }
    rects(A.lC, X, X2);
    const int cw = mg_rw(X), cnx = H.nx[A.lC];
    double* dst = bx[A.lC & 1];
    for (int e = tid; e < mg_rn(X); e += MG_UP_THREADS) dst[e] = H.x[H.off[A.lC] + (size_t)(X.i0 + e / cw) * cnx + X.j0 + e % cw];
    __syncthreads();
  }
  double dv = 0.0;
  for (int l = A.lC - 1; l >= L0; --l) {
    MgRect X, X2, Xc, X2c;
    rects(l, X, X2);
    rects(l + 1, Xc, X2c);
    const int nx = H.nx[l], ny = H.ny[l];
    const int cnx = H.nx[l + 1], cny = H.ny[l + 1];
    const double* e_ = bx[(l + 1) & 1];
    double* xo = bx[l & 1];
    const int w2 = mg_rw(X2), cw = mg_rw(Xc), w = mg_rw(X);
    // x2 = the Jacobi step + P x_(l+1) on X2
    auto x2_at = [&](int e, double wdv, double rv) __attribute__((always_inline)) {
      const int i = X2.i0 + e / w2, j = X2.j0 + e % w2;
      const int I = i >> 1, J = j >> 1;
      const bool oy = (i & 1) && I + 1 <= cny - 1, ox = (j & 1) && J + 1 <= cnx - 1;
      const int I1 = oy ? I + 1 : I, J1 = ox ? J + 1 : J;
      const double fy = oy ? 0.5 : 0.0, fx = ox ? 0.5 : 0.0;
      const double lo = (1.0 - fx) * e_[(I - Xc.i0) * cw + (J - Xc.j0)] + fx * e_[(I - Xc.i0) * cw + (J1 - Xc.j0)];
      const double hi = (1.0 - fx) * e_[(I1 - Xc.i0) * cw + (J - Xc.j0)] + fx * e_[(I1 - Xc.i0) * cw + (J1 - Xc.j0)];
      b2[e] = wdv != 0.0 ? wdv * rv + ((1.0 - fy) * lo + fy * hi) : 0.0;
    };
    if (l == L0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int e = tid + u * MG_UP_THREADS; if (e < mg_rn(X2)) x2_at(e, w2a[u], r2a[u]); }
    } else if (l == myl) x2_at(mye, w2m, r2m);
    __syncthreads();
    // x = x2 + (omega / d) (rhs - A x2) on X
    auto x_at = [&](int e, const double (&a)[9], double wdv, double rv) __attribute__((always_inline)) {
      const int i = X.i0 + e / w, j = X.j0 + e % w;
      const int pe = (i - X2.i0) * w2 + (j - X2.j0);
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int i2 = i + k / 3 - 1, j2 = j + k % 3 - 1;
        if (i2 < 0 || i2 >= ny || j2 < 0 || j2 >= nx) continue;
        t = t + a[k] * b2[pe + (k / 3 - 1) * w2 + (k % 3 - 1)];
      }
      const double xv = wdv != 0.0 ? b2[pe] + wdv * (rv - t) : 0.0;
      xo[e] = xv;
      if (l == L0 && L0 > 0) { if (mg_in(own, i, j)) H.x[H.off[L0] + (size_t)i * nx + j] = xv; }      // (level 0 follows in k_mg_up0)
      if (l == 0) {
        const size_t c = (size_t)i * nx + j;
        if (n_null > 0) {
          mg_st_agent(H.x + c, xv);
#pragma unroll
          for (int q = 0; q < MG_NULL_MAX; ++q) if (q < n_null) gv[q] += A.m0[(size_t)q * n0n + c] * xv;
        } else H.x[c] = xv;
        if (i >= A.row_lo && i < A.row_hi) dv += xv * rv;
      }
    };
    if (l == L0) { if (tid < mg_rn(X)) x_at(tid, a0, w0, r0); }
    else if (l == myl && mye < mg_rn(X)) x_at(mye, am, wm, rm);
    __syncthreads();
  }
  if (A.lC == 0) {      // level 0 is the tail's own level: only the dot product is left
    const int w = mg_rw(own), nx = H.nx[0];
    for (int e = tid; e < mg_rn(own); e += MG_UP_THREADS) {
      const int i = own.i0 + e / w, j = own.j0 + e % w;
      const size_t c = (size_t)i * nx + j;
      const double xv = n_null > 0 ? mg_ld_agent(H.x + c) : H.x[c];      // (written by the tail workgroup of the launch before)
      if (i >= A.row_lo && i < A.row_hi) dv += xv * H.rhs[c];
#pragma unroll
      for (int q = 0; q < MG_NULL_MAX; ++q) if (q < n_null) gv[q] += A.m0[(size_t)q * n0n + c] * xv;
    }
  }
  if (L0 > 0) return;      // (L0 > 0: level 0, the dot product and the epilogue are k_mg_up0's)
  for (int q = 0; q < n_null && q < MG_NULL_MAX; ++q) {
    double g = q == 0 ? gv[0] : q == 1 ? gv[1] : q == 2 ? gv[2] : gv[3];
    g = eu_wave_sum(g);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = g;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int k = 0; k < MG_UP_THREADS / 64; ++k) t += s_red[k];
      mg_st_agent(A.dot_part + (size_t)(1 + q) * MG_DOT_BLOCKS + blockIdx.x, t);
    }
  }
  dv = eu_wave_sum(dv);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = dv;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY thread's x_0 stores have left before the workgroup's ticket is drawn (as in k_mg_down): the gauge pass of the last workgroup reads them
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int k = 0; k < MG_UP_THREADS / 64; ++k) t += s_red[k];
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&A.dot_part[blockIdx.x]), (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  for (int q = 0; q < n_null && q < MG_NULL_MAX; ++q) {      // the gauge: every workgroup's x_0 is in memory (agent-scope stores, drained before the tickets)
    double g = 0.0;
    for (unsigned int k = tid; k < gridDim.x; k += MG_UP_THREADS) g += mg_ld_agent(A.dot_part + (size_t)(1 + q) * MG_DOT_BLOCKS + k);
    g = eu_wave_sum(g);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = g;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < MG_UP_THREADS / 64; ++k) tot += s_red[k];
    const double mn = A.m0[(size_t)MG_NULL_MAX * n0n + q];
    if (mn > 0.0) {
      const double cq = tot / mn;
      for (size_t c = tid; c < n0n; c += MG_UP_THREADS) {
        const double nv = A.n0[(size_t)q * A.nstride + c];
        if (nv != 0.0) mg_st_agent(H.x + c, mg_ld_agent(H.x + c) - nv * cq);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  double t = 0.0;
  for (unsigned int k = tid; k < gridDim.x; k += MG_UP_THREADS)
    t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&A.dot_part[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  t = eu_wave_sum(t);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) {
    double v = 0.0;
    for (int k = 0; k < MG_UP_THREADS / 64; ++k) v += s_red[k];
    if (A.fin_op == MG_FIN_SLOT) { A.sc->comm_val2 = v; if (A.sc->comm_slot) A.sc->comm_slot[1] += v; }      // (row slabs, split cycle: this rank's share travels with the pair)
    else {
      v = A.sc->sigma_new + v;      // k_precond_tile left dot(z_tile, r) there (FIN_STORE_ONLY)
      if (A.fin_op == MFIN_SIGMA_INIT) A.sc->sigma = v;                                                       // main.c:748
      else if (A.fin_op == MFIN_BETA) { A.sc->sigma_new = v; A.sc->beta = v / A.sc->sigma; A.sc->sigma = v; }   // main.c:762-765
      else A.sc->sigma_new = v;
    }
    __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------ the big level: level 0 <-> level 1 on small workgroups
// Level 0 holds most of the hierarchy's nodes (1024^2 of them on an 8192^2 grid).  The kernels above give a workgroup of 1024 threads one tile and hide nothing behind
// anything - right for the small levels, whose cost is the chain of phases.  Level 0 is the opposite case: many tiles, little work in each.  These two kernels take it with
// workgroups of 256 threads, several nodes per thread, loads issued where they are needed: five or six workgroups share a CU and one's phase waits behind another's loads
// (8192^2: 101 + 87 us for the two ends of the cycle as 1024-thread workgroups -> see DESIGN.md).
#ifndef MG_FINE_THREADS
#define MG_FINE_THREADS 256
#endif
#define MG_SMALL_LEVEL0 16384      // nodes of level 0 up to which the cycle runs as two launches of the general kernels (launch_mg_cycle)
// k_mg_down1: level 0's right-hand side from the tiles' partial sums (GATHER) or from H.rhs (row slabs: summed over the ranks before), its residual behind the Jacobi step,
// full weighting -> level 1's right-hand side.  A workgroup owns tile x tile nodes of level 1.
template <bool GATHER>
__global__ __launch_bounds__(MG_FINE_THREADS) void k_mg_down1(MgDownArgs A) {
  extern __shared__ double lds[];
  if (!A.force && (A.sc->done || !A.sc->nonzero)) return;      // (as k_mg_down: nothing behind convergence)
  const MgHier& H = A.H;
  const int tid = threadIdx.x;
  const int nx = H.nx[0], ny = H.ny[0], cnx = H.nx[1], cny = H.ny[1];
  const size_t n = (size_t)nx * ny;
  const int tiles_x = (cnx + A.tile - 1) / A.tile;
  const int ti = A.tile_row0 + blockIdx.x / tiles_x, tj = blockIdx.x % tiles_x;
  MgRect own = {ti * A.tile, (ti + 1) * A.tile, tj * A.tile, (tj + 1) * A.tile};
  if (own.i1 > cny) own.i1 = cny;
  if (own.j1 > cnx) own.j1 = cnx;
  const MgRect O = mg_owned_of(own, cny, cnx, ny, nx), T = mg_fine_of(own, ny, nx), R = mg_grow(T, ny, nx);
  double* prhs = lds;
  double* px1 = lds + A.cap0;
  const double* st = mg_sten(H, 0);
  const int w = mg_rw(R), tw = mg_rw(T);
  for (int e = tid; e < mg_rn(R); e += MG_FINE_THREADS) {
    const int i = R.i0 + e / w, j = R.j0 + e % w;
    const size_t c = (size_t)i * nx + j;
    const double v = GATHER ? mg_gather0(A.part, i, j, A.ntb, A.band_lo, A.band_hi) : ((i >= A.clip_lo && i < A.clip_hi) ? H.rhs[c] : 0.0);      // (a staged, band-by-band gather out of LDS was built and measured: 167 us against 90)
    if (GATHER && mg_in(O, i, j)) H.rhs[c] = v;
    prhs[e] = v;
    const uint8_t in0 = H.inner0[c];      // 1: deep water (constants), 2: no fluid under the node (every entry 0, omega / d = 0: nothing else is loaded), 0: its own nine entries
    px1[e] = (in0 == 1 ? H.icwd : in0 == 2 ? 0.0 : H.wd[c]) * v;
  }
  __syncthreads();
  for (int e = tid; e < mg_rn(T); e += MG_FINE_THREADS) {
    const int i = T.i0 + e / tw, j = T.j0 + e % tw;
    const size_t c = (size_t)i * nx + j;
    const int pe = (i - R.i0) * w + (j - R.j0);
    const uint8_t in0 = H.inner0[c];
    if (in0 == 1) {      // deep water: the stencil is a constant (an interior node has all eight neighbours inside the grid)
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) t = t + H.ic[k] * px1[pe + (k / 3 - 1) * w + (k % 3 - 1)];
      prhs[pe] = prhs[pe] - t;
    } else if (in0 == 2) prhs[pe] = 0.0;      // (d == 0 below, without the loads)
    else {
      const double d = st[(size_t)4 * n + c];
      const double t = mg_apply_patch(st, n, c, i, j, ny, nx, px1, R);
      prhs[pe] = d != 0.0 ? prhs[pe] - t : 0.0;
    }
  }
  __syncthreads();
  const int ow = mg_rw(own);
  for (int e = tid; e < mg_rn(own); e += MG_FINE_THREADS) {
    const int I = own.i0 + e / ow, J = own.j0 + e % ow;
    double t = 0.0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const double wgt = mg_w1(I, 2 * I + dy, ny, cny) * mg_w1(J, 2 * J + dx, nx, cnx);
        if (wgt != 0.0) t = t + wgt * prhs[(2 * I + dy - R.i0) * w + (2 * J + dx - R.j0)];
      }
    H.rhs[H.off[1] + (size_t)I * cnx + J] = t;
  }
}

// k_mg_up0: level 0's result from level 1's (H.x), x_0 . rhs_0 into dot(z, r), the scalar epilogue, the gauge of cut-off regions (k_mg_up's last part, for level 0).
// A workgroup owns tile x tile nodes of level 0.
__global__ __launch_bounds__(MG_FINE_THREADS) void k_mg_up0(MgUpArgs A) {
  extern __shared__ double lds[];
  __shared__ double s_red[MG_FINE_THREADS / 64];
  __shared__ int s_last;
  const MgHier& H = A.H;
  const int tid = threadIdx.x;
  if (!A.force && (A.sc->done || !A.sc->nonzero)) return;      // (as k_mg_up)
  const int n_null = (int)A.nullv[MG_NULL_MAX * 256];
  const int nx = H.nx[0], ny = H.ny[0], cnx = H.nx[1], cny = H.ny[1];
  const size_t n0n = (size_t)nx * ny;
  double gv[MG_NULL_MAX] = {0.0, 0.0, 0.0, 0.0};
  const int tiles_x = (nx + A.tile - 1) / A.tile;
  const int ti = A.tile_row0 + blockIdx.x / tiles_x, tj = blockIdx.x % tiles_x;
  MgRect X = {ti * A.tile, (ti + 1) * A.tile, tj * A.tile, (tj + 1) * A.tile};
  if (X.i1 > ny) X.i1 = ny;
  if (X.j1 > nx) X.j1 = nx;
  const MgRect X2 = mg_grow(X, ny, nx), Xc = mg_coarse_around(X2, cny, cnx);
  double* e_ = lds;                  // level 1's result on Xc
  double* b2 = lds + A.cap;          // x2 on X2
  const double* st = mg_sten(H, 0);
  const int w2 = mg_rw(X2), cw = mg_rw(Xc), w = mg_rw(X);
  for (int e = tid; e < mg_rn(Xc); e += MG_FINE_THREADS) e_[e] = H.x[H.off[1] + (size_t)(Xc.i0 + e / cw) * cnx + Xc.j0 + e % cw];
  __syncthreads();
  for (int e = tid; e < mg_rn(X2); e += MG_FINE_THREADS) {
    const int i = X2.i0 + e / w2, j = X2.j0 + e % w2;
    const size_t c = (size_t)i * nx + j;
    const uint8_t in0 = H.inner0[c];
    if (in0 == 2) { b2[e] = 0.0; continue; }      // (omega / d == 0 below, without the loads: the air above a tank, most of a dam break's grid)
    const double wdv = in0 == 1 ? H.icwd : H.wd[c], rv = H.rhs[c];
    const int I = i >> 1, J = j >> 1;
    const bool oy = (i & 1) && I + 1 <= cny - 1, ox = (j & 1) && J + 1 <= cnx - 1;
    const int I1 = oy ? I + 1 : I, J1 = ox ? J + 1 : J;
    const double fy = oy ? 0.5 : 0.0, fx = ox ? 0.5 : 0.0;
    const double lo = (1.0 - fx) * e_[(I - Xc.i0) * cw + (J - Xc.j0)] + fx * e_[(I - Xc.i0) * cw + (J1 - Xc.j0)];
    const double hi = (1.0 - fx) * e_[(I1 - Xc.i0) * cw + (J - Xc.j0)] + fx * e_[(I1 - Xc.i0) * cw + (J1 - Xc.j0)];
    b2[e] = wdv != 0.0 ? wdv * rv + ((1.0 - fy) * lo + fy * hi) : 0.0;
  }
  __syncthreads();
  double dv = 0.0;
  for (int e = tid; e < mg_rn(X); e += MG_FINE_THREADS) {
    const int i = X.i0 + e / w, j = X.j0 + e % w;
    const size_t c = (size_t)i * nx + j;
    const uint8_t in0 = H.inner0[c];
    const bool inner = in0 == 1, empty = in0 == 2;
    const double wdv = inner ? H.icwd : empty ? 0.0 : H.wd[c], rv = empty ? 0.0 : H.rhs[c];      // (an empty node's x_0 is 0 and adds 0 to x_0 . rhs_0)
    const int pe2 = (i - X2.i0) * w2 + (j - X2.j0);
    double t = 0.0;
    if (inner) {
#pragma unroll
      for (int k = 0; k < 9; ++k) t = t + H.ic[k] * b2[pe2 + (k / 3 - 1) * w2 + (k % 3 - 1)];
    } else if (!empty) t = mg_apply_patch(st, n0n, c, i, j, ny, nx, b2, X2);
    const double xv = wdv != 0.0 ? b2[pe2] + wdv * (rv - t) : 0.0;
    if (n_null > 0) {
      mg_st_agent(H.x + c, xv);
#pragma unroll
      for (int q = 0; q < MG_NULL_MAX; ++q) if (q < n_null && i >= A.row_lo && i < A.row_hi) gv[q] += A.m0[(size_t)q * n0n + c] * xv;
    } else H.x[c] = xv;
    if (i >= A.row_lo && i < A.row_hi) dv += xv * rv;
  }
  for (int q = 0; q < n_null && q < MG_NULL_MAX; ++q) {
    double g = q == 0 ? gv[0] : q == 1 ? gv[1] : q == 2 ? gv[2] : gv[3];
    g = eu_wave_sum(g);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = g;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int k = 0; k < MG_FINE_THREADS / 64; ++k) t += s_red[k];
      mg_st_agent(A.dot_part + (size_t)(1 + q) * MG_DOT_BLOCKS + blockIdx.x, t);
    }
  }
  dv = eu_wave_sum(dv);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = dv;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every thread's x_0 stores have left before the ticket (see k_mg_up)
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int k = 0; k < MG_FINE_THREADS / 64; ++k) t += s_red[k];
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&A.dot_part[blockIdx.x]), (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  if (A.slot && tid < MG_NULL_MAX) A.slot[1 + tid] = 0.0;
  __syncthreads();
  for (int q = 0; q < n_null && q < MG_NULL_MAX; ++q) {      // the gauge: every workgroup's x_0 is in memory (agent-scope stores, drained before the tickets)
    double g = 0.0;
    for (unsigned int k = tid; k < gridDim.x; k += MG_FINE_THREADS) g += mg_ld_agent(A.dot_part + (size_t)(1 + q) * MG_DOT_BLOCKS + k);
    g = eu_wave_sum(g);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = g;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < MG_FINE_THREADS / 64; ++k) tot += s_red[k];
    const double mn = A.m0[(size_t)MG_NULL_MAX * n0n + q];
    if (A.slot) { if (tid == 0) A.slot[1 + q] = tot; }
    else if (mn > 0.0) {
      const double cq = tot / mn;
      for (size_t c = tid; c < n0n; c += MG_FINE_THREADS) {
        const double nv = A.n0[(size_t)q * A.nstride + c];
        if (nv != 0.0) mg_st_agent(H.x + c, mg_ld_agent(H.x + c) - nv * cq);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  double t = 0.0;
  for (unsigned int k = tid; k < gridDim.x; k += MG_FINE_THREADS)
    t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&A.dot_part[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  t = eu_wave_sum(t);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) {
    double v = 0.0;
    for (int k = 0; k < MG_FINE_THREADS / 64; ++k) v += s_red[k];
    if (A.slot) { A.slot[0] = v; __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }      // (split cycle: the ranks' shares are summed first)
    v = A.sc->sigma_new + v;      // k_precond_tile left dot(z_tile, r) there (FIN_STORE_ONLY)
    if (A.fin_op == MFIN_SIGMA_INIT) A.sc->sigma = v;                                                       // main.c:748
    else if (A.fin_op == MFIN_BETA) { A.sc->sigma_new = v; A.sc->beta = v / A.sc->sigma; A.sc->sigma = v; }   // main.c:762-765
    else A.sc->sigma_new = v;
    __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------ host side
static inline int mg_entry_level(const euler_sim* S) {      // the first level of <= MG_TAIL_MAX nodes: where the one-workgroup tail takes over
  int l = 0;
  while (l < S->mg_levels - 1 && (size_t)S->mg_nx[l] * S->mg_ny[l] > MG_TAIL_MAX) ++l;
  return l;
}
static size_t mg_tail_lds(const euler_sim* S, int lB) {
  size_t d = 0;
  for (int l = lB; l < S->mg_levels; ++l) d += 2 * (size_t)S->mg_nx[l] * S->mg_ny[l];      // (the dense level: its right-hand side and result)
  const size_t ntop = (size_t)S->mg_nx[S->mg_levels - 1] * S->mg_ny[S->mg_levels - 1];
  return (d + ntop * ntop) * sizeof(double);
}
static int mg_set_lds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return EULER_OK;
}

// the cycle: H.rhs[level 0] -> H.x[level 0]; `gather`: level 0's right-hand side comes from the tiles' partial sums (else it is in H.rhs already).
// Level 0 of more than MG_TAIL_MAX nodes: k_mg_down1, k_mg_down<false> x 1 .. 2 (the last one with the tail), k_mg_up for the levels between, k_mg_up0.  A small level 0 (grids up
// to 256^2): k_mg_down<gather> with the tail, then k_mg_up for the dot product alone.
static inline unsigned int* mg_tick_down(const euler_sim* S) { return reinterpret_cast<unsigned int*>(S->mg_dot + (1 + MG_NULL_MAX) * MG_DOT_BLOCKS + 1); }
static inline unsigned int* mg_tick_up(const euler_sim* S) { return reinterpret_cast<unsigned int*>(S->mg_dot + (1 + MG_NULL_MAX) * MG_DOT_BLOCKS); }
#ifndef MG_MIN_BLOCKS
#define MG_MIN_BLOCKS 64
#endif
static MgDownArgs mg_down_args(const euler_sim* S, const MgHier& H, int lA, int lB, int lC) {
  MgDownArgs A;
  A.H = H; A.lA = lA; A.lB = lB;
  const int steps = lB - lA;
  A.tile = steps == 3 ? 4 : steps == 2 ? 8 : steps == 1 ? 16 : 32;
  // a small level (grids up to 1024^2): the launch is latency, not work - smaller tiles until MG_MIN_BLOCKS workgroups share it (1024^2: 16 workgroups -> 64)
  while (steps > 0 && A.tile > 4 && (size_t)((S->mg_nx[lB] + A.tile - 1) / A.tile) * ((S->mg_ny[lB] + A.tile - 1) / A.tile) < MG_MIN_BLOCKS) A.tile /= 2;
  int e = A.tile, c[4] = {0, 0, 0, 0};
  c[steps] = e * e;
  for (int k = steps - 1; k >= 0; --k) { e = 2 * e + 3; c[k] = e * e; }
  A.cap0 = c[0] > c[2] ? c[0] : c[2]; A.cap1 = c[1] > c[3] ? c[1] : c[3];
  A.tail = lB == lC;
  A.part = S->mg_part; A.ntb = S->geom.T / 16; A.band_lo = S->band_lo; A.band_hi = S->band_hi;
  A.inv = S->cc_inv; A.ticket = mg_tick_down(S); A.sc = S->sc;
  A.tile_row0 = 0; A.clip_lo = 0; A.clip_hi = 0x7fffffff; A.force = 0;
  return A;
}
// rows [r0, r1) of the output level: the tile rows that cover them (split cycle); r1 < 0: the whole level
static int mg_launch_down(euler_sim* S, MgDownArgs A, bool g, int r0 = 0, int r1 = -1) {
  size_t lds = 2 * ((size_t)A.cap0 + A.cap1) * sizeof(double);
  if (A.tail) { const size_t t = mg_tail_lds(S, A.lB); if (t > lds) lds = t; }
  int trows = (S->mg_ny[A.lB] + A.tile - 1) / A.tile;
  if (r1 >= 0) { A.tile_row0 = r0 / A.tile; trows = (r1 + A.tile - 1) / A.tile - A.tile_row0; if (trows <= 0) return EULER_OK; }
  const unsigned nblk = (unsigned)(((S->mg_nx[A.lB] + A.tile - 1) / A.tile) * trows);
  if (g) { int rc = mg_set_lds(reinterpret_cast<const void*>(&k_mg_down<true>), lds); if (rc) return rc; hipLaunchKernelGGL(k_mg_down<true>, dim3(nblk), dim3(MG_DOWN_THREADS), lds, S->stream, A); }
  else { int rc = mg_set_lds(reinterpret_cast<const void*>(&k_mg_down<false>), lds); if (rc) return rc; hipLaunchKernelGGL(k_mg_down<false>, dim3(nblk), dim3(MG_DOWN_THREADS), lds, S->stream, A); }
  return EULER_OK;
}
static MgUpArgs mg_up_args(const euler_sim* S, const MgHier& H, int lC, int fin_op, int force) {
  MgUpArgs U;
  U.H = H; U.lC = lC; U.l0 = 0; U.tile = 32; U.cap = 36 * 36;
  U.row_lo = 0; U.row_hi = S->mg_ny[0];
  U.sc = S->sc; U.fin_op = fin_op; U.force = force; U.dot_part = S->mg_dot; U.ticket = mg_tick_up(S);
  U.nullv = S->cc_null; U.n0 = S->mg_null0; U.m0 = S->mg_m0; U.nstride = S->mg_cells;
  U.tile_row0 = 0; U.slot = nullptr;
  return U;
}
static int launch_mg_cycle(euler_sim* S, int fin_op, int force, bool gather) {
  eu_prof_begin(S, KC_COARSE_CYCLE);      // ONE event pair around the whole cycle
  const MgHier H = mg_hier(S);
  const int lC = mg_entry_level(S);
  if (S->mg_levels - 1 - lC > MG_TAIL_LEVELS) { eu_set_error("multilevel preconditioner: %d levels behind the entry level (at most %d)", S->mg_levels - 1 - lC, MG_TAIL_LEVELS); return EULER_EINVAL; }
  auto down_args = [&](int lA, int lB) { MgDownArgs A = mg_down_args(S, H, lA, lB, lC); A.force = force; return A; };
  auto launch_down = [&](const MgDownArgs& A, bool g) -> int { return mg_launch_down(S, A, g); };
  const MgUpArgs U = mg_up_args(S, H, lC, fin_op, force);
  const unsigned nblk0 = (unsigned)(((S->mg_nx[0] + 31) / 32) * ((S->mg_ny[0] + 31) / 32));
  if (nblk0 > MG_DOT_BLOCKS) { eu_set_error("multilevel preconditioner: %u tiles of level 0 (at most %d)", nblk0, MG_DOT_BLOCKS); return EULER_EINVAL; }
  if ((size_t)S->mg_nx[0] * S->mg_ny[0] <= MG_SMALL_LEVEL0) {
    // a small level 0 (grids up to 1024^2): the launches are the cost, not the nodes - ONE launch down (gather, every transition, its last workgroup the small levels), ONE back up
    // with the dot product (1024^2: 6 launches per iteration -> 4)
    int lA = 0;
    bool first = true;
    do {
      const int lB = lA + 3 < lC ? lA + 3 : lC;
      int rc = launch_down(down_args(lA, lB), first && gather);
      if (rc) return rc;
      lA = lB;
      first = false;
    } while (lA < lC);
    MgUpArgs V = U;
    while (V.tile > 8 && (size_t)((S->mg_nx[0] + V.tile - 1) / V.tile) * ((S->mg_ny[0] + V.tile - 1) / V.tile) < MG_MIN_BLOCKS) V.tile /= 2;
    V.cap = (V.tile + 4) * (V.tile + 4);
    const unsigned nb = (unsigned)(((S->mg_nx[0] + V.tile - 1) / V.tile) * ((S->mg_ny[0] + V.tile - 1) / V.tile));
    hipLaunchKernelGGL(k_mg_up, dim3(nb), dim3(MG_UP_THREADS), 3 * (size_t)V.cap * sizeof(double), S->stream, V);
  } else {
    {      // level 0 -> 1 on small workgroups
      MgDownArgs A = down_args(0, 1);
      A.tail = 0;
      const size_t lds = 2 * (size_t)A.cap0 * sizeof(double);
      const unsigned nblk = (unsigned)(((S->mg_nx[1] + A.tile - 1) / A.tile) * ((S->mg_ny[1] + A.tile - 1) / A.tile));
      if (gather) hipLaunchKernelGGL(k_mg_down1<true>, dim3(nblk), dim3(MG_FINE_THREADS), lds, S->stream, A);
      else hipLaunchKernelGGL(k_mg_down1<false>, dim3(nblk), dim3(MG_FINE_THREADS), lds, S->stream, A);
    }
    int lA = 1;
    do {      // levels 1 .. lC, three transitions per launch; the one that reaches lC goes on with the tail (lC == 1: the tail alone)
      const int lB = lA + 3 < lC ? lA + 3 : lC;
      int rc = launch_down(down_args(lA, lB), false);
      if (rc) return rc;
      lA = lB;
    } while (lA < lC);
    if (lC > 1) {      // back up to level 1
      MgUpArgs V = U;
      V.l0 = 1;
      const unsigned nblk = (unsigned)(((S->mg_nx[1] + 31) / 32) * ((S->mg_ny[1] + 31) / 32));
      hipLaunchKernelGGL(k_mg_up, dim3(nblk), dim3(MG_UP_THREADS), 3 * (size_t)V.cap * sizeof(double), S->stream, V);
    }
    hipLaunchKernelGGL(k_mg_up0, dim3(nblk0), dim3(MG_FINE_THREADS), 2 * (size_t)U.cap * sizeof(double), S->stream, U);
  }
  eu_prof_end(S, KC_COARSE_CYCLE);
  return EULER_OK;
}


// ------------------------------------------------------------------------------------------ row slabs: the cycle split by rows (round 5)
// Replicated, the cycle costs every rank the whole of it and an all-gather of level 0's right-hand side (cells / 64 doubles: 32 MB at 16384^2) per iteration.  Split:
//   * the way DOWN is linear, so a rank takes what ITS tiles contribute (zero elsewhere) down to the gather level Lg (the first of <= MG_GATHER_MAX nodes) on the rows that
//     share can reach, and the ranks all-gather WINDOWS of that level (own rows + a few): k_mg_split_unpack adds them up in rank order - the same bits everywhere;
//   * from Lg to the dense level and back to Lg everything runs replicated as before (small);
//   * the way back UP below Lg needs the TRUE right-hand sides on a rank's rows and their halo: the neighbours' shares on those rows (ZONES) travel with the edge rows of z
//     in the same neighbour exchange, and every rank computes levels Lg - 1 .. 0 for its own rows (+ 1) only;
//   * the correction's share of dot(z, r), x_0 . rhs_0, then lives on the ranks' own rows: ONE more 5-double all-gather behind the cycle (beta cannot wait for the next
//     exchange point) - three exchange points per iteration instead of two, for a level 0 of 1 / ranks the work and an all-gather of kilobytes.
// Needs every zone to reach into the NEXT rank only (slabs of a few bands); else - and on small grids, where level 0 is the gather level - the replicated form runs.
#define MG_GATHER_MAX 16384
struct MgSplitPlan {
  int valid, Lg, ranks, rank;
  int I[MG_MAXLEV][64][2];        // [level][rank]: the node rows a rank's own tiles can contribute to (its share's support)
  int NR[MG_MAXLEV][2];           // this rank: the node rows whose TRUE right-hand side its way up needs
  int NX[MG_MAXLEV][2];           // ... and the rows of x_l it computes
  int zs[2][MG_MAXLEV][2];        // [side 0 lo / 1 hi][level]: rows of this rank's share the neighbour on that side needs
  int zr[2][MG_MAXLEV][2];        // rows of that neighbour's share this rank needs
  int zoff_s[2][MG_MAXLEV], zoff_r[2][MG_MAXLEV], zone_doubles;      // offsets inside a zone message, the (uniform) message length
  int win_rows, nsmall;           // window rows per rank at level Lg (max over ranks), doubles per all-gather slot: 2 + win_rows * nx_Lg
};
struct MgSplitSeg { int level, row0, row1, off; };      // behind the exchange: rows of a level's right-hand side that change (a neighbour's share arrives, or leftovers must go)
struct MgSplitDst {
  int nseg, total, X;
  MgSplitSeg seg[4 * MG_MAXLEV];
  int own[MG_MAXLEV][2], zr[2][MG_MAXLEV][2], zoff[2][MG_MAXLEV];
};
struct MgSplitMsg { int rows[2][MG_MAXLEV][2], off[2][MG_MAXLEV]; int Lg, X, zone; };
struct MgSplitWin { int n; int lo[64], hi[64]; };
struct MgSplitState {
  MgSplitPlan P;
  MgSplitDst D;
  MgSplitMsg M;
  MgSplitWin W;
  double* msg;      // [4][X + zone]: send lo / hi, receive lo / hi
  double* gc;       // [ranks][1 + MG_NULL_MAX]: {x_0 . rhs_0 over the own rows, gauge sums}
  // the operators, per solve (eu_mg_setup): every level below the gather level is formed where it is OWNED (level 0: the rank's node rows; a coarser node row belongs to the
  // owner of the fine row under it) and MG_SETUP_HALO rows travel to either neighbour, level by level; the gather level's owned rows are all-gathered
  int setup_ok;
  int O[MG_MAXLEV][64][2];
  double* sbuf;     // [4][9 (MG_SETUP_HALO + 1) nx_0]
  double* abuf;     // [ranks][aslot]
  int aslot;
};
#define MG_SETUP_HALO 6
static inline MgSplitState* mg_split_state(const euler_sim* S) { return static_cast<MgSplitState*>(S->mg_split); }

static inline void mg_clip(int* r, int n) { if (r[0] < 0) r[0] = 0; if (r[1] > n) r[1] = n; if (r[1] < r[0]) r[1] = r[0]; }
static int mg_gather_level(const euler_sim* S) {
  const long long forced = S->opt[EULER_OPT_MG_SPLIT_LEVEL];
  const int lC = mg_entry_level(S);
  if (forced < 0) return 0;
  if (forced > 0) return (int)(forced < lC ? forced : lC);
  if ((size_t)S->mg_nx[0] * S->mg_ny[0] <= MG_SMALL_LEVEL0) return 0;
  int l = 0;
  while (l < lC && (size_t)S->mg_nx[l] * S->mg_ny[l] > MG_GATHER_MAX) ++l;
  return l;
}
static const MgSplitPlan* mg_split_plan(euler_sim* S) {
  if (S->mg_split) return mg_split_state(S)->P.valid ? &mg_split_state(S)->P : nullptr;
  MgSplitState* st = new (std::nothrow) MgSplitState();
  if (!st) return nullptr;
  S->mg_split = st;
  MgSplitPlan& P = st->P;
  const int R = S->bulk.nranks, me = S->bulk.rank, Lg = mg_gather_level(S);
  if (!S->has_comm || !S->slab_on || S->p2p_on || R > 64 || R < 2 || Lg < 1 || !S->mg_a) return nullptr;
  P.Lg = Lg; P.ranks = R; P.rank = me;
  for (int r = 0; r < R; ++r) {      // supports of the ranks' shares, level by level
    int a = MG_RPB * S->part_lo[r] - 1, b = MG_RPB * S->part_hi[r] + 1;
    for (int l = 0; l <= Lg; ++l) {
      P.I[l][r][0] = a; P.I[l][r][1] = b;
      mg_clip(P.I[l][r], S->mg_ny[l]);
      a = P.I[l][r][0]; b = P.I[l][r][1];
      const int na = (a - 2 >= 0 ? (a - 2 + 1) / 2 : 0), nb = (b + 1) / 2 + 1;      // ceil((a - 2) / 2) .. floor((b + 1) / 2)
      a = na; b = nb;
    }
  }
  {      // what this rank's way up needs
    int x0 = MG_RPB * S->part_lo[me] - 1, x1 = MG_RPB * S->part_hi[me] + 1;
    for (int l = 0; l < Lg; ++l) {
      P.NX[l][0] = x0; P.NX[l][1] = x1; mg_clip(P.NX[l], S->mg_ny[l]);
      P.NR[l][0] = P.NX[l][0] - 1; P.NR[l][1] = P.NX[l][1] + 1; mg_clip(P.NR[l], S->mg_ny[l]);
      x0 = P.NR[l][0] >> 1; x1 = ((P.NR[l][1] - 1) >> 1) + 2;      // mg_coarse_around
    }
  }
  // zones: side 0 = the rank below (me - 1), side 1 = the rank above.  What I need of a neighbour's share, what it needs of mine (its need is computed like mine)
  int zone = 0;
  for (int side = 0; side < 2; ++side) {
    const int nb = side == 0 ? me - 1 : me + 1;
    int offs = 0, offr = 0;
    for (int l = 0; l < Lg; ++l) {
      P.zs[side][l][0] = P.zs[side][l][1] = P.zr[side][l][0] = P.zr[side][l][1] = 0;
      P.zoff_s[side][l] = offs; P.zoff_r[side][l] = offr;
      if (nb < 0 || nb >= R) continue;
      // the neighbour's need at level l
      int x0 = MG_RPB * S->part_lo[nb] - 1, x1 = MG_RPB * S->part_hi[nb] + 1, nr[2] = {0, 0};
      for (int k = 0; k <= l; ++k) {
        int nx_[2] = {x0, x1}; mg_clip(nx_, S->mg_ny[k]);
        nr[0] = nx_[0] - 1; nr[1] = nx_[1] + 1; mg_clip(nr, S->mg_ny[k]);
        x0 = nr[0] >> 1; x1 = ((nr[1] - 1) >> 1) + 2;
      }
      const int s0 = P.I[l][me][0] > nr[0] ? P.I[l][me][0] : nr[0], s1 = P.I[l][me][1] < nr[1] ? P.I[l][me][1] : nr[1];
      if (s1 > s0) { P.zs[side][l][0] = s0; P.zs[side][l][1] = s1; offs += (s1 - s0) * S->mg_nx[l]; }
      const int r0 = P.I[l][nb][0] > P.NR[l][0] ? P.I[l][nb][0] : P.NR[l][0], r1 = P.I[l][nb][1] < P.NR[l][1] ? P.I[l][nb][1] : P.NR[l][1];
      if (r1 > r0) { P.zr[side][l][0] = r0; P.zr[side][l][1] = r1; offr += (r1 - r0) * S->mg_nx[l]; }
      // a zone may reach into the next rank only: the rank beyond must not contribute to what I need
      const int far = side == 0 ? me - 2 : me + 2;
      if (far >= 0 && far < R) {
        const int f0 = P.I[l][far][0] > P.NR[l][0] ? P.I[l][far][0] : P.NR[l][0], f1 = P.I[l][far][1] < P.NR[l][1] ? P.I[l][far][1] : P.NR[l][1];
        if (f1 > f0) P.valid = -1;
      }
    }
    zone = offs > zone ? offs : zone; zone = offr > zone ? offr : zone;
  }
  // the setup by owners: every rank's owned rows of every level below the gather level at least a halo thick, and what a rank reads of a level within its owned rows + halo
  int setup_bad = 0, maxo = 0;
  for (int r = 0; r < R; ++r) {
    int a = MG_RPB * S->part_lo[r], b = MG_RPB * S->part_hi[r];
    for (int l = 0; l <= Lg; ++l) {
      st->O[l][r][0] = a; st->O[l][r][1] = b;
      if (l < Lg && b - a < MG_SETUP_HALO) setup_bad = 1;
      a = (a + 1) / 2; b = (b + 1) / 2;
    }
    maxo = st->O[Lg][r][1] - st->O[Lg][r][0] > maxo ? st->O[Lg][r][1] - st->O[Lg][r][0] : maxo;
  }
  for (int l = 0; l < Lg; ++l) {
    const int o0 = st->O[l][me][0], o1 = st->O[l][me][1], ny = S->mg_ny[l];
    int lo = P.I[l][me][0] - 1, hi = P.I[l][me][1] + 1;                                     // the way down: the residual one row beyond the share's support
    lo = P.NR[l][0] < lo ? P.NR[l][0] : lo; hi = P.NR[l][1] > hi ? P.NR[l][1] : hi;         // the way up
    const int c0 = 2 * st->O[l + 1][me][0] - 1, c1 = 2 * st->O[l + 1][me][1];               // the next level's owned rows are formed from these
    lo = c0 < lo ? c0 : lo; hi = c1 > hi ? c1 : hi;
    if (lo < 0) lo = 0;
    if (hi > ny) hi = ny;
    if (lo < o0 - MG_SETUP_HALO || hi > o1 + MG_SETUP_HALO) setup_bad = 1;
  }
  st->aslot = 9 * maxo * S->mg_nx[Lg];
  int win = 0;
  for (int r = 0; r < R; ++r) win = P.I[Lg][r][1] - P.I[Lg][r][0] > win ? P.I[Lg][r][1] - P.I[Lg][r][0] : win;
  P.win_rows = win; P.nsmall = 2 + win * S->mg_nx[Lg];
  // every rank must come to the same verdict and the same message length: the plan's inputs are the partition (the same everywhere), so one all-reduce of {failed, zone} settles both
  double v[3] = {P.valid < 0 ? 1.0 : 0.0, (double)zone, (double)setup_bad};
  double* dv = S->mg_dot;      // (scratch: the partials are rewritten by the next cycle)
  if (hipMemcpyAsync(dv, v, sizeof v, hipMemcpyHostToDevice, S->stream) != hipSuccess || S->bulk.allreduce(S->bulk.ctx, dv, 3, 1) != 0 ||
      hipMemcpyAsync(v, dv, sizeof v, hipMemcpyDeviceToHost, S->stream) != hipSuccess || hipStreamSynchronize(S->stream) != hipSuccess) { P.valid = 0; return nullptr; }
  if (v[0] != 0.0) { P.valid = 0; return nullptr; }
  P.zone_doubles = (int)v[1];
  st->setup_ok = v[2] == 0.0;
  // what the kernels either side of the exchange need of the plan
  const int X = S->X;
  MgSplitMsg& M = st->M;
  M.Lg = Lg; M.X = X; M.zone = P.zone_doubles;
  for (int side = 0; side < 2; ++side) for (int l = 0; l < MG_MAXLEV; ++l) { M.rows[side][l][0] = P.zs[side][l][0]; M.rows[side][l][1] = P.zs[side][l][1]; M.off[side][l] = P.zoff_s[side][l]; }
  MgSplitDst& D = st->D;
  int local_fail = 0;      // what can fail on ONE rank (the segment table, the allocations): settled by a second all-reduce, so that every rank runs the same cycle
  D.nseg = 0; D.total = 0; D.X = X;
  for (int l = 0; l < Lg; ++l) {
    D.own[l][0] = P.I[l][me][0]; D.own[l][1] = P.I[l][me][1];
    for (int side = 0; side < 2; ++side) { D.zr[side][l][0] = P.zr[side][l][0]; D.zr[side][l][1] = P.zr[side][l][1]; D.zoff[side][l] = P.zoff_r[side][l]; }
    int run0 = -1;
    for (int row = P.NR[l][0]; row <= P.NR[l][1]; ++row) {
      const bool in = row < P.NR[l][1];
      const bool touched = in && (!(row >= D.own[l][0] && row < D.own[l][1]) || (row >= D.zr[0][l][0] && row < D.zr[0][l][1]) || (row >= D.zr[1][l][0] && row < D.zr[1][l][1]));
      if (touched && run0 < 0) run0 = row;
      if (!touched && run0 >= 0) {
        if (D.nseg >= 4 * MG_MAXLEV) { local_fail = 1; run0 = -1; continue; }      // (agreed on below: no rank leaves the plan alone)
        D.seg[D.nseg++] = MgSplitSeg{l, run0, row, D.total};
        D.total += (row - run0) * S->mg_nx[l];
        run0 = -1;
      }
    }
  }
  MgSplitWin& W = st->W;
  W.n = R;
  for (int r = 0; r < R; ++r) { W.lo[r] = P.I[Lg][r][0]; W.hi[r] = P.I[Lg][r][1]; }
  const size_t mlen = (size_t)X + P.zone_doubles;
  if (hipMalloc((void**)&st->msg, 4 * mlen * sizeof(double)) != hipSuccess || hipMalloc((void**)&st->gc, (size_t)R * (1 + MG_NULL_MAX) * sizeof(double)) != hipSuccess ||
      hipMemsetAsync(st->msg, 0, 4 * mlen * sizeof(double), S->stream) != hipSuccess || hipMemsetAsync(st->gc, 0, (size_t)R * (1 + MG_NULL_MAX) * sizeof(double), S->stream) != hipSuccess ||
      (st->setup_ok && (hipMalloc((void**)&st->sbuf, 4 * (size_t)9 * (MG_SETUP_HALO + 1) * S->mg_nx[0] * sizeof(double)) != hipSuccess ||
                        hipMalloc((void**)&st->abuf, (size_t)R * st->aslot * sizeof(double)) != hipSuccess ||
                        hipMemsetAsync(st->sbuf, 0, 4 * (size_t)9 * (MG_SETUP_HALO + 1) * S->mg_nx[0] * sizeof(double), S->stream) != hipSuccess ||
                        hipMemsetAsync(st->abuf, 0, (size_t)R * st->aslot * sizeof(double), S->stream) != hipSuccess))) {
    local_fail = 1;
  }
  // A rank that failed here alone would run the replicated cycle (two exchange points, an all-reduce of A_0) against the others' split cycle (three, halo + all-gather):
  // mismatched collectives, a hang.  One more all-reduce (once per plan): every rank takes the split cycle or none does.
  double lf = (double)local_fail;
  if (hipMemcpyAsync(dv, &lf, sizeof lf, hipMemcpyHostToDevice, S->stream) != hipSuccess || S->bulk.allreduce(S->bulk.ctx, dv, 1, 1) != 0 ||
      hipMemcpyAsync(&lf, dv, sizeof lf, hipMemcpyDeviceToHost, S->stream) != hipSuccess || hipStreamSynchronize(S->stream) != hipSuccess) lf = 1.0;
  if (lf != 0.0) {
    if (local_fail) eu_set_error("the split cycle's segment table or message buffers could not be set up on this rank; every rank falls back to the replicated cycle");
    if (st->msg) { (void)hipFree(st->msg); st->msg = nullptr; }
    if (st->gc) { (void)hipFree(st->gc); st->gc = nullptr; }
    if (st->sbuf) { (void)hipFree(st->sbuf); st->sbuf = nullptr; }
    if (st->abuf) { (void)hipFree(st->abuf); st->abuf = nullptr; }
    P.valid = 0;
    return nullptr;
  }
  P.valid = 1;
  S->opt[EULER_OPT_MG_SPLIT_ACTIVE] = Lg;
  return &P;
}
bool eu_mg_split(euler_sim* S) { return eu_is_mg(S) && mg_split_plan(S) != nullptr; }
int eu_mg_split_nsmall(euler_sim* S) { return mg_split_state(S)->P.nsmall; }
int eu_mg_split_count(euler_sim* S) { return S->X + mg_split_state(S)->P.zone_doubles; }
double* eu_mg_split_msg(euler_sim* S, int k) { return mg_split_state(S)->msg + (size_t)k * ((size_t)S->X + mg_split_state(S)->P.zone_doubles); }
double* eu_mg_split_gc(euler_sim* S) { return mg_split_state(S)->gc; }
void eu_mg_split_release(euler_sim* S) {
  MgSplitState* st = mg_split_state(S);
  if (!st) return;
  if (st->msg) (void)hipFree(st->msg);
  if (st->gc) (void)hipFree(st->gc);
  if (st->sbuf) (void)hipFree(st->sbuf);
  if (st->abuf) (void)hipFree(st->abuf);
  delete st;
  S->mg_split = nullptr;
  S->opt[EULER_OPT_MG_SPLIT_ACTIVE] = 0;
}

// pack: this rank's share on the rows a neighbour needs (ZONES) behind the edge row of z that k_precond_tile wrote into the message; its window of the gather level into
// its slot of the all-gather
__global__ __launch_bounds__(256) void k_mg_split_pack(MgHier H, MgSplitMsg M, double* __restrict__ msg_lo, double* __restrict__ msg_hi, double* __restrict__ slot, int w0, int w1, int win_rows) {
  const int side = blockIdx.y;      // 0 / 1: the messages; 2: the window
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (side == 2) {
    const int nx = H.nx[M.Lg];
    if (i >= (size_t)win_rows * nx) return;
    const int row = w0 + (int)(i / nx);
    slot[2 + i] = row < w1 ? H.rhs[H.off[M.Lg] + (size_t)row * nx + i % nx] : 0.0;
    return;
  }
  double* msg = side == 0 ? msg_lo : msg_hi;
  const size_t e = i;
  if (e >= (size_t)M.zone) return;
  double v = 0.0;
  for (int l = 0; l < M.Lg; ++l) {
    const size_t n = (size_t)(M.rows[side][l][1] - M.rows[side][l][0]) * H.nx[l];
    if (e >= (size_t)M.off[side][l] && e < (size_t)M.off[side][l] + n) { v = H.rhs[H.off[l] + (size_t)M.rows[side][l][0] * H.nx[l] + (e - M.off[side][l])]; break; }
  }
  msg[M.X + e] = v;
}
// unpack: (y = 0) the rows of D's segments become TRUE right-hand sides: this rank's share where its tiles reach (elsewhere the memory holds leftovers), + the lower neighbour's,
// + the upper one's - in that order; (y = 1, 2) the neighbours' edge rows of z into the compact rows k_search_apply reads; (y = 3) the gather level, whole, from the ranks'
// windows in rank order
__global__ __launch_bounds__(256) void k_mg_split_unpack(MgHier H, MgSplitDst D, int Lg, const double* __restrict__ msg_lo, const double* __restrict__ msg_hi, double* __restrict__ zlo,
                                                         double* __restrict__ zhi, const double* __restrict__ xbuf, int slotlen, MgSplitWin W) {
  const int what = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (what == 3) {
    const int nx = H.nx[Lg];
    if (i >= (size_t)nx * H.ny[Lg]) return;
    const int row = (int)(i / nx);
    double t = 0.0;
    for (int r = 0; r < W.n; ++r)
      if (row >= W.lo[r] && row < W.hi[r]) t = t + xbuf[(size_t)r * slotlen + 2 + (size_t)(row - W.lo[r]) * nx + i % nx];
    H.rhs[H.off[Lg] + i] = t;
    return;
  }
  if (what == 1) { if (zlo && i < (size_t)D.X) zlo[i] = msg_lo[i]; return; }
  if (what == 2) { if (zhi && i < (size_t)D.X) zhi[i] = msg_hi[i]; return; }
  if (i >= (size_t)D.total) return;
  for (int k = 0; k < D.nseg; ++k) {
    const MgSplitSeg& s = D.seg[k];
    const int l = s.level, nx = H.nx[l];
    const size_t n = (size_t)(s.row1 - s.row0) * nx;
    if (i < (size_t)s.off || i >= (size_t)s.off + n) continue;
    const size_t e = i - s.off;
    const int row = s.row0 + (int)(e / nx), col = (int)(e % nx);
    double* dst = H.rhs + H.off[l] + (size_t)row * nx + col;
    double v = (row >= D.own[l][0] && row < D.own[l][1]) ? *dst : 0.0;
    if (row >= D.zr[0][l][0] && row < D.zr[0][l][1]) v = v + msg_lo[D.X + D.zoff[0][l] + (size_t)(row - D.zr[0][l][0]) * nx + col];
    if (row >= D.zr[1][l][0] && row < D.zr[1][l][1]) v = v + msg_hi[D.X + D.zoff[1][l] + (size_t)(row - D.zr[1][l][0]) * nx + col];
    *dst = v;
    return;
  }
}
// behind the cycle: the ranks' {x_0 . rhs_0 over their own rows, gauge sums} -> dot(z, r) with its epilogue; the gauge of cut-off regions on this rank's rows
__global__ __launch_bounds__(256) void k_mg_split_fold(PcgScalars* sc, const double* __restrict__ vals, int R, int fin_op, int force, MgHier H, const double* __restrict__ nullv, const double* __restrict__ n0,
                                                       size_t nstride, const double* __restrict__ m0, int row0, int row1) {
  const bool idle = !force && (sc->done || !sc->nonzero);
  if (idle) return;
  const int n_null = (int)nullv[MG_NULL_MAX * 256];
  const size_t n0n = (size_t)H.nx[0] * H.ny[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double v = 0.0;
    for (int r = 0; r < R; ++r) v += vals[(size_t)r * (1 + MG_NULL_MAX)];
    v = sc->sigma_new + v;
    if (fin_op == MFIN_SIGMA_INIT) sc->sigma = v;
    else if (fin_op == MFIN_BETA) { sc->sigma_new = v; sc->beta = v / sc->sigma; sc->sigma = v; }
    else sc->sigma_new = v;
  }
  for (int q = 0; q < n_null && q < MG_NULL_MAX; ++q) {
    double tot = 0.0;
    for (int r = 0; r < R; ++r) tot += vals[(size_t)r * (1 + MG_NULL_MAX) + 1 + q];
    const double mn = m0[(size_t)MG_NULL_MAX * n0n + q];
    if (!(mn > 0.0)) continue;
    const double cq = tot / mn;
    const size_t lo = (size_t)row0 * H.nx[0], hi = (size_t)row1 * H.nx[0];
    for (size_t c = lo + (size_t)blockIdx.x * 256 + threadIdx.x; c < hi; c += (size_t)gridDim.x * 256) {
      const double nv = n0[(size_t)q * nstride + c];
      if (nv != 0.0) H.x[c] = H.x[c] - nv * cq;
    }
  }
}


// the split cycle's launches.  pre: before the G1 exchange (k_precond_tile has left the tiles' partial sums and, in the messages, the edge rows of z)
int eu_mg_split_pre(euler_sim* S) {
  MgSplitState* st = mg_split_state(S);
  const MgSplitPlan& P = st->P;
  const MgHier H = mg_hier(S);
  const int me = P.rank, Lg = P.Lg;
  eu_prof_begin(S, KC_COARSE_CYCLE);
  {      // level 0 -> 1 from this rank's tiles
    MgDownArgs A = mg_down_args(S, H, 0, 1, -1);
    A.tile_row0 = P.I[1][me][0] / A.tile;
    const int trows = (P.I[1][me][1] + A.tile - 1) / A.tile - A.tile_row0;
    const unsigned nblk = (unsigned)(((S->mg_nx[1] + A.tile - 1) / A.tile) * trows);
    hipLaunchKernelGGL(k_mg_down1<true>, dim3(nblk), dim3(MG_FINE_THREADS), 2 * (size_t)A.cap0 * sizeof(double), S->stream, A);
  }
  for (int lA = 1; lA < Lg;) {
    const int lB = lA + 3 < Lg ? lA + 3 : Lg;
    MgDownArgs A = mg_down_args(S, H, lA, lB, -1);
    A.clip_lo = P.I[lA][me][0]; A.clip_hi = P.I[lA][me][1];
    int rc = mg_launch_down(S, A, false, P.I[lB][me][0], P.I[lB][me][1]);
    if (rc) return rc;
    lA = lB;
  }
  const size_t win = (size_t)P.win_rows * S->mg_nx[Lg], most = win > (size_t)P.zone_doubles ? win : (size_t)P.zone_doubles;
  const size_t mlen = (size_t)S->X + P.zone_doubles;
  hipLaunchKernelGGL(k_mg_split_pack, dim3((unsigned)((most + 255) / 256), 3), dim3(256), 0, S->stream, H, st->M, st->msg, st->msg + mlen, S->mg_xbuf + (size_t)me * P.nsmall,
                     P.I[Lg][me][0], P.I[Lg][me][1], P.win_rows);
  eu_prof_end(S, KC_COARSE_CYCLE);
  return EULER_OK;
}
// mid: behind the exchange - the true right-hand sides, the replicated middle, the way up on the own rows; leaves {x_0 . rhs_0, gauge sums} of the own rows in this rank's slot of gc
int eu_mg_split_mid(euler_sim* S, int fin_op, int force, double* zrecv_lo, double* zrecv_hi) {
  MgSplitState* st = mg_split_state(S);
  const MgSplitPlan& P = st->P;
  const MgHier H = mg_hier(S);
  const int me = P.rank, Lg = P.Lg, lC = mg_entry_level(S);
  if (S->mg_levels - 1 - lC > MG_TAIL_LEVELS) { eu_set_error("multilevel preconditioner: %d levels behind the entry level (at most %d)", S->mg_levels - 1 - lC, MG_TAIL_LEVELS); return EULER_EINVAL; }
  eu_prof_begin(S, KC_COARSE_CYCLE);
  const size_t mlen = (size_t)S->X + P.zone_doubles, ng = (size_t)S->mg_nx[Lg] * S->mg_ny[Lg];
  size_t most = ng > (size_t)st->D.total ? ng : (size_t)st->D.total;
  if ((size_t)S->X > most) most = S->X;
  hipLaunchKernelGGL(k_mg_split_unpack, dim3((unsigned)((most + 255) / 256), 4), dim3(256), 0, S->stream, H, st->D, Lg, st->msg + 2 * mlen, st->msg + 3 * mlen, S->band_lo > 0 ? zrecv_lo : nullptr,
                     S->band_hi < S->geom.nbands ? zrecv_hi : nullptr, S->mg_xbuf, P.nsmall, st->W);
  int lA = Lg;
  do {      // replicated: the gather level down to the entry level, the tail, back up to the entry level
    const int lB = lA + 3 < lC ? lA + 3 : lC;
    MgDownArgs A = mg_down_args(S, H, lA, lB, lC);
    A.force = force;
    int rc = mg_launch_down(S, A, false);
    if (rc) return rc;
    lA = lB;
  } while (lA < lC);
  MgUpArgs U = mg_up_args(S, H, lC, fin_op, force);
  if (lC > Lg) {      // ... and to the gather level
    MgUpArgs V = U;
    V.l0 = Lg;
    const unsigned nblk = (unsigned)(((S->mg_nx[Lg] + 31) / 32) * ((S->mg_ny[Lg] + 31) / 32));
    hipLaunchKernelGGL(k_mg_up, dim3(nblk), dim3(MG_UP_THREADS), 3 * (size_t)V.cap * sizeof(double), S->stream, V);
  }
  if (Lg > 1) {      // the own rows: levels Lg - 1 .. 1
    MgUpArgs V = U;
    V.l0 = 1; V.lC = Lg;
    V.tile_row0 = P.NX[1][0] / 32;
    const int trows = (P.NX[1][1] + 31) / 32 - V.tile_row0;
    hipLaunchKernelGGL(k_mg_up, dim3((unsigned)(((S->mg_nx[1] + 31) / 32) * trows)), dim3(MG_UP_THREADS), 3 * (size_t)V.cap * sizeof(double), S->stream, V);
  }
  U.tile_row0 = P.NX[0][0] / 32;
  const int trows = (P.NX[0][1] + 31) / 32 - U.tile_row0;
  U.row_lo = MG_RPB * S->part_lo[me]; U.row_hi = MG_RPB * S->part_hi[me];
  U.slot = st->gc + (size_t)me * (1 + MG_NULL_MAX);
  hipLaunchKernelGGL(k_mg_up0, dim3((unsigned)(((S->mg_nx[0] + 31) / 32) * trows)), dim3(MG_FINE_THREADS), 2 * (size_t)U.cap * sizeof(double), S->stream, U);
  eu_prof_end(S, KC_COARSE_CYCLE);
  return EULER_OK;
}
// fold: behind the all-gather of gc
int eu_mg_split_fold(euler_sim* S, int fin_op, int force) {
  MgSplitState* st = mg_split_state(S);
  const MgSplitPlan& P = st->P;
  hipLaunchKernelGGL(k_mg_split_fold, dim3(64), dim3(256), 0, S->stream, S->sc, st->gc, P.ranks, fin_op, force, mg_hier(S), S->cc_null, S->mg_null0, (size_t)S->mg_cells, S->mg_m0, P.NX[0][0], P.NX[0][1]);
  return EULER_OK;
}

int eu_mg_solve(euler_sim* S, int fin_op, int force) { return launch_mg_cycle(S, fin_op, force, !S->has_comm); }

// row slabs: from this rank's tiles, the node rows [RPB band_lo - 1, RPB band_hi + 1) of level 0's right-hand side (clipped to the grid) into `dst`
__global__ __launch_bounds__(256) void k_mg_gather_rows(const double* __restrict__ part, double* __restrict__ dst, int nx0, int row0, int row1, int ntb, int band_lo, int band_hi) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nx0 * (row1 - row0)) return;
  dst[k] = mg_gather0(part, row0 + k / nx0, k % nx0, ntb, band_lo, band_hi);
}
int eu_mg_slab_rows(euler_sim* S, int force) {
  (void)force;
  const int ny0 = S->mg_ny[0];
  const int row0 = MG_RPB * S->band_lo - 1 < 0 ? 0 : MG_RPB * S->band_lo - 1, row1 = MG_RPB * S->band_hi + 1 > ny0 ? ny0 : MG_RPB * S->band_hi + 1;
  const int cells = (row1 - row0) * S->mg_nx[0];
  if (cells > 0)
    LAUNCH(S, KC_COARSE_CYCLE, k_mg_gather_rows, dim3((cells + 255) / 256), dim3(256), S->mg_part, S->mg_xbuf + (size_t)S->bulk.rank * S->mg_xslot + 2, S->mg_nx[0], row0, row1, S->geom.T / 16,
           S->band_lo, S->band_hi);
  return EULER_OK;
}

// per solve: A_0 from the tiles, the coarser operators, the dense top level's stencil as five arrays for k_coarse_factor
__global__ __launch_bounds__(256) void k_mg_top_stencil(const double* __restrict__ at, int n, double* __restrict__ out) {      // out: [9][n] -> the factor kernel reads d, e, n, ne, nw = entries 4, 5, 7, 8, 6
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 9 * n) out[i] = at[i];
}
__global__ __launch_bounds__(256) void k_mg_inner0(MgHier H, uint8_t* __restrict__ inner, size_t first, size_t count) {      // a node whose nine entries are deep water's (bit for bit: both sides are exact)
  const size_t n = (size_t)H.nx[0] * H.ny[0], c = first + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= first + count) return;
  bool same = true, none = true;
#pragma unroll
  for (int k = 0; k < 9; ++k) { const double v = H.a[(size_t)k * n + c]; same = same && v == H.ic[k]; none = none && v == 0.0; }
  inner[c] = same ? 1 : none ? 2 : 0;      // 2 (round 6): no fluid under the node's hat - the level-0 kernels load neither its entries nor omega / d nor its right-hand side
}
__global__ __launch_bounds__(256) void k_mg_wd(const double* __restrict__ a, const unsigned int* offs, int nl, MgHier H, double* __restrict__ wd) {
  (void)a; (void)offs; (void)nl;
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int l = blockIdx.y;
  const size_t n = (size_t)H.nx[l] * H.ny[l];
  if (c >= n) return;
  const double* st = mg_sten(H, l);
  const double d = st[4 * n + c];
  double off = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) if (k != 4) off = off + fabs(st[(size_t)k * n + c]);
  double om = d != 0.0 ? MG_THETA / (1.0 + off / d) : 0.0;      // (k_mg.h MG_THETA; oracle: mg_damping - the same expressions)
  if (om > MG_OMEGA) om = MG_OMEGA;
  wd[H.off[l] + c] = d != 0.0 ? om / d : 0.0;
}
// rows of a level's nine planes <-> a message (plane-major there too); mode 0: pack, 1: unpack (replace), 2: unpack, adding on the rows [add0, add1) (level 0: both ranks'
// cells reach the node rows at a slab boundary; the entries are small multiples of 2^-12, their sums exact)
__global__ __launch_bounds__(256) void k_mg_sten_rows(double* __restrict__ a, size_t n, int nx, int row0, int rows, double* __restrict__ msg, int mode, int add0, int add1) {
  const size_t per = (size_t)rows * nx, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * per) return;
  const size_t k = i / per, e = i % per;
  double* cell = a + k * n + (size_t)row0 * nx + e;
  if (mode == 0) msg[i] = *cell;
  else {
    const int row = row0 + (int)(e / nx);
    *cell = (mode == 2 && row >= add0 && row < add1) ? *cell + msg[i] : msg[i];
  }
}
// the gather level's operator, whole, from the owners' rows
struct MgOwners { int n; int lo[64], hi[64]; };
__global__ __launch_bounds__(256) void k_mg_sten_gathered(double* __restrict__ a, size_t n, int nx, const double* __restrict__ abuf, int aslot, MgOwners W) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * n) return;
  const size_t k = i / n, c = i % n;
  const int row = (int)(c / nx);
  for (int r = 0; r < W.n; ++r)
    if (row >= W.lo[r] && row < W.hi[r]) { a[i] = abuf[(size_t)r * aslot + k * (size_t)(W.hi[r] - W.lo[r]) * nx + (size_t)(row - W.lo[r]) * nx + c % nx]; return; }
}
static int mg_setup_split(euler_sim* S, MgSplitState* st) {
  const MgSplitPlan& P = st->P;
  const int me = P.rank, R = P.ranks, Lg = P.Lg, nx0 = S->mg_nx[0];
  const size_t n0 = (size_t)nx0 * S->mg_ny[0];
  const int s0 = P.I[0][me][0], s1 = P.I[0][me][1];      // the rows this rank's tiles reach
  HIPCHK(hipMemset2DAsync(S->mg_a0i + (size_t)s0 * nx0, n0 * sizeof(unsigned long long), 0, (size_t)(s1 - s0) * nx0 * sizeof(unsigned long long), 9, S->stream));
  HIPCHK(hipMemsetAsync(S->mg_part, 0, (S->chunk_cap + 64) * MG_PART * sizeof(double), S->stream));
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_assemble0, dim3(eu_blocks(S->chunk_cap, 4, 2048)), dim3(256), S->cellmask, S->geom, S->chunk_list, S->sc, S->band_lo, nx0, S->mg_ny[0], S->mg_a0i);
  const size_t cnt0 = (size_t)(s1 - s0) * nx0;
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_convert0, dim3((unsigned)((9 * cnt0 + 255) / 256)), dim3(256), S->mg_a0i, S->mg_a, n0, (size_t)s0 * nx0, cnt0);
  const size_t blen = (size_t)9 * (MG_SETUP_HALO + 1) * nx0;
  double *send_lo = st->sbuf, *send_hi = st->sbuf + blen, *recv_lo = st->sbuf + 2 * blen, *recv_hi = st->sbuf + 3 * blen;
  const bool has_lo = me > 0, has_hi = me + 1 < R;
  for (int l = 0; l <= Lg; ++l) {
    const int nx = S->mg_nx[l], ny = S->mg_ny[l], o0 = st->O[l][me][0], o1 = st->O[l][me][1];
    double* a = S->mg_a + 9 * S->mg_off[l];
    const size_t n = (size_t)nx * ny;
    if (l > 0) {
      const size_t cnt = (size_t)9 * (o1 - o0) * nx;
      if (cnt) LAUNCH(S, KC_PRECON_FACTOR, k_mg_coarsen, dim3(mg_coarsen_blocks(nx, o0, o1)), dim3(256), S->mg_a + 9 * S->mg_off[l - 1], S->mg_nx[l - 1], S->mg_ny[l - 1], a, nx, ny, S->sc, o0, o1);
    }
    if (l == Lg) break;
    // the rows either neighbour needs of this rank / this rank of them: MG_SETUP_HALO owned rows; level 0 one more - the shared row across the boundary, as a share
    const int extra = l == 0 ? 1 : 0, rows = MG_SETUP_HALO + extra;
    const unsigned nb = (unsigned)(((size_t)9 * rows * nx + 255) / 256);
    if (has_lo) hipLaunchKernelGGL(k_mg_sten_rows, dim3(nb), dim3(256), 0, S->stream, a, n, nx, o0 - extra, rows, send_lo, 0, 0, 0);
    if (has_hi) hipLaunchKernelGGL(k_mg_sten_rows, dim3(nb), dim3(256), 0, S->stream, a, n, nx, o1 - MG_SETUP_HALO, rows, send_hi, 0, 0, 0);
    COMM_CALL(S->bulk.halo(S->bulk.ctx, send_lo, send_hi, recv_lo, recv_hi, 9 * rows * nx));
    if (has_lo) hipLaunchKernelGGL(k_mg_sten_rows, dim3(nb), dim3(256), 0, S->stream, a, n, nx, o0 - MG_SETUP_HALO, rows, recv_lo, l == 0 ? 2 : 1, s0, s1);
    if (has_hi) hipLaunchKernelGGL(k_mg_sten_rows, dim3(nb), dim3(256), 0, S->stream, a, n, nx, o1 - extra, rows, recv_hi, l == 0 ? 2 : 1, s0, s1);
  }
  {      // the gather level: the owners' rows to everybody
    const int nx = S->mg_nx[Lg], o0 = st->O[Lg][me][0], o1 = st->O[Lg][me][1];
    const size_t n = (size_t)nx * S->mg_ny[Lg];
    double* a = S->mg_a + 9 * S->mg_off[Lg];
    if (o1 > o0) hipLaunchKernelGGL(k_mg_sten_rows, dim3((unsigned)(((size_t)9 * (o1 - o0) * nx + 255) / 256)), dim3(256), 0, S->stream, a, n, nx, o0, o1 - o0, st->abuf + (size_t)me * st->aslot, 0, 0, 0);
    int64_t off[64], cnt[64];
    for (int r = 0; r < R; ++r) { off[r] = (int64_t)8 * st->aslot * r; cnt[r] = (int64_t)8 * st->aslot; }
    COMM_CALL(S->bulk.allgather(S->bulk.ctx, st->abuf, off, cnt));
    MgOwners W;
    W.n = R;
    for (int r = 0; r < R; ++r) { W.lo[r] = st->O[Lg][r][0]; W.hi[r] = st->O[Lg][r][1]; }
    hipLaunchKernelGGL(k_mg_sten_gathered, dim3((unsigned)((9 * n + 255) / 256)), dim3(256), 0, S->stream, a, n, nx, st->abuf, st->aslot, W);
  }
  for (int l = Lg + 1; l < S->mg_levels; ++l)
    LAUNCH(S, KC_PRECON_FACTOR, k_mg_coarsen, dim3(mg_coarsen_blocks(S->mg_nx[l], 0, S->mg_ny[l])), dim3(256), S->mg_a + 9 * S->mg_off[l - 1], S->mg_nx[l - 1], S->mg_ny[l - 1],
           S->mg_a + 9 * S->mg_off[l], S->mg_nx[l], S->mg_ny[l], S->sc, 0, S->mg_ny[l]);
  int i0 = st->O[0][me][0] - MG_SETUP_HALO, i1 = st->O[0][me][1] + MG_SETUP_HALO;
  if (i0 < 0) i0 = 0;
  if (i1 > S->mg_ny[0]) i1 = S->mg_ny[0];
  hipLaunchKernelGGL(k_mg_inner0, dim3((unsigned)(((size_t)(i1 - i0) * nx0 + 255) / 256)), dim3(256), 0, S->stream, mg_hier(S), S->mg_inner0, (size_t)i0 * nx0, (size_t)(i1 - i0) * nx0);
  hipLaunchKernelGGL(k_mg_wd, dim3((unsigned)((n0 + 255) / 256), (unsigned)S->mg_levels), dim3(256), 0, S->stream, (const double*)nullptr, (const unsigned int*)nullptr, 0, mg_hier(S), S->mg_wd);
  return EULER_OK;
}
int eu_mg_setup(euler_sim* S) {
  const size_t n0 = (size_t)S->mg_nx[0] * S->mg_ny[0];
  if (S->has_comm && mg_split_plan(S) && mg_split_state(S)->setup_ok) return mg_setup_split(S, mg_split_state(S));
  HIPCHK(hipMemsetAsync(S->mg_a0i, 0, 9 * n0 * sizeof(unsigned long long), S->stream));
  HIPCHK(hipMemsetAsync(S->mg_part, 0, (S->chunk_cap + 64) * MG_PART * sizeof(double), S->stream));      // (tiles outside this solve's list contribute nothing)
  const unsigned nblk = eu_blocks(S->chunk_cap, 4, 2048);
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_assemble0, dim3(nblk), dim3(256), S->cellmask, S->geom, S->chunk_list, S->sc, S->band_lo, S->mg_nx[0], S->mg_ny[0], S->mg_a0i);
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_convert0, dim3((unsigned)((9 * n0 + 255) / 256)), dim3(256), S->mg_a0i, S->mg_a, n0, (size_t)0, n0);
  // row slabs, the cycle replicated: a node row collects cells of the ranks either side of a slab boundary.  The entries are multiples of 2^-12 far below 2^37: their sums are
  // exact in any order, so ONE all-reduce makes A_0 whole and bit-identical everywhere (the split cycle, above, exchanges rows with the neighbours instead)
  if (S->has_comm) COMM_CALL(S->bulk.allreduce(S->bulk.ctx, S->mg_a, (int32_t)(9 * n0), 0));
  for (int l = 1; l < S->mg_levels; ++l)
    LAUNCH(S, KC_PRECON_FACTOR, k_mg_coarsen, dim3(mg_coarsen_blocks(S->mg_nx[l], 0, S->mg_ny[l])), dim3(256), S->mg_a + 9 * S->mg_off[l - 1], S->mg_nx[l - 1], S->mg_ny[l - 1],
           S->mg_a + 9 * S->mg_off[l], S->mg_nx[l], S->mg_ny[l], S->sc, 0, S->mg_ny[l]);
  hipLaunchKernelGGL(k_mg_inner0, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, S->stream, mg_hier(S), S->mg_inner0, (size_t)0, n0);
  // omega / diagonal per node of every level below the dense one: the cycle's Jacobi steps multiply
  hipLaunchKernelGGL(k_mg_wd, dim3((unsigned)(((size_t)S->mg_nx[0] * S->mg_ny[0] + 255) / 256), (unsigned)S->mg_levels), dim3(256), 0, S->stream, (const double*)nullptr, (const unsigned int*)nullptr, 0, mg_hier(S), S->mg_wd);
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------ P_0 at single cells (the first search direction, ghost rows, the null-space fix)
__global__ __launch_bounds__(256) void k_mg_search_init(double* __restrict__ s, const double* __restrict__ z, const uint8_t* __restrict__ mask, const double* __restrict__ y,
                                                        SkewGeom g, int nx0, int ny0, size_t e_lo, size_t e_cnt, const PcgScalars* sc) {
  if (sc->done || !sc->nonzero) return;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < e_cnt; k += (size_t)gridDim.x * blockDim.x) {
    const size_t e = e_lo + k;
    double v = z[e];
    if (mask[e] & CM_FLUID) {
      int band, t, l;
      skew_decode(g, e, band, t, l);
      v = v + mg_interp0(y, nx0, ny0, t - l, band * 64 + l);
    }
    s[e] = v;
  }
}
int eu_mg_search_init(euler_sim* S) {
  LAUNCH(S, KC_UPDATE_SEARCH, k_mg_search_init, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), S->s, S->z, S->cellmask, S->mg_x, S->geom, S->mg_nx[0], S->mg_ny[0], S->e_lo, S->e_cnt, S->sc);
  return EULER_OK;
}
__global__ __launch_bounds__(256) void k_mg_add_row(double* __restrict__ row, const double* __restrict__ y, int X, int yrow, int nx0, int ny0, const PcgScalars* sc) {
  if (sc->done || !sc->nonzero) return;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < X) row[x] = row[x] + mg_interp0(y, nx0, ny0, x, yrow);
}
int eu_mg_add_row(euler_sim* S, double* row, int yrow) {
  LAUNCH(S, KC_UPDATE_SEARCH, k_mg_add_row, dim3((S->X + 255) / 256), dim3(256), row, S->mg_x, S->X, yrow, S->mg_nx[0], S->mg_ny[0], S->sc);
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------ water cut off from the air: what the gauge of k_mg_up needs, per solve
// m_0[q] = P_0^T (the indicator of region q on the cells): a cell's four weights (multiples of 1 / 256: integer atomics, exact) where the level-0 indicator, sampled
// through P_0, says "inside"; then m_0 . n_0.  Both kernels return at once when no region is cut off (the usual case).
__global__ __launch_bounds__(256) void k_mg_null_mass(const uint8_t* __restrict__ mask, SkewGeom g, size_t e_lo, size_t e_cnt, const double* __restrict__ nullv, const double* __restrict__ n0,
                                                      size_t nstride, int nx0, int ny0, unsigned long long* __restrict__ acc, const PcgScalars* sc) {
  const int count = (int)nullv[MG_NULL_MAX * 256];
  if (!sc->nonzero || count <= 0) return;
  const size_t n0n = (size_t)nx0 * ny0;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < e_cnt; k += (size_t)gridDim.x * 256) {
    const size_t e = e_lo + k;
    if (!(mask[e] & CM_FLUID)) continue;
    int band, t, l;
    skew_decode(g, e, band, t, l);
    const int x = t - l, y = band * 64 + l;
    int jx0, jx1, jy0, jy1;
    double fx, fy;
    mg_cell_w(x, nx0, jx0, jx1, fx);
    mg_cell_w(y, ny0, jy0, jy1, fy);
    const int wx1 = (int)(fx * MG_G0), wx0 = MG_G0 - wx1, wy1 = (int)(fy * MG_G0), wy0 = MG_G0 - wy1;
    for (int q = 0; q < MG_NULL_MAX && q < count; ++q) {
      if (!(mg_interp0(n0 + (size_t)q * nstride, nx0, ny0, x, y) > 0.5)) continue;
      unsigned long long* a = acc + (size_t)q * n0n;
      if (wy0 * wx0) atomicAdd(&a[(size_t)jy0 * nx0 + jx0], (unsigned long long)(wy0 * wx0));
      if (wy0 * wx1) atomicAdd(&a[(size_t)jy0 * nx0 + jx1], (unsigned long long)(wy0 * wx1));
      if (wy1 * wx0) atomicAdd(&a[(size_t)jy1 * nx0 + jx0], (unsigned long long)(wy1 * wx0));
      if (wy1 * wx1) atomicAdd(&a[(size_t)jy1 * nx0 + jx1], (unsigned long long)(wy1 * wx1));
    }
  }
}
__global__ __launch_bounds__(256) void k_mg_null_convert(const unsigned long long* __restrict__ acc, const double* __restrict__ nullv, size_t n, double* __restrict__ m0, const PcgScalars* sc) {
  const int count = (int)nullv[MG_NULL_MAX * 256];
  if (!sc->nonzero || count <= 0) return;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) m0[i] = (double)acc[i] * (1.0 / ((double)MG_G0 * MG_G0));
}
__global__ __launch_bounds__(1024) void k_mg_null_finish(const double* __restrict__ nullv, const double* __restrict__ n0, size_t nstride, size_t n0n, double* __restrict__ m0, const PcgScalars* sc) {
  const int count = (int)nullv[MG_NULL_MAX * 256];
  if (!sc->nonzero || count <= 0) return;
  __shared__ double s_red[16];
  const int q = blockIdx.x;
  if (q >= count) { if (threadIdx.x == 0) m0[(size_t)MG_NULL_MAX * n0n + q] = 0.0; return; }
  double t = 0.0;
  for (size_t c = threadIdx.x; c < n0n; c += 1024) t += m0[(size_t)q * n0n + c] * n0[(size_t)q * nstride + c];
  t = eu_wave_sum(t);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) { double v = 0.0; for (int k = 0; k < 16; ++k) v += s_red[k]; m0[(size_t)MG_NULL_MAX * n0n + q] = v; }
}
int eu_mg_null_setup(euler_sim* S) {      // behind k_mg_null_prolong (k_coarse.hip: eu_launch_coarse_consistent)
  const size_t n0n = (size_t)S->mg_nx[0] * S->mg_ny[0];
  if (S->has_comm) {
    // row slabs: a region's cells lie on several ranks, so m_0 is a sum over the ranks - an all-reduce of 4 n_0 doubles that only a job with a cut-off region needs.  Whether
    // there is one is the same on every rank (the dense level is replicated): one host round trip per solve decides (row-slab handles only)
    double cnt = 0.0;
    HIPCHK(hipMemcpyAsync(&cnt, S->cc_null + MG_NULL_MAX * 256, sizeof(double), hipMemcpyDeviceToHost, S->stream));
    HIPCHK(hipStreamSynchronize(S->stream));
    if (!(cnt > 0.0)) return EULER_OK;
  }
  HIPCHK(hipMemsetAsync(S->mg_a0i, 0, MG_NULL_MAX * n0n * sizeof(unsigned long long), S->stream));      // (A_0's integer sums are converted by now: the array is scratch)
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_null_mass, dim3(eu_blocks(S->e_cnt, 256 * 8, 512)), dim3(256), S->cellmask, S->geom, S->e_lo, S->e_cnt, S->cc_null, S->mg_null0, S->mg_cells,
         S->mg_nx[0], S->mg_ny[0], S->mg_a0i, S->sc);
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_null_convert, dim3((unsigned)((MG_NULL_MAX * n0n + 255) / 256)), dim3(256), S->mg_a0i, S->cc_null, MG_NULL_MAX * n0n, S->mg_m0, S->sc);
  // m_0 . n_0 from this rank's share of m_0 - it lies on the rows its cells reach, where the rank holds the indicator whatever the cycle's form (a split cycle prolongs it on
  // the own rows only) - then shares and scalars are summed over the ranks together (the shares: multiples of 1 / 64, exact in any order)
  LAUNCH(S, KC_PRECON_FACTOR, k_mg_null_finish, dim3(MG_NULL_MAX), dim3(1024), S->cc_null, S->mg_null0, S->mg_cells, n0n, S->mg_m0, S->sc);
  if (S->has_comm) COMM_CALL(S->bulk.allreduce(S->bulk.ctx, S->mg_m0, (int32_t)(MG_NULL_MAX * n0n + MG_NULL_MAX), 0));
  return EULER_OK;
}
