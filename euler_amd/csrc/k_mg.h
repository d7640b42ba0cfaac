// k_mg.h — internal: what the multilevel preconditioner's coarse part (k_mg.hip) shares with the fine-grid passes of k_pcg.hip and with k_coarse.hip.
#pragma once

#include "euler_dev.h"

#define MG_OMEGA 0.8         // damped Jacobi on every level below the dense one (tools/r05/mg_proto.py: 0.7 - 1.0 are within two iterations of each other, 1.1 costs 40 %)
// The Jacobi steps damp per node, omega_i = min(MG_OMEGA, MG_THETA / (1 + sum |off-diagonals| / d)): Gershgorin then keeps D~^-1 A below MG_THETA < 2 everywhere.  A regular node's
// off-diagonals add up to its diagonal (0.8 either way); a drop of spray - one fluid cell between four nodes - is a rank-one block of eigenvalue 4 d that plain omega = 0.8 multiplies
// by 1 - 3.2 per step: as the waterfall filled with spray the cycle stopped approximating the coarse solve and PCG fell back to the tile-local mode's iteration counts (oracle: mg_damping)
#define MG_THETA 1.6
#define MG_TOP_MAX 64        // nodes of the dense top level at most (its inverse lives in LDS: 32 KB)
#define MG_MAXLEV 12
// Level 0's node spacing in grid cells: node (I, J) sits AT the centre of cell (G0 J + G0 / 2, G0 I + G0 / 2).  8 (round 5): a tank at rest needs 30 PCG iterations to 1e-6 where
// 16 - the tile width - needs 52 and round 4's aggregates of 16 needed 104 (tools/r05/mg_proto.py; dam break at impact 39 / 65 / 109, waterfall 37 / 63 / 124), for a level 0 of
// four times the nodes.  Everything below follows from it: a band of 64 rows holds MG_RPB node rows; the lanes 8 G - 4 .. 8 G + 3 of a tile (GROUP G = (lane + 4) >> 3: 0 .. 8,
// the first and the last half groups) lie between the same two node rows I0 = 8 band + G - 1 and I0 + 1, and their 16 + 7 columns between the four node columns
// Jq .. Jq + 3, Jq = 2 k - G - 1 (k: the tile's index in its band): a tile leaves 9 groups x 2 node rows x 4 node columns = 72 sums (160 per tile with groups of four lanes,
// until the two quads between the same node rows were merged: the gather of k_mg.hip then takes 4 loads per node instead of 16).
#ifndef MG_G0
#define MG_G0 8
#endif
#define MG_LOG 3
#define MG_RPB (64 / MG_G0)
#define MG_LG 4                                // the quad: what a DPP butterfly adds in registers; the two quads of a group meet in LDS
#define MG_NGRP 9
#define MG_NSEG (16 / MG_G0 + 1)               // node intervals a lane's 16 columns can touch
#define MG_NSLOT 4
#define MG_PART (MG_NGRP * 2 * MG_NSLOT)       // doubles k_precond_tile leaves per tile: group x row slot x column slot, stored [band][group][tile][row slot][column slot] (mg_gather0)
#define MG_NI ((19 + MG_G0 - 1) / MG_G0 + 1)  // node intervals the 20 records of a k_search_apply run and its window can touch
static_assert(MG_G0 == 8, "MG_G0: the tiles' partial sums (k_precond_tile, mg_gather0) are laid out for nodes 8 cells apart");
#define MG_NULL_MAX 4        // indicators of fluid regions cut off from the air that are kept (k_coarse.hip CC_NULL_MAX)
#define MG_DOT_BLOCKS 4096   // workgroups of k_mg_up at most (tiles of 32 x 32 nodes of level 0: 16384^2 has 4096 with G0 = 8)
#define MG_FIN_SLOT 7        // k_mg_up's epilogue on row slabs with a split cycle: x_0 . rhs_0 of the own rows is ADDED to the rank's slot instead of applied

int  eu_mg_alloc(euler_sim* S);
void eu_mg_release(euler_sim* S);
int  eu_mg_setup(euler_sim* S);                                   // per solve: the operators of every level (the dense top level's stencil is S->mg_a + 9 * S->mg_off[top])
int  eu_mg_solve(euler_sim* S, int fin_op, int force);            // per iteration: the V-cycle, x_0 . rhs_0 into dot(z, r), the scalar epilogue
int  eu_mg_search_init(euler_sim* S);                             // s = z + P_0 x_0
int  eu_mg_add_row(euler_sim* S, double* row, int yrow);          // + P_0 x_0 on a compact row of cells
int  eu_mg_null_setup(euler_sim* S);                              // per solve, behind the indicators' way down the levels: what the gauge of a cut-off region needs
int  eu_mg_slab_rows(euler_sim* S, int force);                   // row slabs: this rank's share of level 0's right-hand side into its slot of the exchange buffer
// row slabs, the cycle split by rows (k_mg.hip "row slabs: the cycle split by rows"): whether this handle runs it, the all-gather's doubles per rank, the neighbour messages
// (k = 0, 1 send to the rank below / above, 2, 3 what arrived from them; doubles each: eu_mg_split_count), the ranks' {x_0 . rhs_0, gauge sums}
bool eu_mg_split(euler_sim* S);
int  eu_mg_split_nsmall(euler_sim* S);
int  eu_mg_split_count(euler_sim* S);
double* eu_mg_split_msg(euler_sim* S, int k);
double* eu_mg_split_gc(euler_sim* S);
int  eu_mg_split_pre(euler_sim* S);
int  eu_mg_split_mid(euler_sim* S, int fin_op, int force, double* zrecv_lo, double* zrecv_hi);
int  eu_mg_split_fold(euler_sim* S, int fin_op, int force);
void eu_mg_split_release(euler_sim* S);

#ifdef __HIPCC__
// A cell's column c (or row) against n nodes, node j AT the centre of cell G0 j + G0 / 2: the node left of / at the cell and the weight f (a multiple of 1 / G0) of the next one;
// beyond the outermost nodes the interpolant is constant (weight clamps, oracle: mg_w0)
__device__ __forceinline__ void mg_cell_w(int c, int n, int& j0, int& j1, double& f) {
  const int u = c - MG_G0 / 2;
  int j = u >> MG_LOG;
  double w = (double)(u & (MG_G0 - 1)) * (1.0 / MG_G0);
  if (j < 0) { j = 0; w = 0.0; }
  int k = j + 1;
  if (k > n - 1) { k = n - 1; w = 0.0; }
  j0 = j; j1 = k; f = w;
}
// the two values a row of cells sees at a node column: rows combined first (oracle mg_interp0: lo / hi), then mg_lerp_x along the row
__device__ __forceinline__ double mg_rows(double v0, double v1, double fy) { return (1.0 - fy) * v0 + fy * v1; }
__device__ __forceinline__ double mg_lerp_x(double lo, double hi, double fx) { return (1.0 - fx) * lo + fx * hi; }
// (P_0 y) at cell (x, yrow)
__device__ __forceinline__ double mg_interp0(const double* __restrict__ y, int nx0, int ny0, int x, int yrow) {
  int jx0, jx1, jy0, jy1;
  double fx, fy;
  mg_cell_w(x, nx0, jx0, jx1, fx);
  mg_cell_w(yrow, ny0, jy0, jy1, fy);
  const double lo = mg_rows(y[(size_t)jy0 * nx0 + jx0], y[(size_t)jy1 * nx0 + jx0], fy);
  const double hi = mg_rows(y[(size_t)jy0 * nx0 + jx1], y[(size_t)jy1 * nx0 + jx1], fy);
  return mg_lerp_x(lo, hi, fx);
}
#endif
