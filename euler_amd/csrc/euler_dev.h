// euler_dev.h — internal: device-resident state, launch plumbing and the device functions shared
// by the kernel files of libeuler_hip.so.  gfx950 (MI355X, wave64) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "euler.h"
#include "euler_host.h"

// ------------------------------------------------------------------------------------------
// reference constants (main.c:58-60)
#define EU_H 1.0f        // k_side_length
#define EU_RHO 1.0f      // k_density
#define EU_G (-10.0f)    // k_gravity

// cell mask byte (EULER_F_CELLMASK)
#define CM_FLUID 0x01
#define CM_RIGHT 0x02   // fluid at (x+1,y)
#define CM_UP    0x04   // fluid at (x,y+1)
#define CM_LEFT  0x08   // fluid at (x-1,y)
#define CM_DOWN  0x10   // fluid at (x,y-1)
#define CM_DIAG_SHIFT 5 // a_diag = mask >> 5  (0..4)
#define CM_INTERIOR 0x9F   // fluid, four fluid neighbours, a_diag 4: the mask of a cell deep inside the water
#define EU_CHUNK_INTERIOR 0x80000000u   // chunk_list entry: every cell of the chunk is CM_INTERIOR (euler_sim.chunk_list)

// kernel classes for euler_profile_*
enum {
  KC_TIMESTEP = 0, KC_MARKER_ADVECT, KC_MARKER_EVENTS, KC_MARKER_BIN, KC_MARKER_COMPACT, KC_SOURCES,
  KC_SELECT, KC_EXTRAPOLATE, KC_ADVECT_VELOCITY, KC_BUILD_SYSTEM, KC_PRECON_FACTOR,
  KC_FORWARD_SOLVE, KC_BACKWARD_SOLVE, KC_APPLY_A, KC_DOT, KC_UPDATE_PR, KC_UPDATE_SEARCH,
  KC_REDUCE_FINAL, KC_VELOCITY_UPDATE, KC_JACOBI, KC_MISC, KC_PRECOND_TILE, KC_COARSE_CYCLE, KC_RESIDENT, KC__COUNT
};

// Device-resident PCG scalars: no host round trip inside the iteration (reference: locals of
// project(), main.c:748-765).
struct PcgScalars {
  double sigma, zs, sigma_new, alpha, beta, rnorm;
  double tol;
  double comm_val;   // multi-rank: a rank-local reduction result on its way through the all-reduce
  double comm_val2;  // k_precond_tile leaves two: max |r| in comm_val, dot(z,r) here
  double* comm_slot; // several ranks: this rank's slot of the small all-gather buffer (euler_comm_ops.exchange), set with the communicator
  double alpha_prev; // alpha of the iteration before the last one that ran
  double alpha_hist[8];   // alpha of iteration k at [k & 7]: p += alpha s is applied N iterations at a time (N = 2, 4 or 8 fmadds of main.c:753, in their order) by the
                          // k_search_apply pass of every N-th iteration, which reads s_(k-1) anyway, finds s_(k-N) in the array it is about to overwrite (the search
                          // directions live in a ring of N arrays) and fetches the N - 2 in between; k_finish_p applies what is left when a solve ends
  int nonzero;   // !all_zero(r)  (main.c:742)
  int done;      // inf_norm(r) <= tol reached (main.c:756)
  int iters;     // apply_a calls so far (main.c:750)
  int max_iters;
  unsigned int n_chunks, pad_;   // active 16-record chunks of this solve (euler_sim.chunk_list)
  // peer-to-peer mailboxes (comm_p2p.hip), set once by euler_p2p_connect and never reset: the mapped mailboxes of all
  // ranks, this rank, and the DEVICE-side exchange counters (only exchanges that really run count, so that the tags of
  // all ranks stay in step although launches after convergence return at once)
  void** p2p_boxes;
  int* p2p_error;
  int p2p_rank, p2p_n;
  unsigned int p2p_seq, p2p_halo_seq;
};

// Device-resident marker bookkeeping (reference: g_markers_length, g_source_exhausted, the
// function-static rng_state; main.c:93-94,204).
struct MarkerState {
  unsigned long long n;            // g_markers_length
  unsigned long long max_markers;  // MAX_MARKER_COUNT = 4*X*Y
  unsigned long long rng_state;
  int exhausted;
  unsigned int n_events;      // candidate dt-shortening collisions found by the speculative pass
  unsigned int n_actual;      // those that fire in array order
  unsigned int n_deleted;     // markers removed by the last refresh
  unsigned int n_append;      // markers appended by the last source update
  unsigned long long n0_append;  // index of the first appended marker
  unsigned long long total_dt_events, multi_events;
  float dt_final;             // dt after the last marker (debug)
  float dt;                   // calculate_timestep result
  unsigned int max_u2_bits, max_v2_bits;
  unsigned int dt_ticket;     // k_maxsq: blocks that have delivered their maxima (the last one forms dt on a whole-grid handle)
  int error;                  // sticky device-side error (bounded waits)
  // row slabs (k_slab.hip): n above is the GLOBAL marker count (the reference's g_markers_length); this rank holds n_loc of them
  unsigned long long n_loc;
  unsigned int n_rm;          // local markers leaving the array in this refresh (deleted here + migrated away)
  unsigned int n_del_glob;    // markers deleted on all ranks together in this refresh
  unsigned int n_recv;        // markers received from the neighbouring slabs in this substep
  unsigned int src_k_lo;      // sources: eligible cells on lower ranks (= this rank's first index into the substep's append order)
  unsigned long long rng0;    // sources: the generator's state at the start of the substep's draws (the parallel draw kernel jumps from it)
};

// Band-skewed layout of the solver's private arrays (k_pcg.hip header): element (x,y) lives at
// record t = x + y%64 of band y/64, lane y%64; RECORDS COME IN PAIRS: the two elements (t even, t+1) of a lane
// are adjacent, index = ((y/64)*TS + (t & ~1))*64 + 2*(y%64) + (t & 1), so that one 16-byte access per lane
// serves two steps of the IC(0) sweeps (a lone wave pays per memory INSTRUCTION: tools/micro/step_bench2).  A band has X + 63 live records,
// T = that rounded up to whole units of 96 records (the unit of the tile-local preconditioner: a multiple of the sweeps' 32- and
// 24-step loop bodies, so that a tile is a whole number of bodies in either direction) and a stride of
// TS = T + 64 records; S = nbands*TS*64 elements in total; padding carries mask 0.
// Every skewed array also has EU_SKEW_SLACK elements of zeroed slack in front of element 0.
#define EU_SKEW_SLACK (64 * 64)
#ifndef EU_ARRAY_STAGGER
#define EU_ARRAY_STAGGER 0   // bytes (driver.hip: euler_create)
#endif
#ifndef EU_RED_ELEMS
#define EU_RED_ELEMS (256 * 16)   // skewed elements per block of the reduction kernels (apply_a, update_pr, dot), at most 2048 blocks;
                                  // 1024^2 frame: 2048 -> 79.8 ms, 4096 -> 77.0 ms, 8192 -> 79.5 ms
#endif
struct SkewGeom {
  int X, Y, nbands, T, TS;
  size_t S;
};
static inline __host__ __device__ size_t skew_index(const SkewGeom& g, int x, int y) {
  const int l = y & 63, t = x + l;
  return ((size_t)(y >> 6) * g.TS + (size_t)(t & ~1)) * 64 + 2 * l + (t & 1);
}
// inverse within the skewed arrays: element -> (band, record t, lane)
static inline __host__ __device__ void skew_decode(const SkewGeom& g, size_t e, int& band, int& t, int& l) {
  const size_t per_band = (size_t)g.TS * 64;
  band = (int)(e / per_band);
  const size_t r = e % per_band;
  l = (int)((r & 127) >> 1);
  t = (int)((r >> 7) << 1) + (int)(r & 1);
}

struct SelectScratch {
  unsigned int* block_sums;   // [nblocks]
  unsigned int* total;        // [1]
  size_t capacity_blocks;
};

struct euler_sim {
  euler_config cfg;
  int X, Y;
  size_t C;
  hipStream_t stream;
  int loaded;
  long long opt[32];      // euler_set_option (include/euler.h EULER_OPT_*), defaults set by euler_create
  int dot_mode_user;      // the dot mode the caller's configuration resolved to (the coarse modes force EULER_DOT_TREE while they are selected; leaving them restores this)

  // fields (main.c:64-73,96-97)
  float *u, *v, *utmp, *vtmp;
  uint8_t *solid, *source, *sink, *count, *prev_count;
  unsigned int* count32;   // the binning counters of the window, COLUMN-major: [x][y - win_lo] (k_markers.hip), never shifted
  int p_pending;           // k_velocity_update_para finished and clamped the pressure in LDS only: 1 - memory still lacks the last fmadds and the clamp, 2 - the clamp (eu_pressure_current)
  int maxsq_state;         // ms->max_u2_bits / max_v2_bits: 0 - zero, 2 - the maxima of u, v as they stand (k_velocity_update_para), 1 - maxima of a state that has been edited since
  int uv_clean;            // 1: the last velocity update left every sample without the fluid property / with the solid property zero; 2: ... and exactly one refresh_marker_counts has run since (k_zero_bounds4<true>); 0: unknown
  int uv_zb;               // zero_bounds has run with the count grid as it stands (k_velocity_update_para need not store the zeros of the air and the walls again)
  int count32_dirty;       // the counters hold something (between a binning launch and the k_narrow_counts<true> that clears them; whole-grid handles)
  int prebin_valid;        // the advection stage in front binned its own output (k_advect_bin_a2): the next refresh keeps the counters and the delete ballot
  unsigned long long* delmask;   // [ceil(max_markers/64)] that pass's delete ballot (whole-grid handles)
  // Round 6, the TILE MAP (whole-grid handles): per 64 x 64 cells, "some cell of the tile holds markers" for the count grid as it stands (tmap) and for the previous one
  // (tmap + tmap_n) - a superset: a set flag over an empty tile is harmless.  k_narrow_counts<true> writes both at every refresh, k_source_place sets what it fills; the
  // grid passes whose output over a tile with no water in or next to it in EITHER grid is the zeros already there leave at once (k_advect_velocity, k_zero_bounds4<true>,
  // k_extrapolate4, k_transpose_for_markers; with the count grid's flags alone: k_build_system_para<true>, k_velocity_update_para; k_narrow_counts<true> with the third
  // array, below): most of a dam break's grid.  DESIGN.md section 4 "tiles without water".
  uint8_t* tmap;
  int tmap_nx, tmap_n;     // tiles per row; tiles in all (the previous grid's flags start at tmap + tmap_n; at tmap + 2 tmap_n: a marker has entered the tile since the last refresh)
  int tmap_valid;          // both maps describe count / prev_count (0: somebody else wrote the grids since the last refresh)
  int utmp_clean;          // 1: utmp / vtmp are zero wherever the typed fluid property of the count grid does not hold (k_advect_velocity wrote them); 2: ... of prev_count (one refresh since); 0: unknown
  int countT_clean;        // the same for countT (k_transpose_for_markers)
  uint8_t* blockedT;       // sink | solid per cell, COLUMN-major like count32 (whole-grid handles; k_bin_markers); rebuilt when blocked_dirty
  int blocked_dirty;
  float *uT, *vT; uint8_t *countT, *solidT;   // COLUMN-major copies of u, v (made in front of every marker advection), per cell the typed fluid properties of an interpolation's four corners (countT: k_transpose_for_markers) and of the solid grid (when blocked_dirty): whole-grid handles only
  int solidT_dirty;
  float* dye[6];          // --rainbow only (cfg.rainbow): g_r, g_g, g_b, g_rtmp, g_gtmp, g_btmp (main.c:76-81)
  // markers, ping-pong (main.c:95)
  float2* markers[2];
  int cur;
  size_t max_markers;
  unsigned long long n_markers_host;   // exact at substep start
  size_t n_source_cells;
  MarkerState* ms;        // device
  MarkerState* ms_host;   // pinned mirror
  // marker scratch
  unsigned long long* evmask;   // [ceil(max_markers/64)] also reused as delete mask
  float* ev_theta;              // [max_markers] written only where an event exists
  float* ev_delta;
  unsigned int* sel_idx;        // ordered selection output (events / deletions / eligible cells)
  size_t sel_cap;
  unsigned int* act_idx; float* act_dt;   // actual events
  unsigned long long* cellmask64;         // [ceil(C/64)] eligibility mask
  float* draws;                           // [2*n_source_cells]
  SelectScratch sel;

  // pressure solve (main.c:552,577-578,716-745): band-skewed arrays of geom.S elements
  SkewGeom geom;
  double *b, *p, *r, *z, *s, *q, *precon;
  double* s2;             // second search-direction array (k_search_apply ping-pongs s / s2)
  void* skew_alloc[9];    // the raw allocations behind b p r z s s2 q precon cellmask (each array is staggered inside its own)
  uint8_t* cellmask;
  uint8_t* sys_m; float* sys_div;   // row-major scratch of the assembly (k_cell_system -> k_build_system): mask byte and float divergence per cell of the window, indexed i - win_off
  unsigned int* fbits_fwd; unsigned int* fbits_bwd;   // fluid flags of the sweeps, 8 steps to a dword (k_pack_fbits)
  int fb_stride;          // words per band and lane
  int4* band_ranges;      // per band: active block ranges of the forward / backward sweeps (per solve)
  // EULER_PRECOND_IC0_TILE2 (two-level; k_coarse.hip): coarse cells of g x g grid cells, g = 64 m = 1 << coarse_shift, nx x ny of them (<= 256)
  int coarse_m, coarse_shift, coarse_nx, coarse_ny, coarse_n;
  int *cc_diag, *cc_right, *cc_up;   // P^T A P as a 5-point stencil over the coarse grid: integer sums of A's entries (exact, order-free)
  int* cc_pinned;                    // [n]: dense-level cells the factor pinned (water cut off from the air)
  double* cc_sten;                   // [5][n]: that stencil as doubles (d, e, n, ne, nw): what k_coarse_factor reads
  int cc_n;                          // size of the dense level of the current solve (two-level: coarse_n; multilevel: the top level's nodes)
  double* cc_fac;                    // [n][n]: the dense (banded) Cholesky factor of the dense level's matrix, per solve
  double* cc_inv;                    // [n][n]: its inverse, per solve
  double* cc_part;                   // [chunks][3]: per tile, the sums of r over its fluid cells by coarse column (k_precond_tile)
  double* cc_y;                      // [n]: the coarse correction of the iteration (k_coarse_solve)
  double* cc_null;                   // [4][CC_MAX] + 1: the indicators of up to four fluid regions cut off from the air (null vectors of the dense level) and, last, how many (k_coarse_nullfix)
  // EULER_PRECOND_IC0_TILE_MG (multilevel; k_mg.hip): levels 0 .. mg_levels - 1 of node grids, mg_nx[l] x mg_ny[l] nodes (level 0: a node per MG_G0 x MG_G0 = 8 x 8 cells, AT the centre of cell (8 J + 4, 8 I + 4) - k_mg.h; each level
  // above every other node of the one below; the last level is the dense one), pooled arrays with level l at offset mg_off[l]
  int mg_levels, mg_nx[12], mg_ny[12];
  size_t mg_off[12], mg_cells;
  double* mg_a;                      // [9][nodes] per level (level l at 9 * mg_off[l]): the Galerkin operators as nine-point stencils
  unsigned long long* mg_a0i;        // [9][nodes of level 0]: A_0 in units of 1 / MG_G0^4 = 2^-12, summed by integer atomics per solve
  double *mg_rhs, *mg_x;             // right-hand side and result of the V-cycle per level (level 0's result is what k_search_apply adds to z through P_0)
  double* mg_wd;                     // omega / diagonal per node (0: no fluid under the node), per solve
  uint8_t* mg_inner0; double mg_ic[9];   // level 0: nodes whose stencil is deep water's (mg_ic, a constant of the node spacing), per solve
  double* mg_part;                   // MG_PART = 72 doubles per tile, stored [band][group of 8 lanes (9: the first and the last are half groups)][tile][row slot (2)][column slot (4)] - the weighted sums of r (k_precond_tile writes, mg_gather0 reads; k_mg.h)
  double* mg_dot;                    // per-workgroup partials of x_0 . rhs_0 (+ the ticket counters behind them)
  double* mg_null0;                  // [4][mg_cells]: the indicators of cut-off regions on every level (k_mg_null_prolong)
  double* mg_m0;                     // [4][nodes of level 0] + [4]: P_0^T of those indicators on the cells, and m_0 . n_0 (the gauge of k_mg_up)
  void* mg_split;                    // row slabs: the plan and buffers of the split cycle (k_mg.hip MgSplitState), built on first use
  double* mg_xbuf; int mg_xslot;     // row slabs: [ranks][mg_xslot] - every rank's {max |r|, dot(z,r), its share of the level-0 right-hand side}, ONE all-gather inside the G1 exchange
  // the resident solver (k_resident.hip): the tile-local PCG of a grid whose chunks all find a wave on the chip at once, in ONE persistent launch
  unsigned long long* res_gran;   // [2][3][768] 16-byte {value, generation} granules of its grid-wide reductions
  unsigned long long res_tag;     // generation of the next launch's first reduction (never reset: stale granules never match)
  int* res_err;                   // pinned host word the kernel raises when a wait ran out; the host then solves with the multi-kernel path
  int res_disabled;               // ... and stops using the kernel on this handle
  unsigned int res_last_chunks;   // active chunks of the previous solve (from sc_host): decides whether the next one is launched resident without asking the device
  int res_have_last, res_skip_once;
  unsigned long long res_solves, res_fallbacks;
  int pcg_fields_resident;        // the last solve ran in the resident kernel: S->z / S->s / S->q hold published halo cells only (euler_get_field refuses them)
  int tile_w;             // EULER_PRECOND_IC0_TILE: records per tile (include/euler.h precond_tile_records)
  double* s_ring[8];                // the search directions' ring (k_pcg.hip p_steps): [0], [1] = s, s2 as the solve found them, the others allocated when first needed
  void* s_ring_alloc[8];            // raw allocations of [2..]
  double* s_base[2];                // the handle's own two search arrays (S->s / S->s2 point into the ring while a solve runs)
  int s_ring_n;                     // arrays of the current / last solve's ring
  int s_launched;                   // iterations the last solve LAUNCHED (launches behind convergence return at once but still turn the ring)
  int s_none;                       // the last solve's right-hand side was all zero (main.c:742): no search direction exists, EULER_F_PCG_S reads as +0
  int s_stale;                      // non-fluid elements of the search arrays hold values of earlier solves (nothing reads them unmasked): EULER_F_PCG_S shows them as +0
  const double* tile_as_override;   // where A s sits for the r update of the non-tile modes (z behind the first apply_a of a solve, else q)
  double* partial2;       // second set of reduction partials (k_precond_tile reduces max |r| and dot(z,r) at once)
  struct RngJump* rng_jump;   // xorshift64* jump-ahead matrices (device)
  // Active chunks of a solve: a chunk = 16 consecutive records of a band = one tile of the tile-local preconditioner (W = 16) = one
  // run of k_search_apply (8 pair-records).  k_build_system flags every chunk that holds fluid (a byte each, packed to bits), an ordered select turns the bits
  // into the ascending list both kernels walk - no wave reads the masks of an empty chunk, and a listed chunk's masks travel with its data.
  uint8_t* chunk_flag;    // this solve: the chunk holds fluid
  uint8_t* chunk_prev;    // the previous solve's flags (k_build_system<true>: where stale masks / p / r may sit)
  uint8_t* chunk_part;    // this solve: some cell of the chunk is not CM_INTERIOR (the listed entry of an interior chunk carries EU_CHUNK_INTERIOR)
  size_t hbm_bytes;       // device memory this handle allocated: at creation, plus the search directions' ring when the first multi-kernel solve needs it (euler_hbm_bytes)
  int lean_ok;            // the solver arrays have only been written by solves since chunk_prev was current (else k_build_system writes them whole)
  double* tile_table;     // [8][64][2]: E^-1 of an interior tile of 16 records (k_tile_table) - the same for every interior tile, so k_precond_tile never streams it
  unsigned long long* chunk_bits;
  unsigned int* chunk_list;
  size_t chunk_words, chunk_cap;
  // several ranks, tile-local mode without mailboxes ("ghost rows", k_pcg.hip): compact edge rows of X doubles each
  double* xrows;          // one allocation: z send lo / hi, z recv lo / hi, the ghost rows of s below / above (two generations each)
  size_t xrow_len;        // doubles per row (X rounded up)
  int gs_cur;             // which generation of the ghost rows of s holds the current search direction
  double* alpha_buf;      // [ranks]: every rank's partial of dot(s, A s), exchanged and folded in rank order
  double* pair_buf;       // [ranks][2]: every rank's {max |r|, dot(z,r)} of one iteration, exchanged by ONE all-gather (tile-local mode without mailboxes)
  double* rowmajor_tmp;   // lazily allocated C doubles for euler_get/set_field of skewed arrays
  PcgScalars* sc;
  PcgScalars* sc_host;    // pinned
  PcgScalars* poll_host;  // pinned [2]: convergence polls taken one chunk late (eu_launch_project)
  hipEvent_t poll_event[2];
  double* partial;        // reduction partials
  unsigned int* red_counter;   // arrival ticket of the "last block reduces" epilogue (self-resetting)
  int red_blocks;
  // band sweep
  unsigned long long* granules;  // [nbands][gran_stride][2]
  int gran_stride;
  unsigned int* ticket;
  unsigned int ticket_base;
  unsigned int epoch;
  unsigned long long* sweep_timeline;   // [nbands][4], written by every band sweep (euler_sweep_timeline)

  float interp_lim[4];    // nextafterf(extent-1, 0) for U.x, U.y, V.x, V.y (main.c:339-340)

  euler_stats stats;

  // multi-GPU slab decomposition of the pressure solve (include/euler.h euler_comm_ops)
  euler_comm_ops comm;
  int has_comm, couple;
  int band_lo, band_hi;       // this rank's bands [lo, hi)
  size_t e_lo, e_cnt;         // the same range in skewed elements
  double* halo_buf;           // 4 rows of X doubles: send_lo, send_hi, recv_lo, recv_hi
  int own_stream;
  // Row slabs for EVERY stage (euler_config.slab_nranks > 1; k_slab.hip): this rank owns the rows of its bands,
  // [row_lo, row_hi).  Every row-major array holds only the window [win_lo, win_hi) = owned rows + EU_GHOST_LO rows below +
  // EU_GHOST_HI above (clipped to the grid) and is addressed with GLOBAL (x, y): the base pointers are shifted by
  // -win_lo * X (win_off).  The band-skewed arrays hold the own bands + one band either side, shifted likewise (skew_off).
  // Without slabs the window is the whole grid and both offsets are 0.
  int slab_on, shifted;
  int row_lo, row_hi, win_lo, win_hi;
  size_t win_off, Cw, skew_off, Sw;
  int ab_lo, ab_hi;           // bands the skewed arrays hold
  unsigned int* keys[2];      // slab mode: the GLOBAL array index (the reference's position in g_markers) of each local marker
  euler_comm_ops bulk;        // the communicator the caller installed (the mailboxes replace S->comm's all-reduce / halo only)
  struct SlabScratch* slab;   // exchange buffers (k_slab.hip)
  int part_lo[64], part_hi[64];   // every rank's band range (eu_slab_check_partition), for the snapshot manifest and the render gather
  void* rccl;                 // the built-in RCCL communicator (comm_rccl.hip), if euler_set_comm_rccl installed one
  void* p2p;                  // peer-to-peer mailboxes for the scalar all-reduces and ghost rows (comm_p2p.hip)
  int p2p_on;                 // connected: reductions finish their all-reduce in their own last block, ghost rows go direct

  // profiling: hipEvent pairs per launch; PCG launches carry (solve, iteration) so that launches
  // that returned at once (after convergence / all-zero rhs) are NOT counted
  uint64_t prof_mask;
  uint64_t prof_open;      // classes whose current launch carries a begin event (EULER_OPT_PROFILE_STRIDE: not every launch does)
  unsigned int prof_seq[32];      // launches of each enabled class seen so far
  double prof_ms[KC__COUNT];
  uint64_t prof_launches[KC__COUNT];
  uint64_t prof_idle[KC__COUNT];
  hipEvent_t* ev_pool; int* ev_cls; int* ev_solve; int* ev_iter; int ev_used, ev_cap;
  int solve_seq;          // index of the solve being enqueued
  int prof_iter;          // iteration tag for the next launches: -2 = not a PCG launch, -1 = solve prologue
  int solve_iters[256];   // final iteration count per finished solve (ring), -1 = rhs was all zero
};

// ------------------------------------------------------------------------------------------
void eu_set_error(const char* fmt, ...);
int eu_hip_fail(hipError_t e, const char* what, const char* file, int line);
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return eu_hip_fail(_e, #call, __FILE__, __LINE__); } while (0)

void eu_prof_begin(euler_sim* S, int cls);
void eu_prof_end(euler_sim* S, int cls);
int  eu_prof_flush(euler_sim* S);

#define LAUNCH(S, CLS, KERNEL, GRID, BLOCK, ...)                                   \
  do {                                                                             \
    eu_prof_begin((S), (CLS));                                                     \
    hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, (S)->stream, __VA_ARGS__);          \
    eu_prof_end((S), (CLS));                                                       \
  } while (0)

static inline unsigned eu_blocks(size_t n, unsigned per_block, unsigned cap = 0x7fffffffu) {
  size_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Kernels whose neighbouring blocks touch the same
// cache lines of a row-major field (markers lie in the array roughly in spatial order; the solver's diagonal order crosses a
// row's line 16 times) give each XCD one CONTIGUOUS eighth of the block range instead: a bijection of [0, gridDim.x).
__device__ __forceinline__ size_t eu_xcd_block() {
  const unsigned int b = blockIdx.x, full = gridDim.x & ~7u;
  return b < full ? (size_t)(b & 7) * (full >> 3) + (b >> 3) : (size_t)b;
}

#define EU_GHOST_LO 1   // ghost rows below / above a slab: u, v need 1 / 1, the count grids 1 / 2, vtmp 1 / 0
#define EU_GHOST_HI 2
// slab mode (k_slab.hip)
int  eu_slab_alloc(euler_sim* S);
size_t eu_slab_bytes(const euler_sim* S);   // device memory of the slab exchange buffers
void eu_slab_release(euler_sim* S);
int  eu_slab_substep(euler_sim* S, float dt);
int  eu_slab_stage(euler_sim* S, int stage, float dt);   // euler_stage on a row-slab handle (collective): the stage with its exchanges
int  eu_slab_timestep(euler_sim* S, float frame_time_left);
int  eu_slab_after_load(euler_sim* S);
int  eu_slab_check_partition(euler_sim* S);   // collective: the ranks' band ranges tile the grid
int  eu_slab_exchange_uv(euler_sim* S);     // ghost rows of u, v (euler_set_field on a slab handle)
int  eu_slab_exchange_dye(euler_sim* S);    // ghost rows of g_r, g_g, g_b (--rainbow)
int  eu_slab_render(euler_sim* S, int wx, int wy, char* out, int cap, int* len);   // snapshot.hip: euler_render on a row-slab handle (collective)
int  eu_slab_after_restore(euler_sim* S);   // collective: the job-wide facts a scenario load sets up (source cells of all ranks)
int  eu_slab_error_sync(euler_sim* S);      // collective: the ranks' sticky error words -> their maximum on every rank
int  eu_slab_same_everywhere(euler_sim* S, const double* vals, int n, int* same);   // collective: do these host-side values agree on every rank?
int  eu_slab_status_sync(euler_sim* S, int local_rc, int* worst);   // collective: a host-side status code -> non-zero on every rank if any rank failed
// the grids of the handle were (re)written whole - allocation, a scenario load, a snapshot: nothing one stage prepared for the next describes them any more
static inline void eu_state_replaced(euler_sim* S) {
  S->blocked_dirty = 1; S->solidT_dirty = 1;      // (the column-major copies of the solid / sink grids)
  S->prebin_valid = 0;
  if (S->maxsq_state == 2) S->maxsq_state = 1;
  S->p_pending = 0;
  S->uv_clean = 0; S->uv_zb = 0;
  S->tmap_valid = 0; S->utmp_clean = 0; S->countT_clean = 0;
}
// the tile map describes both count grids and this handle's passes may use it (EULER_OPT_NO_TILE_MAP: rounds 1-5's forms)
static inline bool eu_tile_map_on(const euler_sim* S) { return S->tmap && S->tmap_valid && !S->slab_on && S->opt[EULER_OPT_NO_TILE_MAP] == 0; }
// launch groups implemented in the kernel files
int eu_launch_timestep(euler_sim* S, float frame_time_left);
int eu_launch_advect_markers(euler_sim* S, float dt);
int eu_launch_refresh_counts(euler_sim* S);
int eu_launch_sources(euler_sim* S);
int eu_launch_extrapolate(euler_sim* S);
int eu_launch_advect_velocity(euler_sim* S, float dt);
int eu_launch_diffuse(euler_sim* S, float dt);     // the diffusion extension alone (row slabs: behind the ghost rows of utmp / vtmp)
int eu_launch_project(euler_sim* S, float dt);
int eu_resident_capacity(euler_sim* S, int f32);            // k_resident.hip: workgroups of the resident PCG kernel the device holds at once
int eu_launch_resident(euler_sim* S, unsigned int n_chunks);
bool eu_resident_eligible(const euler_sim* S);               // driver.hip: this handle's configuration and grid allow the resident solver
int eu_launch_colorize(euler_sim* S);            // k_dye.hip: no-ops without cfg.rainbow
int eu_launch_dye_extrapolate(euler_sim* S);
int eu_launch_dye_sources(euler_sim* S);
int eu_launch_dye_advect(euler_sim* S, float dt);
int eu_launch_pcg_op(euler_sim* S, int op, float dt, double a, double* out);
int eu_ordered_select(euler_sim* S, const unsigned long long* mask, size_t nwords, unsigned int* out_idx,
                      unsigned int* out_total);
int eu_sync_marker_state(euler_sim* S);
void eu_rccl_release(euler_sim* S);   // comm_rccl.hip
void eu_p2p_release(euler_sim* S);    // comm_p2p.hip
int eu_p2p_halo_skewed(euler_sim* S, double* s_skewed);   // ghost rows of a band-skewed vector, straight from / into the array
// hand-off rows of the band pipeline across slabs (exact coupling): where this rank's first band reads, where its last band writes
void eu_p2p_xgran(euler_sim* S, int backward, const unsigned long long** in, unsigned long long** out);
// the neighbouring ranks' z and (current) s arrays, IPC-mapped; null where there is no such rank / nothing is mapped
int  eu_p2p_has_neighbour_arrays(const euler_sim* S);
void eu_p2p_neighbour_arrays(euler_sim* S, const double** z_dn, const double** s_dn, const double** z_up, const double** s_up);
int eu_install_comm(euler_sim* S, const euler_comm_ops* ops, int32_t coupling, int allow_single);
static inline bool eu_is_tile(const euler_sim* S) { return S->cfg.precond == EULER_PRECOND_IC0_TILE || S->cfg.precond == EULER_PRECOND_IC0_TILE2 || S->cfg.precond == EULER_PRECOND_IC0_TILE_MG; }
static inline bool eu_is_two_level(const euler_sim* S) { return S->cfg.precond == EULER_PRECOND_IC0_TILE2 || S->cfg.precond == EULER_PRECOND_IC0_TILE_MG; }   // "has a coarse correction"
static inline bool eu_is_mg(const euler_sim* S) { return S->cfg.precond == EULER_PRECOND_IC0_TILE_MG; }
// two-level preconditioner (k_coarse.hip)
int  eu_coarse_alloc(euler_sim* S, bool mg);  // lazily, when the mode is first selected (mg: the multilevel mode's hierarchy as well)
void eu_coarse_release(euler_sim* S);
int  eu_launch_coarse_setup(euler_sim* S);    // per solve: P^T A P, its factor and inverse
int  eu_launch_coarse_consistent(euler_sim* S);   // per solve: if water is cut off from the air, take the part of r = b along that region's indicator out (host sync for the count)
int  eu_launch_coarse_solve(euler_sim* S, int fin_op, int force);   // per iteration: y = (P^T A P)^-1 P^T r, dot(z,r) += y . r_c, the scalar epilogue
int  eu_launch_coarse_search_init(euler_sim* S);   // s = z + P y (the first search direction of a solve)
int  eu_coarse_comm_slots(euler_sim* S);      // row slabs: doubles per rank in S->mg_xbuf (allocated on demand), < 0 on error
int  eu_launch_coarse_pre(euler_sim* S, int force);   // row slabs: this rank's share of the level-0 right-hand side into its slot, before the exchange
int  eu_launch_coarse_scatter(euler_sim* S);          // row slabs: behind the exchange, the ranks' shares added up: the level-0 right-hand side whole on every rank
int  eu_launch_coarse_add_row(euler_sim* S, double* row, int yrow);   // row slabs: + P y on a compact ghost row (grid row yrow)

// ------------------------------------------------------------------------------------------
// device helpers
#ifdef __HIPCC__

#define EU_WAVE 64

struct GridRef {
  int X, Y;
  const uint8_t* count;   // g_fluid alias (main.c:99)
  float ux_lim, uy_lim, vx_lim, vy_lim;
};

// typed cell property (p_property/u_property/v_property, main.c:119-138)
__device__ __forceinline__ bool eu_prop_p(const uint8_t* g, size_t i) { return g[i] != 0; }
// the tile map (euler_sim.tmap): no water in the tiles (ty, tx0 .. tx0 + ntx - 1), the tile to their right and the row of tiles above them - in the count grid and in the
// previous one.  Uniform over a workgroup when its arguments are (scalar loads).  Column tnx - 1 and the rows behind the grid's last tile row are a ring of zeros.
__device__ __forceinline__ bool eu_tiles_idle(const uint8_t* __restrict__ tmap, int tnx, int tn, int ty, int tx0, int ntx) {
  unsigned int a = 0;
  for (int k = 0; k <= ntx; ++k) {
    const int tx = tx0 + k < tnx ? tx0 + k : tnx - 1;
    const int t = ty * tnx + tx;
    a |= (unsigned int)tmap[t] | tmap[t + tnx] | tmap[tn + t] | tmap[tn + t + tnx];
  }
  return a == 0;
}
// ... in the count grid as it stands only, the tile columns tx0 .. tx1 and one more to the right, the tile row ty and (up) the one above
__device__ __forceinline__ bool eu_tiles_idle_cur(const uint8_t* __restrict__ tmap, int tnx, int ty, int tx0, int tx1, bool up) {
  unsigned int a = 0;
  for (int tx = tx0; tx <= tx1 + 1; ++tx) {
    const int t = ty * tnx + (tx < tnx ? tx : tnx - 1);
    a |= (unsigned int)tmap[t] | (up ? (unsigned int)tmap[t + tnx] : 0u);
  }
  return a == 0;
}
__device__ __forceinline__ bool eu_prop_u(const uint8_t* g, size_t i) { return (g[i] != 0) | (g[i + 1] != 0); }
__device__ __forceinline__ bool eu_prop_v(const uint8_t* g, size_t i, int X) { return (g[i] != 0) | (g[i + X] != 0); }

__device__ __forceinline__ float eu_lerp(float x0, float x1, float f) { return (1.f - f) * x0 + f * x1; }   // main.c:311-313
__device__ __forceinline__ float eu_frac(float f, bool start_ok, bool end_ok) {                            // main.c:301-309
  return !start_ok ? 1.f : (!end_ok ? 0.f : f);
}
__device__ __forceinline__ float eu_clampf(float lo, float x, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// interpolate(), main.c:337-364.  TYPE 0 = cell centres (P), 1 = U samples, 2 = V samples.
// TR: q and g.count are stored COLUMN-major, [x][y] (the marker stage's copies: k_markers.hip) - the same values, the strides swapped
// PROPS: g.count does not hold counts but, per cell, the typed fluid properties of the four corners of an interpolation whose BASE cell it is (bits 0-3 U-typed, 4-7 V-typed,
// corner order v00 v01 v10 v11; k_transpose_for_markers): one byte gather instead of six
template <int TYPE, bool TR = false, bool PROPS = TR>
__device__ __forceinline__ float eu_interp(const GridRef& g, const float* __restrict__ q, float ix, float iy) {
  ix = eu_clampf(0.f, ix, TYPE == 1 ? g.ux_lim : g.vx_lim);   // P extent = (X, Y): x like V, y like U
  iy = eu_clampf(0.f, iy, TYPE == 2 ? g.vy_lim : g.uy_lim);
  float wx, wy;
  const float fx = modff(ix, &wx), fy = modff(iy, &wy);
  const int bx = (int)wx, by = (int)wy;
  const size_t sx = TR ? (size_t)g.Y : (size_t)1, sy = TR ? (size_t)1 : (size_t)g.X;      // one step in x / in y
  const size_t i00 = (size_t)by * sy + (size_t)bx * sx;
  // the four samples are loaded unconditionally and next to the mask bytes (the clamps keep all of them inside the arrays,
  // ghost rows included): one memory round trip instead of two - the kernels that interpolate are latency-bound
  const float r00 = q[i00], r01 = q[i00 + sx], r10 = q[i00 + sy], r11 = q[i00 + sy + sx];
  bool v00, v01, v10, v11;
  if (PROPS && TYPE != 0) {
    // ONE byte instead of six counts - the passes that interpolate are bound by their gather instructions and the lines each touches
    const unsigned int nb = g.count[i00] >> (TYPE == 1 ? 0 : 4);
    v00 = (nb & 1u) != 0; v01 = (nb & 2u) != 0; v10 = (nb & 4u) != 0; v11 = (nb & 8u) != 0;
  } else
  if (TYPE == 0) {
    v00 = g.count[i00] != 0; v01 = g.count[i00 + sx] != 0;
    v10 = g.count[i00 + sy] != 0; v11 = g.count[i00 + sy + sx] != 0;
  } else if (TYPE == 1) {
    const bool c0 = g.count[i00] != 0, c1 = g.count[i00 + sx] != 0, c2 = g.count[i00 + 2 * sx] != 0;
    const bool d0 = g.count[i00 + sy] != 0, d1 = g.count[i00 + sy + sx] != 0, d2 = g.count[i00 + sy + 2 * sx] != 0;
    v00 = c0 | c1; v01 = c1 | c2; v10 = d0 | d1; v11 = d1 | d2;
  } else {
    const bool c0 = g.count[i00] != 0, c1 = g.count[i00 + sx] != 0;
    const bool d0 = g.count[i00 + sy] != 0, d1 = g.count[i00 + sy + sx] != 0;
    const bool e0 = g.count[i00 + 2 * sy] != 0, e1 = g.count[i00 + 2 * sy + sx] != 0;
    v00 = c0 | d0; v01 = c1 | d1; v10 = d0 | e0; v11 = d1 | e1;
  }
  const float q00 = v00 ? r00 : 0.f, q01 = v01 ? r01 : 0.f;
  const float q10 = v10 ? r10 : 0.f, q11 = v11 ? r11 : 0.f;
  const float lf = eu_frac(fy, v00, v10), rf = eu_frac(fy, v01, v11);
  const float lv = eu_lerp(q00, q10, lf), rv = eu_lerp(q01, q11, rf);
  const float hf = eu_frac(fx, v00 | v10, v01 | v11);
  return eu_lerp(lv, rv, hf);
}

// xorshift64* (misc/rng.c:5-20) jump-ahead.  The state update x ^= x >> 12; x ^= x << 25; x ^= x >> 27 is linear over GF(2):
// one step is a 64 x 64 bit matrix M, k steps are M^k.  jump[i][b] = M^(2^i) e_b (the image of bit b), filled once on the
// host (driver.hip); a state is advanced k steps by applying M^(2^i) for the set bits of k - 64 conditional XORs each.  This
// is what lets thousands of threads draw the substep's source positions (main.c:288) from ONE sequential stream.
#define EU_RNG_JUMPS 40
struct RngJump { unsigned long long col[EU_RNG_JUMPS][64]; };
__device__ __forceinline__ unsigned long long eu_rng_step(unsigned long long st) {
  st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
  return st;
}
__device__ __forceinline__ unsigned long long eu_rng_jump(const RngJump* __restrict__ J, unsigned long long st, unsigned long long k) {
  for (int i = 0; k && i < EU_RNG_JUMPS; ++i, k >>= 1) {
    if (!(k & 1)) continue;
    unsigned long long y = 0, x = st;
    const unsigned long long* c = J->col[i];
    while (x) { const int b = __ffsll((long long)x) - 1; y ^= c[b]; x &= x - 1; }
    st = y;
  }
  return st;
}
__device__ __forceinline__ float eu_rng_float(unsigned long long st) {   // randf(), main.c:203-207: closed [0, 1]
  const unsigned int hi = (unsigned int)((st * 0x2545F4914F6CDD1Dull) >> 32);
  return (float)(hi / (double)4294967295u);
}

// wave64 reductions by shuffles
__device__ __forceinline__ double eu_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double eu_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double w = __shfl_down(v, o, 64); v = w > v ? w : v; }
  return v;
}
__device__ __forceinline__ float eu_wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { float w = __shfl_down(v, o, 64); v = w > v ? w : v; }
  return v;
}

// Markers are seeded four to a cell, consecutively (main.c:255-266), and mostly stay neighbours in the array: a lane that
// opens a run of lanes binning into the same cell adds the run's length with ONE atomic (up to 4x fewer atomics, each a
// 32-byte memory-side transaction in a column-ordered array: k_bin_markers at 8192^2: 5.1 ms with one atomic per marker).
__device__ __forceinline__ void bin_aggregated(unsigned int* count32, bool live, size_t c) {
  // (the adds are executed at the memory side, ~11 G/s whatever their width: at 8192^2 they, not the marker reads, set the
  // binning kernel's time - 3.4 ms against 0.5 ms without them.)  One add per RUN of lanes binning into the same cell - and, since the
  // counters are column-major like the seeding order (main.c:243-266: consecutive runs are consecutive cells of one column), ONE 64-bit add
  // for two runs whose cells are an even / odd pair of neighbouring counters: the low word takes the even cell's run, the high word the odd
  // cell's (no carry: a counter stays far below 2^32).
  const unsigned int lo = (unsigned int)c, hi = (unsigned int)(c >> 32);
  const int lane = threadIdx.x & 63;
  const unsigned int plo = __shfl_up(lo, 1, 64), phi = __shfl_up(hi, 1, 64);
  const bool plive = __shfl_up((int)live, 1, 64) != 0;
  const bool head = live && (lane == 0 || !plive || plo != lo || phi != hi);     // first lane of a run of equal cells
  const unsigned long long heads = __ballot(head), lives = __ballot(live);
  // the run ends before the next head or the next dead lane
  const unsigned long long above = lane == 63 ? 0ull : ((heads | ~lives) >> (lane + 1));
  const int run = above ? __ffsll((long long)above) : 64 - lane;
  // the run behind this one: starts at lane + run if that lane is a head (i.e. no dead lane in between)
  const int nl = lane + run;
  const bool has_next = head && nl < 64 && ((heads >> nl) & 1ull);
  const int src = has_next ? nl : lane;
  const unsigned int nlo = __shfl(lo, src, 64), nhi = __shfl(hi, src, 64);
  const int nrun = __shfl(run, src, 64);
  const bool pairs = has_next && !(lo & 1u) && nlo == lo + 1u && nhi == hi;      // (c even: the 64-bit add is aligned)
  // is this head the second of such a pair?  The head below it is the highest head bit under this lane
  const unsigned long long below = heads & ((1ull << lane) - 1ull);
  const int pl = below ? 63 - __clzll((long long)below) : lane;
  const unsigned int qlo = __shfl(lo, pl, 64), qhi = __shfl(hi, pl, 64);
  const int qrun = __shfl(run, pl, 64);
  const bool absorbed = head && below && (lo & 1u) && qlo + 1u == lo && qhi == hi && pl + qrun == lane;
  if (head && !absorbed) {
    if (pairs) atomicAdd(reinterpret_cast<unsigned long long*>(&count32[c]), (unsigned long long)(unsigned int)run | ((unsigned long long)(unsigned int)nrun << 32));
    else atomicAdd(&count32[c], (unsigned int)run);
  }
}


// ---- peer-to-peer mailboxes (comm_p2p.hip): self-validating 16-byte granules {lo32, tag, hi32, tag}, one
// system-scope write-through store / system-scope load each - the band pipeline's hand-off form across GPUs
#define P2P_MAXR 16
#define P2P_SPIN_LIMIT (1u << 25)   // x ~1.5 us per poll: a peer may lag by many seconds (start-up skew of a job), not for ever
#define P2P_HDR_BYTES 4096
typedef unsigned int p2p_u32x4 __attribute__((ext_vector_type(4)));
struct P2PBoxHeader {
  p2p_u32x4 scalar[2][P2P_MAXR];     // [parity][sending rank]
};
__device__ __forceinline__ void p2p_store(void* p, double v, unsigned int tag) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  const p2p_u32x4 g = {(unsigned int)bits, tag, (unsigned int)(bits >> 32), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ bool p2p_poll(const void* p, unsigned int tag, double* v) {
  p2p_u32x4 g;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory");
  *v = __hiloint2double((int)g[2], (int)g[0]);
  return g[1] == tag && g[3] == tag;
}
// All-reduce of one double across the ranks, called by EVERY thread of a block of >= P2P_MAXR threads; v is taken from
// thread 0 and the result is valid in thread 0.  Lane j writes this rank's value into rank j's mailbox and polls the
// own mailbox for rank j's; the fold runs in rank order, so every rank gets the same bits.
template <bool IS_MAX>
__device__ __forceinline__ double p2p_allreduce_block(PcgScalars* sc, double v) {
  __shared__ double s_got[P2P_MAXR];
  __shared__ double s_v;
  __shared__ unsigned int s_seq;
  if (threadIdx.x == 0) { s_v = v; s_seq = ++sc->p2p_seq; }
  __syncthreads();
  const int j = threadIdx.x, n = sc->p2p_n, rank = sc->p2p_rank;
  if (j < n) {
    const unsigned int seq = s_seq;
    const int par = seq & 1;
    const double mine = s_v;
    p2p_store(&static_cast<P2PBoxHeader*>(sc->p2p_boxes[j])->scalar[par][rank], mine, seq);
    const P2PBoxHeader* own = static_cast<const P2PBoxHeader*>(sc->p2p_boxes[rank]);
    double got = 0.0;
    unsigned int spins = 0;
    while (!p2p_poll(&own->scalar[par][j], seq, &got)) {
      if (++spins > P2P_SPIN_LIMIT) { atomicExch(sc->p2p_error, 3); got = mine; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    s_got[j] = got;
  }
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    t = s_got[0];
    for (int k = 1; k < n; ++k) t = IS_MAX ? (s_got[k] > t ? s_got[k] : t) : t + s_got[k];
  }
  return t;
}

#endif  // __HIPCC__
