/*
 * euler.h — C ABI of libeuler_hip.so: the MI355X-native replacement for the simulation path of
 * cgmb/euler (reference main.c: sim_init :209, sim_step :843, draw :953 and the file-scope
 * arrays g_u/g_v/g_marker_count/... :64-100 they communicate through).
 *
 * Plain C, plain pointers and sizes; no HIP or torch types.  Host code (the `euler` CLI, the
 * reference's own main loop, bench.py, pytest via ctypes) sees only this file.
 *
 * Every function returns EULER_OK (0) or a negative EULER_E* code and never calls exit()
 * (reference behaviour replaced: main.c:212-215 fprintf+exit(1)).  euler_last_error() returns
 * a thread-local human-readable message for the last failure.
 *
 * A handle is single-caller (not re-entrant), like the reference's globals.
 */
#ifndef EULER_H
#define EULER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EULER_ABI_VERSION 2   /* 2: euler_comm_ops.exchange, row-slab snapshots */

enum {
  EULER_OK = 0,
  EULER_EINVAL = -1,     /* bad argument */
  EULER_ENOMEM = -2,     /* host or device allocation failed */
  EULER_EIO = -3,        /* scenario file unreadable (reference: "Could not load %s!", main.c:213) */
  EULER_EHIP = -4,       /* a HIP runtime call failed / no gfx950 device */
  EULER_ESTATE = -5,     /* call order violated (e.g. step before a scenario is loaded) */
  EULER_ETIMEOUT = -6,   /* an in-kernel bounded wait expired (band pipeline) */
  EULER_ECOMM = -7       /* multi-GPU communicator failure */
};

/* How dot(a,b) (reference main.c:629-639, a sequential row-major double sum) is evaluated. */
enum {
  EULER_DOT_AUTO = 0,        /* SEQUENTIAL when X*Y <= 65536, else TREE */
  EULER_DOT_SEQUENTIAL = 1,  /* one thread adds in the reference's order: bit-identical results */
  EULER_DOT_TREE = 2         /* fixed-shape parallel reduction: deterministic, differs in the last ulps */
};

/* Preconditioner of the pressure solve. */
enum {
  EULER_PRECOND_IC0 = 0,     /* the reference's incomplete Cholesky (main.c:580-627), evaluated as a
                                dependency-ordered wavefront: bit-identical to the sequential sweep */
  EULER_PRECOND_JACOBI = 1,  /* z = r/diag: NOT the reference's iterates; for roofline comparison only */
  EULER_PRECOND_IC0_TILE = 2, /* tile-local IC(0) (SURVEY 7 hard part 1(b): "tile-local IC(0) = block-Jacobi-IC"): the reference's
                                three recurrences (main.c:586-626) restricted to blocks - a block = the cells of one 64-row band
                                whose skew records t = x + y % 64 fall into one tile of euler_config.precond_tile_records records
                                (a parallelogram of 64 rows x W columns); couplings between blocks are dropped from the factor and
                                from both triangular solves.  A block lives in the registers of one wave, so r -= alpha A s,
                                max |r|, both solves and dot(z,r) are ONE pass over memory (k_precond_tile), bandwidth-bound
                                instead of latency-bound, and nothing couples row slabs.  NOT the reference's iterates: the same
                                solution where PCG converges (tolerance parity), ~25-35 % more iterations (tools/precond_study.py).
                                Restated in the oracle (eo_sim.tile_records): GPU = oracle bit for bit in EULER_DOT_SEQUENTIAL. */
  EULER_PRECOND_IC0_TILE2 = 3,/* TWO-LEVEL (round 3): the tile-local IC(0) above (64 x 16 blocks) plus a coarse correction,
                                    z = M_tile^-1 r + P (P^T A P)^-1 P^T r,
                                P = piecewise constants over coarse cells of (64 m)^2 grid cells restricted to the fluid, m the smallest
                                power of two that leaves at most 256 coarse cells (16 x 16 of them on a square grid).  The block-local
                                factor has no coupling beyond a block; the coarse space restores the long-range part of the inverse, which is
                                what the first hundred iterations of a large solve live on: 2-4x fewer iterations than the REFERENCE's IC(0)
                                to the reference's tolerance (2048^2 tank: 402 vs 1726), the reference's residual-after-100 in ~44 iterations
                                on the saturated 8192^2 tank (docs/solver_two_level.md).  Same traffic
                                per iteration as the tile-local mode (the coarse part is 3 doubles per tile out, one double per coarse cell
                                in) + one small launch.  NOT the reference's iterates; symmetric positive definite, so PCG converges to the same
                                solution.  Restated in the oracle (eo_sim.coarse_m): GPU = oracle to rounding (tolerance, not bits: the
                                coarse sums are formed in another order).  One GPU only; EULER_DOT_TREE. */
  EULER_PRECOND_IC0_TILE_MG = 4 /* MULTILEVEL (round 3; round 5: bilinear node grids): the same, with the coarse correction taken from a hierarchy instead of one level,
                                    z = M_tile^-1 r + P_0 V(P_0^T r),
                                level 0 = a grid of nodes 8 cells apart (node (I, J) at the centre of cell (8 J + 4, 8 I + 4)), P_0 = bilinear interpolation from the four nodes
                                around a cell, restricted to the fluid; every further level = every other node of the one below, bilinear again (full weighting); Galerkin
                                operators (exact nine-point stencils), one symmetric V-cycle (damped Jacobi before and after: 0.8, per node less where a Gershgorin bound of 1.6
                                demands it - spray drops), the top level (<= 64 nodes) solved
                                with the dense pseudo-inverse of the two-level mode.  The number of iterations to the reference's tolerance does not grow with the grid:
                                ~30 on a tank at rest at any size, 40-50 on moving water, where the reference's IC(0) needs 231 (256^2), 880 (1024^2), thousands (8192^2)
                                and rounds 3-4's piecewise-constant aggregates of 16 cells needed 105-150.  Cost per iteration: the tile-local mode's two passes (+ 72
                                doubles of partial sums per 1024-cell tile) + five launches over node grids 1/64 the size of the grid and less.  Restated in the oracle
                                (eo_sim.coarse_mg): GPU = oracle to rounding.  EULER_DOT_TREE.  One GPU, or row slabs without mailboxes (euler_config.slab_*).  On slabs the cycle is SPLIT BY ROWS
                                wherever a level of <= 16384 nodes lies above level 0 and the slabs are a few bands thick (docs/solver_multilevel.md; EULER_OPT_MG_SPLIT_LEVEL): a rank takes
                                its own tiles' share down to that level, windows of it are all-gathered inside the G1 exchange (0.16 MB per iteration at 16384^2 on 8 ranks), the
                                neighbours' shares on a rank's halo rows travel with the edge rows of z, the coarse levels run replicated, the fine levels on the own node rows, and
                                one more 40-byte exchange sums the correction's share of dot(z, r); per solve the operators are formed by their owners.  Otherwise (small grids,
                                thin slabs, EULER_OPT_MG_SPLIT_LEVEL = -1) the cycle runs replicated: one all-reduce per solve makes the level-0 operator whole, every rank
                                contributes its node rows to the G1 all-gather (cells / 64 doubles over all ranks: 32 MB at 16384^2).  Either way every rank computes the same bits.  Water cut off from the air (a singular
                                system): the right-hand side is made compatible with the region's indicator, and the correction is kept mean-free over the region so that
                                the pressure's constant - which the reference's clamp p >= 0 makes observable - is the tile-local factor's, i.e. very nearly the reference's. */
};

/* IC(0) sweep implementation (same arithmetic, different schedule). */
enum {
  EULER_SWEEP_AUTO = 0,
  EULER_SWEEP_BAND = 1,      /* one wave per 64-row band streaming band-skewed records, bands pipelined
                                through tagged granules (DESIGN.md "IC(0) sweeps") */
  EULER_SWEEP_SIMPLE = 2     /* one workgroup, one barrier per anti-diagonal (debug / cross-check) */
};

typedef struct euler_config {
  int32_t abi_version;     /* EULER_ABI_VERSION */
  int32_t X, Y;            /* grid size; reference: compile-time enum X=100, Y=40 (main.c:22-25) */
  int32_t device;          /* HIP device ordinal */
  int32_t max_iterations;  /* PCG cap, reference 100 (main.c:735) */
  double  tol;             /* inf-norm tolerance, reference (double)1e-6f (main.c:736) */
  int32_t dot_mode;        /* EULER_DOT_* */
  int32_t precond;         /* EULER_PRECOND_* */
  int32_t sweep_mode;      /* EULER_SWEEP_* */
  int32_t max_substeps;    /* reference 8 (main.c:851) */
  float   frame_time;      /* reference 0.1f (main.c:849) */
  float   viscosity;       /* 0 = inviscid like the reference (no diffusion stage exists there) */
  int32_t pcg_poll_interval; /* PCG iterations launched between convergence polls (default 8) */
  int32_t rainbow;         /* args_t.rainbow / g_rainbow_enabled (main.c:54,75,1020): carry the dye fields */
  int32_t precond_tile_records; /* EULER_PRECOND_IC0_TILE: records per tile, 8, 16 or 32 (0 = default 16); tile k of a band =
                                   records [k * W, (k + 1) * W) */
  int32_t slab_rank, slab_nranks; /* slab_nranks >= 1 (0 = off): ROW SLABS FOR EVERY STAGE.  This handle is rank slab_rank of a job of
                                   slab_nranks processes, one per GPU; it owns the rows of its 64-row bands (euler_slab_info) and
                                   allocates ONLY those rows (+ 1 ghost row below, 2 above) of every grid and only the markers inside
                                   them - per-rank memory ~ 1/slab_nranks.  Install a communicator of the same rank / size
                                   (euler_set_comm / euler_set_comm_rccl) BEFORE loading a scenario.  Fields are then exchanged as the
                                   owned rows (euler_get_field), markers as the local ones with their global array index
                                   (EULER_F_MARKER_KEYS); euler_set_field(U / V) takes the own rows and is COLLECTIVE (it refreshes the
                                   neighbours' ghost rows).  Needs EULER_PRECOND_IC0_TILE or slab-local IC(0) coupling. */
  int32_t slab_band_lo, slab_band_hi; /* row slabs: this rank's 64-row bands [lo, hi) given explicitly (hi > lo) instead of the even split
                                   nbands * rank / nranks - for partitions that balance the FLUID (a dam break settles into the
                                   lowest third of the tank: even row slabs leave most ranks with air).  The ranks' ranges must tile
                                   [0, nbands) in rank order; euler_set_comm* checks it against the neighbours.  0, 0 = even split. */
  int32_t pcg_precision;   /* EULER_PCG_F64 (0, default: the reference's double vectors, main.c:577-578,716) or EULER_PCG_F32: every solver vector in float, sums and
                              scalars in double - BASELINE configs[1]'s "fp32".  NOT the reference's iterates (tolerance parity only, restated in the oracle:
                              eo_sim.pcg_f32); runs in the resident solver, so it needs what that needs (below) and euler_create refuses it otherwise. */
  int32_t resident;        /* EULER_RESIDENT_AUTO (0): a solve whose ACTIVE 16-record chunks (those holding fluid) all find a wave on the chip at once - at most
                              4 x (resident workgroups): 2048 chunks = 2 M cells of water in double, 3072 in float on an MI355X; BASELINE configs[1] (1024^2 dam break)
                              always, configs[4] (4096^2 waterfall) while its water is below that - runs the tile-local PCG (EULER_PRECOND_IC0_TILE, one GPU, EULER_DOT_TREE,
                              tiles of 16 records) as ONE persistent launch whose vectors stay in registers (csrc/k_resident.hip): same arithmetic, sums folded per
                              workgroup, so the iterates agree with the multi-kernel form to rounding; decided per solve.  EULER_RESIDENT_OFF (1): always the
                              multi-kernel form. */
} euler_config;

enum { EULER_PCG_F64 = 0, EULER_PCG_F32 = 1 };
enum { EULER_RESIDENT_AUTO = 0, EULER_RESIDENT_OFF = 1 };

typedef struct euler_sim euler_sim; /* opaque */

/* Arrays readable/writable through euler_get_field/euler_set_field.  All grids are row-major
 * [Y][X], full size even for the staggered U/V samples (main.c:62-67). */
enum {
  EULER_F_U = 0,          /* float  g_u     main.c:64 */
  EULER_F_V,              /* float  g_v     main.c:65 */
  EULER_F_UTMP,           /* float  g_utmp  main.c:66 */
  EULER_F_VTMP,           /* float  g_vtmp  main.c:67 */
  EULER_F_SOLID,          /* uint8  g_solid main.c:71 */
  EULER_F_SOURCE,         /* uint8  g_source main.c:72 */
  EULER_F_SINK,           /* uint8  g_sink  main.c:73 */
  EULER_F_COUNT,          /* uint8  g_marker_count main.c:96 */
  EULER_F_PREV_COUNT,     /* uint8  g_prev_marker_count main.c:97 */
  EULER_F_MARKERS,        /* float2[n_markers] g_markers main.c:95, in the reference's array order */
  EULER_F_PRECON,         /* double g_precon main.c:577 (persistent state) */
  EULER_F_PRESSURE,       /* double p — a stack local of project(), main.c:739; exposed here.  Whole-grid handles form the finished, clamped pressure in device memory only when
                             it is asked for (the velocity update keeps it in LDS): the call costs one pass over the solver arrays the first time after a substep */
  EULER_F_PCG_B, EULER_F_PCG_R, EULER_F_PCG_Z, EULER_F_PCG_S, EULER_F_PCG_Q, /* double, main.c:716-745,578.  Test surface: after a solve S is the search
                                                              direction of the last iteration that ran (+0 off the fluid and when the right-hand side was
                                                              all zero); Q holds A s only where a solve stores it (docs/solver_tile_local.md: most do not any more).
                                                              After a solve that ran in the resident kernel (euler_resident_info) Z, S and Q do not exist in
                                                              memory: euler_get_field returns EULER_ESTATE for them until a multi-kernel solve has run */
  EULER_F_CELLMASK,       /* uint8: bit0 fluid, bit1..4 fluid at x+1,y+1,x-1,y-1, bits5-7 a_diag (g_a, main.c:552) */
  EULER_F_DYE_R, EULER_F_DYE_G, EULER_F_DYE_B,             /* float g_r, g_g, g_b (main.c:76-78); euler_config.rainbow only */
  EULER_F_DYE_RTMP, EULER_F_DYE_GTMP, EULER_F_DYE_BTMP,    /* float g_rtmp, g_gtmp, g_btmp (main.c:79-81): state, because the
                                                              reference copies them back whole (main.c:875-881) */
  EULER_F_MARKER_KEYS,    /* uint32[n local markers]: row-slab handles only - the position of each local marker in the reference's
                             g_markers array (EULER_F_MARKERS of all ranks, ordered by key, IS that array) */
  EULER_F__COUNT
};

/* Stages of one substep, in the reference's order (main.c:855-893).  Fused stages are listed
 * as the reference stages they cover; the arrays they leave behind equal the reference's after
 * the LAST covered stage. */
enum {
  EULER_STAGE_ADVECT_MARKERS = 0,   /* advect_markers            main.c:464-537 */
  EULER_STAGE_REFRESH_COUNTS,       /* refresh_marker_counts     main.c:102-117 */
  EULER_STAGE_SOURCES,              /* update_fluid_sources      main.c:276-298 */
  EULER_STAGE_EXTRAPOLATE,          /* extrapolate(U), extrapolate(V), zero_bounds(U), zero_bounds(V)  main.c:865-868 */
  EULER_STAGE_ADVECT_VELOCITY,      /* advect_u, advect_v, apply_body_forces, zero_bounds x2          main.c:871-889 */
  EULER_STAGE_PROJECT,              /* project                   main.c:709-806 */
  EULER_STAGE__COUNT
};

/* Single PCG building blocks on the handle's device vectors (kernel-level parity tests and
 * micro-benchmarks).  Operands are the EULER_F_PCG_* arrays. */
enum {
  EULER_OP_BUILD_SYSTEM = 0, /* b, r=b, p=0, cell mask from UTMP/VTMP and dt      main.c:713-741 */
  EULER_OP_PRECON_FACTOR,    /* E^-1 into PRECON                                  main.c:586-600 */
  EULER_OP_FORWARD_SOLVE,    /* Q = L^-1 R                                        main.c:602-613 */
  EULER_OP_BACKWARD_SOLVE,   /* Z = L^-T Q                                        main.c:615-626 */
  EULER_OP_APPLY_A,          /* Z = A S                                           main.c:679-691 */
  EULER_OP_DOT_ZR,           /* scalar <- dot(Z,R)                                main.c:629-639 */
  EULER_OP_DOT_ZS,           /* scalar <- dot(Z,S) */
  EULER_OP_INF_NORM_R,       /* scalar <- max |R|                                 main.c:654-667 */
  EULER_OP_UPDATE_PR,        /* P += a S; R -= a Z (a = scalar argument)          main.c:753-754 */
  EULER_OP_UPDATE_SEARCH,    /* S = Z + a S                                       main.c:669-677 */
  EULER_OP__COUNT
};

typedef struct euler_stats {
  uint64_t frames;              /* euler_step calls completed */
  uint64_t total_substeps;
  uint64_t total_pcg_iterations;
  int32_t  last_substeps;       /* substeps of the last frame (1..8) */
  int32_t  last_pcg_iterations; /* PCG iterations summed over the last frame */
  double   last_residual;       /* inf-norm of r when the last solve stopped */
  float    last_dt;             /* dt of the last substep */
  uint64_t n_markers;
  int32_t  source_exhausted;    /* g_source_exhausted, main.c:94 */
  uint64_t rng_state;           /* xorshift64* state (function-static in the reference, main.c:204) */
  uint64_t marker_dt_events;    /* collisions that shortened dt for later markers (main.c:501,518) */
  uint64_t marker_multi_events; /* markers with more than one such collision (expected 0 under the CFL bound) */
  uint64_t fluid_cells;
} euler_stats;

/* ---- life cycle ------------------------------------------------------------------------- */
int  euler_config_default(euler_config* cfg);              /* reference constants, X=100, Y=40 */
int  euler_create(const euler_config* cfg, euler_sim** out);   /* allocates HBM; needs a gfx950 GPU */
void euler_destroy(euler_sim* sim);
const char* euler_last_error(void);
int  euler_abi_version(void);
int  euler_resident_info(euler_sim* sim, uint64_t out[3]); /* {this handle's solves may use the resident solver (euler_config.resident), solves it ran, solves that fell back to the multi-kernel path} */

/* ---- scenario = sim_init (main.c:209-274) ----------------------------------------------- */
/* upscale = 0: the reference's streaming parser at native resolution.
 * upscale = 1: nearest-neighbour resample of the text onto the interior (this build's extension;
 *              DESIGN.md "Scenario upscaling"). */
int euler_load_scenario_mem(euler_sim* sim, const char* text, int32_t len, int32_t upscale);
int euler_load_scenario_file(euler_sim* sim, const char* path, int32_t upscale);
int euler_load_half_tank(euler_sim* sim);   /* synthetic config 3: solid ring, fluid in y < Y/2, at rest */
/* `tanks` such tanks on top of each other, each closed (solid where two meet): tank k is the single tank of an X x (Y / tanks) grid
 * in rows [k Y / tanks, (k+1) Y / tanks).  The weak-scaling workload: one tank per row slab.  tanks = 1 is euler_load_half_tank. */
int euler_load_half_tanks(euler_sim* sim, int32_t tanks);

/* Host-only pieces of sim_init, usable without a GPU (parser / marker seeding parity tests).
 * Outputs are caller-owned; solid/source/sink/fluid are [Y][X] uint8, markers float2[4*X*Y]. */
int euler_parse_scenario(const char* text, int32_t len, int32_t X, int32_t Y, int32_t upscale,
                         uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid);
int euler_seed_markers(const uint8_t* fluid, int32_t X, int32_t Y, uint64_t* rng_state,
                       float* markers_xy, uint64_t* n_markers);

/* ---- stepping = sim_step (main.c:843-900) ------------------------------------------------ */
int euler_step(euler_sim* sim);                    /* one frame: <= max_substeps CFL substeps */
int euler_timestep(euler_sim* sim, float frame_time_left, float* dt);   /* calculate_timestep, main.c:834-841 */
int euler_substep(euler_sim* sim, float dt);       /* stages 2..11 of sim_step with a given dt */
int euler_stage(euler_sim* sim, int32_t stage, float dt);   /* one EULER_STAGE_* (teacher-forced tests).  Row-slab handles: COLLECTIVE - the stage over the own rows with the
                                                                exchanges that belong to it (stages 0 .. 5 with one dt are a substep); not with euler_config.rainbow */
int euler_pcg_op(euler_sim* sim, int32_t op, float dt, double scalar_in, double* scalar_out);   /* one EULER_OP_* (kernel-level parity tests);
 * EULER_EINVAL for the preconditioner operations of a handle in EULER_PRECOND_IC0_TILE2 (its coarse level exists inside a solve only) */
/* Switch the preconditioner of the following solves (EULER_PRECOND_*; tile_records as euler_config.precond_tile_records,
 * 0 = default).  The solver's arrays do not depend on it; g_precon keeps whatever the last factorisation left. */
int euler_set_precond(euler_sim* sim, int32_t precond, int32_t tile_records);
/* The iteration budget and tolerance of the following solves (euler_config.max_iterations / tol; reference: 100 and 1e-6f,
 * main.c:735-736).  max_iterations <= 0 or tol < 0 leave the respective value as it is. */
int euler_set_solver(euler_sim* sim, int32_t max_iterations, double tol);

/* ---- state access ------------------------------------------------------------------------ */
int euler_get_field(euler_sim* sim, int32_t field, void* dst, size_t dst_bytes);
int euler_set_field(euler_sim* sim, int32_t field, const void* src, size_t src_bytes);
int euler_set_markers(euler_sim* sim, const float* xy, uint64_t n);
int euler_set_rng(euler_sim* sim, uint64_t rng_state, int32_t source_exhausted);
int euler_get_stats(euler_sim* sim, euler_stats* out);

/* ---- state snapshots: checkpoint / resume --------------------------------
 * Everything the reference keeps in file-scope variables (main.c:64-100, 204, 577), so that a resumed
 * run continues bit for bit.  File layout, little-endian:
 *   char[8] "EULERSNP"; u32 version = 1; i32 X, Y; u32 0; u64 n_markers; u64 rng_state;
 *   i32 source_exhausted; i32 0; u64 frames, total_substeps, total_pcg_iterations;          (72 bytes)
 *   f32[Y][X] u, v, utmp, vtmp;  u8[Y][X] solid, source, sink, count, prev_count;  f64[Y][X] precon;
 *   [version = 2, handles created with euler_config.rainbow: f32[Y][X] g_r, g_g, g_b, g_rtmp, g_gtmp, g_btmp;]
 *   f32[n_markers][2] markers in array order;  u64 FNV-1a-64 of all preceding bytes.
 * euler_load_state needs a handle of the same X, Y.  (euler_amd.read_snapshot / write_snapshot mirror
 * the format in numpy.)
 * ROW SLABS (version 3): every rank of a job calls euler_save_state with the SAME path and writes its own part file
 * "<path>.<rank>of<nranks>": the 72-byte header above with version = 3, then i32 nranks, rank, band_lo, band_hi, row_lo, row_hi,
 * u64 n_local_markers; the arrays above restricted to the rank's OWN rows [row_lo, row_hi); f32[n_local][2] markers,
 * u32[n_local] their keys (positions in the reference's g_markers); FNV-1a-64.  (Handles created with euler_config.rainbow: the
 * header's reserved word is 1 and the six dye arrays follow precon, as in version 2.)  Rank 0 also writes the manifest at <path>:
 * char[8] "EULERMAN"; u32 3; i32 X, Y, nranks; nranks x {i32 band_lo, band_hi}.
 * euler_load_state is independent of how a state was written: a whole-grid handle or a slab of ANY partition loads from a single
 * file or from a manifest, taking its rows (ghost rows included) and the markers inside its rows from the parts that hold them.
 * Resuming a 3-rank job on 2 ranks, on one GPU, or on a re-balanced partition is save + load.  On a row-slab handle the call is
 * collective (communicator installed first, like euler_load_scenario_*). */
int euler_save_state(euler_sim* sim, const char* path);
int euler_load_state(euler_sim* sim, const char* path);
size_t euler_field_bytes(const euler_sim* sim, int32_t field);   /* current size in bytes */

/* ---- render = draw_rows (main.c:914-951) ------------------------------------------------- */
/* Fetches only the visible window of the count grid from HBM.  Writes at most cap bytes into out
 * and stores the full length in *len (call with cap=0 to size the buffer).  On a row-slab handle the call is COLLECTIVE: the
 * visible rows are gathered from the ranks that own them and every rank returns the same frame. */
int euler_render(euler_sim* sim, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len);
/* The same formatter over caller-supplied host grids (no GPU needed). */
int euler_render_grids(const uint8_t* solid, const uint8_t* sink, const uint8_t* count,
                       int32_t X, int32_t Y, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len);
/* --rainbow: every water glyph carries a 24-bit colour escape built from the dye (main.c:902-912, 936-937;
 * misc/color.h:6-14).  euler_render uses it when the handle was created with euler_config.rainbow. */
int euler_render_grids_rgb(const uint8_t* solid, const uint8_t* sink, const uint8_t* count,
                           const float* r, const float* g, const float* b,
                           int32_t X, int32_t Y, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len);
/* colorize() (main.c:187-201), the reference's 'r' key (main.c:970-973): recolour the current fluid. */
int euler_colorize(euler_sim* sim);

/* ---- multi-GPU: 1-D row slabs (DESIGN.md "Multi-GPU") ------------------------------ */
/* One process per GPU.  Two layouts share the communicator interface below:
 *   row slabs for EVERY stage (euler_config.slab_nranks >= 1; the default of bench.py --gpus N): a handle holds only the rows of
 *     its 64-row bands (+ ghost rows) of every field and the markers inside them; ghost rows of u / v / counts, marker migration,
 *     the dt all-reduce and the distributed PCG all go through these operations;
 *   round 1's layout (euler_set_comm on a full-size handle): only the pressure solve - project(), main.c:709-806 - is partitioned
 *     into slabs of bands, the cheap stages run replicated on the whole grid.  Kept because its exact IC(0) coupling
 *     (EULER_SLAB_EXACT) is the one multi-rank mode that reproduces the single-GPU ITERATES.
 * The library drives the substep and calls these operations at its exchange points; the host program implements them
 * (euler_amd/slab.py: torch.distributed) or installs the library's own RCCL communicator (euler_set_comm_rccl).
 * All pointers are DEVICE pointers into the handle's buffers; operations must be ordered on the
 * stream given to euler_set_stream.  Each returns 0 on success. */
typedef struct euler_comm_ops {
  void*   ctx;
  int32_t rank, nranks;
  /* in-place all-reduce of `count` doubles: sum (is_max = 0) or max (is_max = 1) */
  int (*allreduce)(void* ctx, void* dev_f64, int32_t count, int32_t is_max);
  /* ghost rows of the search vector: send_lo/recv_lo <-> rank-1, send_hi/recv_hi <-> rank+1
   * (`count` doubles each; the ends of the chain skip the missing side) */
  int (*halo)(void* ctx, void* send_lo, void* send_hi, void* recv_lo, void* recv_hi, int32_t count);
  /* point-to-point: rank `src` sends nbytes at dev_ptr, rank `dst` receives into its own dev_ptr */
  int (*chain)(void* ctx, void* dev_ptr, int64_t nbytes, int32_t src, int32_t dst);
  /* all-gather: rank r contributes bytes [off[r], off[r]+cnt[r]) of the array at dev_base */
  int (*allgather)(void* ctx, void* dev_base, const int64_t* off, const int64_t* cnt);
  /* ONE exchange for a PCG iteration's latency-bound traffic (SURVEY 8e "fuse ... into one message pair"; optional - null: the
   * library issues halo + allgather instead): `count` doubles to / from each neighbour exactly like halo (count = 0: none), AND
   * an all-gather of `nsmall` doubles per rank: rank r's contribution sits at small[r * nsmall] (in place), every rank ends up
   * with all of them, which it folds in rank order - the same bits everywhere.  The built-in RCCL communicator issues all of it
   * as one group of sends and receives. */
  int (*exchange)(void* ctx, void* send_lo, void* send_hi, void* recv_lo, void* recv_hi, int32_t count, void* small, int32_t nsmall);
} euler_comm_ops;

enum {
  EULER_SLAB_EXACT = 1,   /* band hand-off rows forwarded rank to rank: the reference's IC(0), the
                             1-GPU iterates; the triangular sweeps then run one slab after another */
  EULER_SLAB_LOCAL = 0    /* IC(0) restricted to each slab (block-Jacobi across slabs): slabs sweep
                             concurrently; NOT the reference's iterates (tolerance only) */
};
int euler_set_comm(euler_sim* sim, const euler_comm_ops* ops, int32_t coupling);

/* The library's own communicator: the four operations above issued straight to RCCL (xGMI on an
 * MI355X node) on the handle's stream - no host code between two kernels of a PCG iteration.
 * Rank 0 creates an id (EULER_RCCL_ID_BYTES bytes = ncclUniqueId) and the launcher hands the same
 * bytes to every rank over whatever host channel it has (torchrun's store, MPI, a file); then every
 * rank calls euler_set_comm_rccl collectively, after hipSetDevice-equivalent euler_create on its own
 * GPU.  RCCL is bound at run time (dlopen of librccl.so.1; a copy already in the process is reused);
 * EULER_ECOMM if it is absent.  nranks = 1 is accepted and keeps the communicator code path (self-test). */
#define EULER_RCCL_ID_BYTES 128
int euler_rccl_unique_id(void* id_out, int32_t cap);
int euler_rccl_version(void);                                  /* NCCL-style version code, -1 if unavailable */
int euler_set_comm_rccl(euler_sim* sim, const void* unique_id, int32_t id_bytes, int32_t rank, int32_t nranks,
                        int32_t coupling);
/* Per-handle options (round 5): the forms and test hooks that used to be EULER_* environment variables read once per process.  Two handles of one process may
 * differ.  euler_set_option validates the value and the moment (a key that shapes allocations or a communicator must be set before what it shapes exists) and leaves
 * the handle untouched when it refuses; none of the keys changes a result's bits unless its line says so. */
enum {
  EULER_OPT_P_STEPS = 1,            /* 2, 4 or 8 (default): p += alpha s is applied N iterations at a time out of a ring of N search arrays (the fmadds of main.c:753 in their order: the same bits) */
  EULER_OPT_TILE_STORE_AS = 2,      /* 1: k_search_apply stores A s' and the r update reads it back (round 3's form; the same bits); 0 (default): formed twice */
  EULER_OPT_TILE_REVERSE = 3,       /* 0: k_precond_tile walks the chunk list upwards; 1 (default): downwards */
  EULER_OPT_RESIDENT_CAP = 4,       /* > 0: the resident solver's capacity in workgroups, at most the device's own (tests: a scene that outgrows the chip); 0: the device's */
  EULER_OPT_GRID4_MIN_CELLS = 5,    /* grids of at least this many cells run extrapolate / zero_bounds four cells per thread (default 2^22; the same bits) */
  EULER_OPT_SLAB_FUSION = 6,        /* 1: with the mailboxes connected, read the neighbouring slabs' edge rows where they live (set on every rank, before euler_p2p_connect) */
  EULER_OPT_RCCL_SMALL = 7,         /* the small all-gather inside the RCCL exchange: 0 by size (default), 1 always ncclAllGather, 2 always sends / receives (before euler_set_comm_rccl) */
  EULER_OPT_RCCL_NO_EXCHANGE = 8,   /* 1: the built-in communicator without the fused exchange (halo + allgather instead: the same traffic; before euler_set_comm_rccl) */
  EULER_OPT_MARKERS_ROWMAJOR = 9,   /* 1: the marker stages read the row-major grids (A-B timing; the same bits) */
  EULER_OPT_SA_RUN = 10,            /* 8 (default), 16, 32: pair-records per wave of k_search_apply (experiments).  16 / 32: the parity and the plain tile-local mode on one GPU only - refused (EULER_ESTATE) while a communicator or a coarse-correction mode is installed, and a handle that gets one later runs runs of 8 whatever the value */
  EULER_OPT_NO_INTERIOR = 11,       /* 1: no constant-mask instantiation for interior chunks (experiments; the same bits) */
  EULER_OPT_BUILD_GATHER = 12,      /* 1: the assembly as one diagonal gather (rounds 1-2; the same bits) */
  EULER_OPT_RESIDENT_FORCE_TIMEOUT = 13, /* test hook: the next n resident launches give up at once as if a wait had run out (error word 1): the time-out path */
  EULER_OPT_MG_SPLIT_LEVEL = 14,    /* multilevel mode on row slabs: the level whose right-hand side the ranks all-gather (below it every rank works on its own rows, docs/solver_multilevel.md): 0 (default) by size,
                                       n > 0 that level (tests: the split on small grids), -1 never (the cycle replicated from level 0 on).  The same value on every rank */
  EULER_OPT_MG_SPLIT_ACTIVE = 15,   /* read only: the gather level the last multilevel solve on row slabs ran with, 0 while the cycle runs replicated */
  EULER_OPT_MARKERS_TWO_PASS = 16,  /* 1: advect_markers and refresh_marker_counts as separate passes over the marker array (rounds 1-5; A-B timing; the same bits); 0 (default): the advection pass bins what it writes */
  EULER_OPT_BUILD_TWO_PASS = 17,    /* 1: the assembly as a row-major pass + a skewed gather (rounds 3-5; A-B timing; the same bits); 0 (default): one pass over parallelograms of the band-skewed layout */
  EULER_OPT_VELOCITY_TWO_PASS = 18, /* 1: k_finish_p + k_velocity_update as in rounds 1-5 (A-B timing; the same bits); 0 (default): one pass */
  EULER_OPT_NO_TILE_MAP = 19,       /* 1: the grid passes visit every cell as in rounds 1-5 (A-B timing; the same bits); 0 (default): tiles of 64 x 64 cells with no water in or next to them in
                                       the count grid and the previous one are left alone by advect_u / advect_v, zero_bounds, extrapolate and the marker stage's copies (their output there is the zeros already in place) */
  EULER_OPT_PROFILE_STRIDE = 20,    /* n >= 1 (default 1): euler_profile_enable brackets every n-th launch of an enabled kernel class with its event pair (a pair costs the stream
                                       a few microseconds of serialisation: all ~230 launches of an 8192^2 substep bracketed cost bench.py's headline 3.9 %); the class's time and
                                       launch count are those of the bracketed launches */
  EULER_OPT__COUNT
};
int euler_set_option(euler_sim* sim, int32_t key, int64_t value);
int euler_get_option(euler_sim* sim, int32_t key, int64_t* value);

int euler_comm_calls(euler_sim* sim, uint64_t out[5]);         /* built-in communicator: allreduce, halo, chain, allgather, exchange calls so far */

/* Peer-to-peer mailboxes for the latency-bound exchanges (csrc/comm_p2p.hip): the three 8-byte all-reduces
 * and the ghost-row exchange of every PCG iteration become direct writes into the peers' mailboxes (HIP IPC
 * mappings of fine-grained device memory; xGMI between the GPUs of a node) - one small kernel each, no
 * collective library in the loop, sums formed in rank order (bit-identical on every rank).  The bulk
 * transfers (band hand-off rows, all-gather of p) stay on the communicator installed before.
 *   1. every rank:  euler_p2p_export(sim, handle)          -> EULER_P2P_HANDLE_BYTES bytes (IPC handles)
 *   2. the launcher gathers the handles of all ranks, in rank order, on every rank (any host channel)
 *   3. every rank:  euler_p2p_connect(sim, handles, nranks) after euler_set_comm[_rccl]; maps the mailboxes and
 *      proves the path with an all-reduce of known values.  On failure the installed communicator stays as it is. */
#define EULER_P2P_HANDLE_BYTES 256   /* four hipIpcMemHandle_t: the mailbox and the z / s / s2 arrays (read across slab boundaries) */
int euler_p2p_export(euler_sim* sim, void* handle_out, int32_t cap);
int euler_p2p_connect(euler_sim* sim, const void* handles, int32_t nranks);
int euler_p2p_disconnect(euler_sim* sim);                      /* back to the installed communicator; frees the mailbox */
int euler_p2p_calls(euler_sim* sim, uint64_t out[2]);          /* all-reduces, ghost-row exchanges over the mailboxes so far */
int euler_set_stream(euler_sim* sim, void* hip_stream);   /* run on the caller's HIP stream (e.g. torch's) */
int euler_slab_info(euler_sim* sim, int32_t* band_lo, int32_t* band_hi, int32_t* nbands);

/* ---- measurement ------------------------------------------------------------------------- */
/* Per-kernel-class timing with HIP events on the library's own stream.  class_mask bit i
 * enables class i (see euler_profile_class_name); 0 disables. */
int euler_profile_enable(euler_sim* sim, uint64_t class_mask);
int euler_profile_class_count(void);
const char* euler_profile_class_name(int32_t cls);
int euler_profile_get(euler_sim* sim, int32_t cls, double* total_ms, uint64_t* launches);
int euler_profile_reset(euler_sim* sim);
/* Device-to-device copy bandwidth probe (float4 copy kernel), GB/s read+write. */
int euler_measure_copy_bandwidth(euler_sim* sim, size_t bytes, int32_t reps, double* gbps);
int euler_measure_exchange(euler_sim* sim, int32_t reps, int32_t row_doubles, int32_t nsmall, double* us_per_exchange);   /* collective: one exchange point of a distributed PCG iteration over the installed communicator, HIP-event time per call */
int euler_device_name(euler_sim* sim, char* out, int32_t cap);
uint64_t euler_hbm_bytes(const euler_sim* sim);   /* device memory the handle allocated (a row-slab handle: its slab only) */
/* Diagnostics: the band pipeline of the most recent IC(0) sweep launch.  For each of this rank's bands in
 * sweep order, 8 words: wave entry, first block's boundary ready, wave exit (100 MHz constant clock
 * ticks), (blocks run << 32 | blocks that had to wait for the previous band), and four hand-off time
 * stamps that only a development build fills (k_pcg.hip SW_TRACE_HANDOFF; else 0).  `out` holds
 * 8 * cap_bands words.  Returns the number of bands written (<= cap_bands) or a negative error. */
int euler_sweep_timeline(euler_sim* sim, uint64_t* out, int32_t cap_bands);

#ifdef __cplusplus
}
#endif
#endif /* EULER_H */
