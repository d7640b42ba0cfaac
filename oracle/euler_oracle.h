/*
 * euler_oracle.h — TEST ORACLE, not product code.
 *
 * A from-scratch CPU restatement of the cgmb/euler simulation path (reference main.c:102-900)
 * with a run-time grid size.  It exists so that
 *   (1) tests/ can check the HIP path bit-for-bit / to tolerance on grids the reference cannot
 *       run (the reference grid is a compile-time enum, main.c:22-25), and
 *   (2) bench.py's `cpu_baseline` leg can time "the reference CPU path" on the GPU node.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product library (euler_amd/csrc) never does.
 *
 * Pinning: at X=100,Y=40 the oracle is checked bit-for-bit against the compiled, unmodified
 * reference (oracle/_ref/libeuler_ref.so) and against the committed fixtures in tests/golden/
 * (tests/test_oracle_golden.py, tests/test_oracle_vs_ref.py).
 */
#ifndef EULER_ORACLE_H
#define EULER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eo_vec2f { float x, y; } eo_vec2f;

typedef struct eo_sim {
  int X, Y;                 /* grid size (reference: enum X=100,Y=40, main.c:22-25) */
  float *u, *v, *utmp, *vtmp;           /* main.c:64-67, all [Y][X] */
  uint8_t *solid, *source, *sink;       /* main.c:71-73 */
  uint8_t *count, *prev_count;          /* g_marker_count / g_prev_marker_count, main.c:96-97 */
  eo_vec2f *markers;                    /* main.c:95, capacity 4*X*Y */
  size_t n_markers, max_markers;        /* main.c:92-93 */
  int source_exhausted;                 /* main.c:94 */
  uint64_t rng_state;                   /* main.c:204 (static inside randf) */
  int8_t *a_diag;                       /* g_a, main.c:552 (stale on non-fluid cells) */
  double *precon, *q;                   /* g_precon, g_q, main.c:577-578 (precon persists) */
  double *b, *p, *r, *z, *s;            /* stack locals of project(), main.c:716-745 */
  /* solver parameters (reference constants main.c:735-736, 849, 851, 838) */
  int max_iterations;                   /* 100 */
  double tol;                           /* (double)1e-6f */
  float viscosity;                      /* EXTENSION, not in the reference: 0 = off = the reference */
  /* counters the reference does not keep */
  uint64_t total_substeps, total_pcg_iterations;
  int last_substeps, last_pcg_iterations;
  double last_residual;
  float last_dt;
  uint32_t frame_count;
  /* --rainbow dye (main.c:75-83): g_r/g/b and their advection scratch; g_rainbow_enabled.  The arrays
   * are always there (update_fluid_sources writes the source colour regardless, main.c:292-294). */
  int rainbow;
  float *cr, *cg, *cb, *crtmp, *cgtmp, *cbtmp;
  /* EXTENSION (no reference counterpart; parity unpinned by the reference): tile-local IC(0), the block-Jacobi
   * restriction of the reference's preconditioner.  0 = the reference's IC(0) (main.c:580-627); else the records per
   * tile.  See eo_tile_start. */
  int tile_records;
  /* EXTENSION on top of it (round 3): a coarse correction - the "two-level" preconditioner
   *     z = M_tile^-1 r + P (P^T A P)^-1 P^T r,
   * P = piecewise constants over coarse cells of (64 m) x (64 m) grid cells restricted to the fluid (coarse_m = m > 0; 0 = off).
   * The block-local IC(0) has no coupling beyond a 64 x 16 block; the coarse space restores the long-range part of the inverse
   * (the hydrostatic mode of a tank, say), which is what the first hundred iterations of a solve live on.  Still symmetric positive
   * definite: PCG converges to the same solution.  The coarse matrix (at most 256 x 256) is factored once per system
   * (eo_build_system invalidates it). */
  int coarse_m;
  int coarse_n, coarse_nx;      /* filled by the first application: number of coarse cells, coarse cells per row */
  double* coarse_chol;          /* dense lower Cholesky factor of P^T A P (coarse_n^2), NULL = not factored yet */
  /* EXTENSION on top of that (round 3; round 5: bilinear coarse spaces): the MULTILEVEL form of the coarse correction (coarse_mg != 0, with coarse_m > 0):
   *     z = M_tile^-1 r + P_0 V(P_0^T r)
   * P_0 = bilinear interpolation from a grid of nodes, one per 16 x 16 grid cells at the block's centre, restricted to the fluid; V = one symmetric V-cycle for
   * A_0 = P_0^T A P_0 over a hierarchy of node grids, each with a node per 2 x 2 nodes of the one below and bilinear interpolation again (Galerkin operators:
   * nine-point stencils), damped Jacobi (omega = 0.8) once before and once after the coarse-grid correction, the top level (at most 64 nodes) solved exactly
   * with coarse_chol.  A fixed symmetric positive definite operator: PCG converges to the same solution, and the iteration count does not grow with the grid
   * (tank at rest to 1e-6: 880 iterations with the reference's IC(0) at 1024^2, 104 with round 4's piecewise-constant aggregates, 52 with these).
   * The hierarchy is rebuilt per system (eo_build_system invalidates it). */
  int coarse_mg;
  void* mg;                     /* the hierarchy (euler_oracle.c: mg_hierarchy), NULL = not built yet */
  /* fluid cut off from the air (a closed box full of water): P^T A P is singular along the indicator of such a component; the factor pins one of its cells
   * (the pivot falls back to the diagonal, like main.c:595) and the solve projects the indicators out on both sides - the pseudo-inverse */
  int coarse_npinned, coarse_pinned[16];
  double* coarse_null;          /* [coarse_npinned][coarse_n] */
  /* EXTENSION (round 4; BASELINE configs[1] "fp32"): the tile-local PCG with every solver vector in FLOAT - r, z, s, p, A s and E^-1 are rounded to float after
   * every operation, the dot products and the scalars alpha, beta, sigma stay double (products of two floats are exact in double).  Not the reference's iterates
   * (its PCG is double, main.c:577-578,716): tolerance parity only.  Needs tile_records > 0 and no coarse correction.  The arrays keep their double storage and
   * hold float values. */
  int pcg_f32;
  int coarse_bw;                /* half-bandwidth of coarse_chol */
  double mg_theta;              /* tests: the bound of the cycle's per-node Jacobi damping (mg_damping); 0 = the product's MG_THETA, 1e30 = plain omega on every node */
} eo_sim;

eo_sim* eo_create(int X, int Y);
void    eo_destroy(eo_sim* s);

/* Scenario text -> solid/source/sink + jittered markers (main.c:209-274).
 * upscale = 0: the reference's streaming parser at the native resolution.
 * upscale = 1: SURVEY.md §8d nearest-neighbour resample of the W x H text onto the interior. */
int eo_load_scenario_mem(eo_sim* s, const char* text, int len, int upscale);
int eo_load_scenario_file(eo_sim* s, const char* path, int upscale);
/* Synthetic half-filled tank: solid ring at 1 / N-2, fluid y in [2,Y/2). */
int eo_load_half_tank(eo_sim* s);

/* One frame = sim_step() (main.c:843-900): <= 8 CFL substeps of 0.1 s total. */
void  eo_step(eo_sim* s);
/* One substep with a given dt (stages 2-11 of sim_step). Returns PCG iterations. */
int   eo_substep(eo_sim* s, float dt);

/* individual stages, in sim_step order (for teacher-forced tests) */
float eo_calculate_timestep(const eo_sim* s, float frame_time);        /* main.c:834-841 */
void  eo_advect_markers(eo_sim* s, float dt);                          /* main.c:464-537 */
void  eo_refresh_marker_counts(eo_sim* s);                             /* main.c:102-117 */
void  eo_update_fluid_sources(eo_sim* s);                              /* main.c:276-298 */
void  eo_extrapolate(eo_sim* s, float* q, int type);                   /* main.c:173-185 */
void  eo_zero_bounds(const eo_sim* s, float* q, int type);             /* main.c:822-832 */
void  eo_advect_u(const eo_sim* s, const float* u, const float* v, float dt, float* out);
void  eo_advect_v(const eo_sim* s, const float* u, const float* v, float dt, float* out);
void  eo_apply_body_forces(const eo_sim* s, float* v, float dt);       /* main.c:539-545 */
void  eo_colorize(eo_sim* s);                                          /* main.c:187-201 */
void  eo_advect_p(const eo_sim* s, const float* q, const float* u, const float* v, float dt, float* out); /* main.c:424-438 */
/* EXTENSION (no reference counterpart): one explicit diffusion step of the typed field q
 * over its live faces (fluid property and not solid), out = q + nu*dt/h^2 * sum over live 4-neighbours (q_n - q). */
void  eo_diffuse(const eo_sim* s, const float* q, int type, float dt, float* out);
int   eo_project(eo_sim* s, float dt, const float* u, const float* v, float* uout, float* vout);
float eo_interpolate(const eo_sim* s, const float* q, float ix, float iy, int type);

/* PCG building blocks (main.c:580-702), exposed for kernel-level parity tests. */
void   eo_build_system(eo_sim* s, float dt, const float* u, const float* v); /* b, a_diag */
void   eo_apply_preconditioner(eo_sim* s, const double* r, double* z);
void   eo_apply_a(const eo_sim* s, const double* in, double* out);
double eo_dot(const eo_sim* s, const double* a, const double* b);
double eo_inf_norm(const eo_sim* s, const double* r);
/* EXTENSION: the blocks of the tile-local preconditioner.  The cells of a 64-row band are numbered by their
 * skew record t = x + y % 64; tile k of a band = the records [k * W, (k + 1) * W), W = tile_records.  A block = the cells
 * of one band whose records fall into one tile (a parallelogram of 64 rows x W columns).  IC(0) is applied to the
 * block-diagonal part of A: every coupling between cells of different blocks is dropped from the factor and from both
 * triangular solves (the diagonal of A stays whole).
 * Returns 1 when record t starts a tile (couplings arriving from record t-1 are cut), else 0. */
int    eo_tile_start(int tile_records, int t);
/* the coarse-cell width the product picks for a grid: the smallest power of two m with ceil(X / 64m) * ceil(Y / 64m) <= 256 */
int    eo_coarse_m(int X, int Y);

/* ASCII frame as draw_rows() emits it (main.c:914-951), no cursor codes. Returns length. */
int eo_render_rows(const eo_sim* s, int wx, int wy, char* out, int cap);

uint64_t eo_fnv1a64(const void* data, size_t n);

enum { EO_P = 0, EO_U = 1, EO_V = 2 };

#ifdef __cplusplus
}
#endif
#endif
