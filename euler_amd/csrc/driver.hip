// driver.hip — the C ABI of libeuler_hip.so (include/euler.h): handle life cycle, scenario upload,
// the sim_step() driver (reference main.c:843-900), state access, render, measurement.
#include "euler_dev.h"
#include "k_mg.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

int eu_launch_build_system(euler_sim* S, float dt);
int eu_launch_velocity_update(euler_sim* S, float dt, int finish);
int eu_pressure_current(euler_sim* S);      // k_grid.hip
int eu_unskew(euler_sim* S, const void* skew, void* rowmajor, int elem_bytes);
int eu_unskew_fluid(euler_sim* S, const double* skew, double* rowmajor);
double* eu_current_search_direction(euler_sim* S);
int eu_launch_tile_table(euler_sim* S);
int eu_skew(euler_sim* S, const void* rowmajor, void* skew, int elem_bytes);

// ------------------------------------------------------------------------------------------
// errors
static thread_local char g_err[512] = "";

void eu_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int eu_hip_fail(hipError_t e, const char* what, const char* file, int line) {
  eu_set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return EULER_EHIP;
}

extern "C" const char* euler_last_error(void) { return g_err; }
extern "C" int euler_abi_version(void) { return EULER_ABI_VERSION; }

// ------------------------------------------------------------------------------------------
// profiling: hipEvent pairs around the launches of the enabled kernel classes
static const char* k_class_names[KC__COUNT] = {
    "timestep", "marker_advect", "marker_events", "marker_bin", "marker_compact", "sources", "select",
    "extrapolate", "advect_velocity", "build_system", "precon_factor", "forward_solve", "backward_solve",
    "apply_a", "dot", "update_pr", "update_search", "reduce_final", "velocity_update", "jacobi", "misc", "precond_tile", "coarse_cycle", "resident_pcg"};

void eu_prof_begin(euler_sim* S, int cls) {
  if (!((S->prof_mask >> cls) & 1)) return;
  const long long stride = S->opt[EULER_OPT_PROFILE_STRIDE];
  if (stride > 1 && (S->prof_seq[cls]++ % (unsigned int)stride) != 0) return;      // (an event pair serialises the stream for a few microseconds: every n-th launch is a sample)
  S->prof_open |= 1ull << cls;
  if (S->ev_used + 2 > S->ev_cap) eu_prof_flush(S);
  const int k = S->ev_used / 2;
  S->ev_cls[k] = cls;
  S->ev_solve[k] = S->solve_seq;
  S->ev_iter[k] = S->prof_iter;
  (void)hipEventRecord(S->ev_pool[S->ev_used], S->stream);
}
void eu_prof_end(euler_sim* S, int cls) {
  if (!((S->prof_open >> cls) & 1)) return;
  S->prof_open &= ~(1ull << cls);
  (void)hipEventRecord(S->ev_pool[S->ev_used + 1], S->stream);
  S->ev_used += 2;
}
// Accumulate finished event pairs.  A PCG launch of solve q / iteration i did work iff the solve's
// rhs was non-zero and i < the iteration count the DEVICE reached (launches enqueued beyond
// convergence return at once and must not dilute the average launch time).
int eu_prof_flush(euler_sim* S) {
  if (S->ev_used == 0) return EULER_OK;
  HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  const int cur_iters = S->sc_host->nonzero ? S->sc_host->iters : -1;
  for (int k = 0; k < S->ev_used; k += 2) {
    const int e = k / 2, cls = S->ev_cls[e];
    bool active = true;
    if (S->ev_iter[e] > -2) {
      const int q = S->ev_solve[e];
      const int iters = q == S->solve_seq ? cur_iters : (S->solve_seq - q < 256 ? S->solve_iters[q & 255] : 0);
      active = iters >= 0 && S->ev_iter[e] < iters;
      if (S->ev_iter[e] == -1) active = iters >= 0;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, S->ev_pool[k], S->ev_pool[k + 1]) != hipSuccess) continue;
    if (active) { S->prof_ms[cls] += ms; S->prof_launches[cls] += 1; }
    else S->prof_idle[cls] += 1;
  }
  S->ev_used = 0;
  return EULER_OK;
}

extern "C" int euler_profile_class_count(void) { return KC__COUNT; }
extern "C" const char* euler_profile_class_name(int32_t c) { return (c >= 0 && c < KC__COUNT) ? k_class_names[c] : ""; }
extern "C" int euler_profile_enable(euler_sim* S, uint64_t mask) {
  if (!S) return EULER_EINVAL;
  int rc = eu_prof_flush(S);
  S->prof_mask = mask;
  S->prof_open = 0;
  return rc;
}
extern "C" int euler_profile_get(euler_sim* S, int32_t cls, double* ms, uint64_t* launches) {
  if (!S || cls < 0 || cls >= KC__COUNT) return EULER_EINVAL;
  int rc = eu_prof_flush(S);
  if (ms) *ms = S->prof_ms[cls];
  if (launches) *launches = S->prof_launches[cls];
  return rc;
}
extern "C" int euler_profile_reset(euler_sim* S) {
  if (!S) return EULER_EINVAL;
  int rc = eu_prof_flush(S);
  memset(S->prof_ms, 0, sizeof(S->prof_ms));
  memset(S->prof_launches, 0, sizeof(S->prof_launches));
  memset(S->prof_idle, 0, sizeof(S->prof_idle));
  return rc;
}

// ------------------------------------------------------------------------------------------
// life cycle
extern "C" int euler_config_default(euler_config* c) {
  if (!c) return EULER_EINVAL;
  memset(c, 0, sizeof(*c));
  c->abi_version = EULER_ABI_VERSION;
  c->X = 100; c->Y = 40;            // main.c:22-25
  c->device = 0;
  c->max_iterations = 100;          // main.c:735
  c->tol = (double)1e-6f;           // main.c:736
  c->dot_mode = EULER_DOT_AUTO;
  c->precond = EULER_PRECOND_IC0;
  c->sweep_mode = EULER_SWEEP_AUTO;
  c->max_substeps = 8;              // main.c:851
  c->frame_time = 0.1f;             // main.c:849
  c->viscosity = 0.f;
  c->pcg_poll_interval = 8;
  return EULER_OK;
}

// records per tile of the tile-local preconditioner (include/euler.h precond_tile_records; the oracle's eo_tile_start)
static int eu_set_tiles(euler_sim* S, int w) {
  if (w <= 0) w = 16;
  if (w != 8 && w != 16 && w != 32) { eu_set_error("precond_tile_records = %d: 8, 16 or 32 (0 = 16)", w); return EULER_EINVAL; }
  S->cfg.precond_tile_records = w;
  S->tile_w = w;
  return EULER_OK;
}

// the resident solver (k_resident.hip) can run this handle's solves: plain tile-local preconditioner, one GPU, tree dots, and EVERY chunk of the grid
// finds a wave on the chip at once (so that no scene ever outgrows it)
bool eu_resident_eligible(const euler_sim* S) {
  if (S->res_disabled || S->cfg.resident == EULER_RESIDENT_OFF) return false;
  if (S->cfg.precond != EULER_PRECOND_IC0_TILE || S->tile_w != 16 || S->cfg.sweep_mode == EULER_SWEEP_SIMPLE || S->cfg.dot_mode != EULER_DOT_TREE) return false;
  if (S->has_comm || S->slab_on || S->p2p_on) return false;
  // double: a solve whose active chunks do not all fit takes the multi-kernel path (decided per solve, eu_launch_project); float has no other path, so
  // the whole grid must fit
  const int cap = eu_resident_capacity(const_cast<euler_sim*>(S), S->cfg.pcg_precision == EULER_PCG_F32);
  if (S->cfg.pcg_precision == EULER_PCG_F32) return cap > 0 && S->geom.nbands * (S->geom.T / 16) <= 4 * cap;
  return cap > 0;      // (any grid: what counts is the ACTIVE chunks of a solve - BASELINE configs[4], the 4096^2 waterfall, starts at 1.6 % water: 71 -> 11.7 us per iteration until it has grown past a million cells; where a solve does not fit, the extra host round trip is ~30 us against milliseconds of solve)
}
extern "C" int euler_resident_info(euler_sim* S, uint64_t out[3]) {
  if (!S || !out) return EULER_EINVAL;
  out[0] = eu_resident_eligible(S) ? 1 : 0; out[1] = S->res_solves; out[2] = S->res_fallbacks;
  return EULER_OK;
}

// Per-handle options (include/euler.h EULER_OPT_*): validated here, read where they act.  A refused call changes nothing.
extern "C" int euler_set_option(euler_sim* S, int32_t key, int64_t value) {
  if (!S || key <= 0 || key >= EULER_OPT__COUNT) { eu_set_error("euler_set_option: unknown key %d", (int)key); return EULER_EINVAL; }
  bool ok = true;
  const char* when = nullptr;
  switch (key) {
    case EULER_OPT_P_STEPS: ok = value == 2 || value == 4 || value == 8; break;
    case EULER_OPT_SA_RUN: ok = value == 8 || value == 16 || value == 32;
      if (ok && value != 8 && (S->has_comm || eu_is_two_level(S))) when = "on a handle without a communicator, in the parity or the plain tile-local mode (the others run runs of 8)";
      break;
    case EULER_OPT_RCCL_SMALL: ok = value >= 0 && value <= 2; if (S->rccl) when = "before euler_set_comm_rccl"; break;
    case EULER_OPT_RCCL_NO_EXCHANGE: ok = value == 0 || value == 1; if (S->rccl) when = "before euler_set_comm_rccl"; break;
    case EULER_OPT_SLAB_FUSION: ok = value == 0 || value == 1; if (S->p2p_on) when = "before euler_p2p_connect"; break;
    case EULER_OPT_RESIDENT_CAP: case EULER_OPT_GRID4_MIN_CELLS: case EULER_OPT_RESIDENT_FORCE_TIMEOUT: ok = value >= 0; break;
    case EULER_OPT_MG_SPLIT_LEVEL: ok = value >= -1 && value < 12; break;
    case EULER_OPT_PROFILE_STRIDE: ok = value >= 1 && value <= 1024; break;
    case EULER_OPT_MG_SPLIT_ACTIVE: ok = false; break;
    default: ok = value == 0 || value == 1; break;
  }
  if (!ok) { eu_set_error("euler_set_option: key %d does not take the value %lld", (int)key, (long long)value); return EULER_EINVAL; }
  if (when) { eu_set_error("euler_set_option: key %d must be set %s", (int)key, when); return EULER_ESTATE; }
  HIPCHK(hipStreamSynchronize(S->stream));
  if (key == EULER_OPT_P_STEPS || key == EULER_OPT_SA_RUN) {      // the ring is about to be forgotten: the pressure the last velocity update left unfinished in memory is finished from it first
    int rcp = eu_pressure_current(S);
    if (rcp) return rcp;
  }
  S->opt[key] = value;
  if (key == EULER_OPT_P_STEPS || key == EULER_OPT_SA_RUN) S->s_ring_n = 0;      // (the next solve sets its ring up afresh)
  if (key == EULER_OPT_MG_SPLIT_LEVEL) { eu_mg_split_release(S); S->opt[EULER_OPT_MG_SPLIT_ACTIVE] = 0; }      // (... plans its cycle afresh)
  return EULER_OK;
}
extern "C" int euler_get_option(euler_sim* S, int32_t key, int64_t* value) {
  if (!S || !value || key <= 0 || key >= EULER_OPT__COUNT) { eu_set_error("euler_get_option: unknown key %d", (int)key); return EULER_EINVAL; }
  *value = S->opt[key];
  return EULER_OK;
}

// which preconditioner / communicator combinations a solve can run (k_pcg.hip eu_launch_project): checked where the combination is made, not in the middle of a solve
static const char* eu_precond_combination(const euler_sim* S, int precond, int tile_w, int has_comm, int slab_on, int p2p_on) {
  const bool coarse = precond == EULER_PRECOND_IC0_TILE2 || precond == EULER_PRECOND_IC0_TILE_MG;
  if (!coarse) return nullptr;
  if (tile_w != 16 || S->cfg.sweep_mode == EULER_SWEEP_SIMPLE) return "the coarse-correction preconditioners run on tiles of 16 records with the band schedule";
  if (precond == EULER_PRECOND_IC0_TILE2 && (has_comm || slab_on)) return "EULER_PRECOND_IC0_TILE2 runs on one GPU (row slabs: EULER_PRECOND_IC0_TILE_MG)";
  if (precond == EULER_PRECOND_IC0_TILE_MG && has_comm && (!slab_on || p2p_on)) return "EULER_PRECOND_IC0_TILE_MG with several ranks: row-slab handles without mailboxes only";
  return nullptr;
}

extern "C" int euler_set_precond(euler_sim* S, int32_t precond, int32_t tile_records) {
  if (!S || precond < EULER_PRECOND_IC0 || precond > EULER_PRECOND_IC0_TILE_MG) { eu_set_error("euler_set_precond: bad argument"); return EULER_EINVAL; }
  // validate first: a refused call leaves the handle exactly as it was
  const int w = tile_records <= 0 ? 16 : tile_records;
  if (w != 8 && w != 16 && w != 32) { eu_set_error("precond_tile_records = %d: 8, 16 or 32 (0 = 16)", tile_records); return EULER_EINVAL; }
  if (const char* why = eu_precond_combination(S, precond, w, S->has_comm, S->slab_on, S->p2p_on)) { eu_set_error("euler_set_precond: %s", why); return EULER_EINVAL; }
  if (S->cfg.pcg_precision == EULER_PCG_F32 && (precond != EULER_PRECOND_IC0_TILE || w != 16)) { eu_set_error("euler_set_precond: an EULER_PCG_F32 handle runs EULER_PRECOND_IC0_TILE with tiles of 16 records only"); return EULER_EINVAL; }
  HIPCHK(hipStreamSynchronize(S->stream));
  const bool coarse = precond == EULER_PRECOND_IC0_TILE2 || precond == EULER_PRECOND_IC0_TILE_MG;
  if (coarse) { int rc = eu_coarse_alloc(S, precond == EULER_PRECOND_IC0_TILE_MG); if (rc) return rc; }      // their arrays come with the first use
  (void)eu_set_tiles(S, w);
  // the coarse modes fold their sums as trees: EULER_DOT_TREE while one of them is selected, the caller's own mode (EULER_DOT_SEQUENTIAL: the
  // reference's order of the dot products, the bit-identical parity mode) again afterwards
  S->cfg.dot_mode = coarse ? EULER_DOT_TREE : S->dot_mode_user;
  if (S->has_comm && S->cfg.dot_mode == EULER_DOT_SEQUENTIAL) S->cfg.dot_mode = EULER_DOT_TREE;
  // (a row-slab handle runs whatever preconditioner it is given WITHOUT coupling between the slabs - eu_install_comm forced S->couple = 0 and
  // refuses EULER_SLAB_EXACT for it - so EULER_PRECOND_IC0 here means slab-local IC(0): valid, and not the single-GPU iterates)
  S->cfg.precond = precond;
  S->lean_ok = 0;      // the next assembly writes the solver arrays whole (k_build_system)
  return EULER_OK;
}

extern "C" int euler_set_solver(euler_sim* S, int32_t max_iterations, double tol) {
  if (!S) return EULER_EINVAL;
  HIPCHK(hipStreamSynchronize(S->stream));
  if (max_iterations > 0) S->cfg.max_iterations = max_iterations;
  if (tol >= 0.0) S->cfg.tol = tol;
  return EULER_OK;
}

static thread_local size_t g_alloc_bytes = 0;      // what the handle under construction has allocated (euler_hbm_bytes)
template <typename T>
static int dalloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  g_alloc_bytes += n * sizeof(T);
  HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
  return EULER_OK;
}
#define DALLOC(p, n) do { int _rc = dalloc(&(p), (n)); if (_rc) { euler_destroy(S); return _rc; } } while (0)

extern "C" void euler_destroy(euler_sim* S) {
  if (!S) return;
  if (S->stream) (void)hipStreamSynchronize(S->stream);
  eu_p2p_release(S);
  eu_rccl_release(S);
  eu_slab_release(S);
  eu_coarse_release(S);
  // row-major arrays are held by base pointers shifted to global (x, y) indexing: allocation = pointer + win_off
  // (a handle that failed half-way through euler_create still holds the raw allocations: S->shifted)
  const size_t wo = S->shifted ? S->win_off : 0;
  for (float* f : {S->u, S->v, S->utmp, S->vtmp}) if (f) (void)hipFree(f + wo);
  for (uint8_t* g : {S->solid, S->source, S->sink, S->count, S->prev_count}) if (g) (void)hipFree(g + wo);
  if (S->count32) (void)hipFree(S->count32);
  if (S->blockedT) (void)hipFree(S->blockedT);
  if (S->tmap) (void)hipFree(S->tmap);
  if (S->uT) (void)hipFree(S->uT);
  if (S->vT) (void)hipFree(S->vT);
  if (S->countT) (void)hipFree(S->countT);
  if (S->solidT) (void)hipFree(S->solidT);
  if (S->sys_m) (void)hipFree(S->sys_m);
  if (S->sys_div) (void)hipFree(S->sys_div);
  void* dev[] = {S->markers[0], S->markers[1], S->keys[0], S->keys[1], S->ms, S->evmask, S->delmask, S->ev_theta, S->ev_delta, S->sel_idx, S->act_idx,
                 S->act_dt, S->cellmask64, S->draws, S->sel.block_sums, S->sc, S->partial, S->red_counter, S->granules, S->ticket, S->sweep_timeline, S->halo_buf, S->band_ranges, S->partial2, S->pair_buf, S->xrows, S->alpha_buf, S->rng_jump, S->chunk_flag, S->chunk_prev, S->chunk_part, S->tile_table, S->chunk_bits, S->chunk_list,
                 S->rowmajor_tmp};
  for (void* p : dev) if (p) (void)hipFree(p);
  for (float* d : S->dye) if (d) (void)hipFree(d + wo);
  // band-skewed arrays: shifted to global element indexing as well (skew_off), behind EU_SKEW_SLACK elements of slack
  const size_t so = S->shifted ? S->skew_off : 0, sl = S->shifted ? (size_t)EU_SKEW_SLACK : 0;
  (void)so; (void)sl;
  for (void* d : S->s_ring_alloc) if (d) (void)hipFree(d);
  for (void* d : S->skew_alloc) if (d) (void)hipFree(d);      // (the raw allocations: s / s2 swap during solves, and each array has its own stagger)
  const size_t fb_off = S->shifted ? (size_t)S->ab_lo * S->fb_stride * 64 : 0;
  if (S->fbits_fwd) (void)hipFree(S->fbits_fwd + fb_off);
  if (S->fbits_bwd) (void)hipFree(S->fbits_bwd + fb_off);
  if (S->ms_host) (void)hipHostFree(S->ms_host);
  if (S->sc_host) (void)hipHostFree(S->sc_host);
  if (S->poll_host) (void)hipHostFree(S->poll_host);
  if (S->res_err) (void)hipHostFree(S->res_err);
  if (S->res_gran) (void)hipFree(S->res_gran);
  for (hipEvent_t e : S->poll_event) if (e) (void)hipEventDestroy(e);
  if (S->ev_pool) { for (int k = 0; k < S->ev_cap; ++k) if (S->ev_pool[k]) (void)hipEventDestroy(S->ev_pool[k]); free(S->ev_pool); }
  free(S->ev_cls); free(S->ev_solve); free(S->ev_iter);
  if (S->stream && S->own_stream) (void)hipStreamDestroy(S->stream);
  free(S);
}

extern "C" int euler_create(const euler_config* cfg, euler_sim** out) {
  if (!cfg || !out) { eu_set_error("euler_create: null argument"); return EULER_EINVAL; }
  if (cfg->abi_version != EULER_ABI_VERSION) { eu_set_error("euler_create: ABI version %d != %d", cfg->abi_version, EULER_ABI_VERSION); return EULER_EINVAL; }
  if (cfg->X < 8 || cfg->Y < 8 || (size_t)cfg->X * cfg->Y > (size_t)1 << 30) { eu_set_error("euler_create: grid %dx%d out of range", cfg->X, cfg->Y); return EULER_EINVAL; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    eu_set_error("euler_create: no HIP device (%s); libeuler_hip has no CPU path", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return EULER_EHIP;
  }
  if (cfg->device < 0 || cfg->device >= ndev) { eu_set_error("euler_create: device %d of %d", cfg->device, ndev); return EULER_EINVAL; }
  HIPCHK(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    eu_set_error("euler_create: device is %s; this library carries gfx950 code only", prop.gcnArchName);
    return EULER_EHIP;
  }

  euler_sim* S = (euler_sim*)calloc(1, sizeof(euler_sim));
  if (S) { S->opt[EULER_OPT_P_STEPS] = 8; S->opt[EULER_OPT_TILE_REVERSE] = 1; S->opt[EULER_OPT_GRID4_MIN_CELLS] = 1ll << 22; S->opt[EULER_OPT_SA_RUN] = 8; S->opt[EULER_OPT_PROFILE_STRIDE] = 1; }      // (euler_set_option's defaults)
  if (!S) return EULER_ENOMEM;
  g_alloc_bytes = 0;
  S->cfg = *cfg;
  if (S->cfg.max_iterations <= 0) S->cfg.max_iterations = 100;
  if (S->cfg.max_substeps <= 0) S->cfg.max_substeps = 8;
  if (S->cfg.frame_time <= 0.f) S->cfg.frame_time = 0.1f;
  S->X = cfg->X; S->Y = cfg->Y;
  const size_t C = S->C = (size_t)cfg->X * cfg->Y;
  if (S->cfg.dot_mode == EULER_DOT_AUTO) S->cfg.dot_mode = C <= 65536 ? EULER_DOT_SEQUENTIAL : EULER_DOT_TREE;
  S->dot_mode_user = S->cfg.dot_mode;
  if (S->cfg.sweep_mode == EULER_SWEEP_AUTO) S->cfg.sweep_mode = EULER_SWEEP_BAND;
  S->geom.X = S->X; S->geom.Y = S->Y;
  S->geom.nbands = (S->Y + 63) / 64;
  S->geom.T = (S->X + 63 + 95) / 96 * 96;   // records per band in whole units of 96 (euler_dev.h); even: records are stored in pairs
  S->geom.TS = (S->geom.T + 31) / 32 * 32 + 64;   // the sweeps run whole groups of three / four 8-step blocks and prefetch up to 24 steps further
  S->geom.S = (size_t)S->geom.nbands * S->geom.TS * 64;
  // row slabs: this handle holds the rows of its bands only (+ ghost rows); without slabs the window is the grid
  const int nbands = S->geom.nbands;
  S->slab_on = cfg->slab_nranks >= 1;      // (1 rank: the same code path with nobody to exchange with - transport self-tests)
  if (S->slab_on) {
    if (cfg->slab_rank < 0 || cfg->slab_rank >= cfg->slab_nranks || cfg->slab_nranks > nbands) {
      eu_set_error("euler_create: slab rank %d of %d for %d bands of 64 rows", cfg->slab_rank, cfg->slab_nranks, nbands);
      free(S); return EULER_EINVAL;
    }
    if (S->cfg.sweep_mode == EULER_SWEEP_SIMPLE) {
      eu_set_error("euler_create: row slabs do not run EULER_SWEEP_SIMPLE");
      free(S); return EULER_EINVAL;
    }
    S->band_lo = (int)((int64_t)nbands * cfg->slab_rank / cfg->slab_nranks);
    S->band_hi = (int)((int64_t)nbands * (cfg->slab_rank + 1) / cfg->slab_nranks);
    if (cfg->slab_band_hi > cfg->slab_band_lo || cfg->slab_band_lo != 0) {      // an explicit partition (include/euler.h)
      const int lo = cfg->slab_band_lo, hi = cfg->slab_band_hi;
      const bool first = cfg->slab_rank == 0, last = cfg->slab_rank + 1 == cfg->slab_nranks;
      if (lo < 0 || hi <= lo || hi > nbands || (first && lo != 0) || (last && hi != nbands) || (!first && lo < cfg->slab_rank) ||
          (!last && nbands - hi < cfg->slab_nranks - 1 - cfg->slab_rank)) {
        eu_set_error("euler_create: slab rank %d of %d cannot own bands [%d, %d) of %d (every rank needs a band; the ranges tile the grid in rank order)",
                     cfg->slab_rank, cfg->slab_nranks, lo, hi, nbands);
        free(S); return EULER_EINVAL;
      }
      S->band_lo = lo; S->band_hi = hi;
    }
    if (S->cfg.dot_mode == EULER_DOT_SEQUENTIAL) S->cfg.dot_mode = EULER_DOT_TREE;   // the replay order is a 1-rank notion
  } else { S->band_lo = 0; S->band_hi = nbands; }
  S->row_lo = 64 * S->band_lo; S->row_hi = 64 * S->band_hi < S->Y ? 64 * S->band_hi : S->Y;
  S->win_lo = S->row_lo - EU_GHOST_LO > 0 ? S->row_lo - EU_GHOST_LO : 0;
  S->win_hi = S->row_hi + EU_GHOST_HI < S->Y ? S->row_hi + EU_GHOST_HI : S->Y;
  S->win_off = (size_t)S->win_lo * S->X;
  S->Cw = (size_t)(S->win_hi - S->win_lo) * S->X;
  S->ab_lo = S->band_lo > 0 ? S->band_lo - 1 : 0; S->ab_hi = S->band_hi < nbands ? S->band_hi + 1 : nbands;
  S->skew_off = (size_t)S->ab_lo * S->geom.TS * 64;
  S->Sw = (size_t)(S->ab_hi - S->ab_lo) * S->geom.TS * 64;
  S->e_lo = (size_t)S->band_lo * S->geom.TS * 64; S->e_cnt = (size_t)(S->band_hi - S->band_lo) * S->geom.TS * 64;
  const size_t SS = S->Sw;
  const size_t Cw = S->Cw;
  // supported maximum per handle = BASELINE's largest configuration, 16384^2 cells; marker indices (4 per cell, global) are 32-bit
  if (C > ((size_t)1 << 29) || Cw > ((size_t)1 << 28) || (SS + EU_SKEW_SLACK) * 8 >= ((size_t)1 << 32)) {
    eu_set_error("grid %d x %d is larger than the supported maximum (2^28 cells per GPU, 2^29 in all)", cfg->X, cfg->Y);
    free(S);
    return EULER_EINVAL;
  }
  // from here on every failure releases what has been created so far (euler_destroy copes with a half-built handle)
#define CREATECHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { int _rc = eu_hip_fail(_e, #call, __FILE__, __LINE__); euler_destroy(S); return _rc; } } while (0)
  CREATECHK(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
  S->own_stream = 1;

  // (allocated un-shifted; all base pointers are shifted together once every allocation has succeeded: S->shifted)
  DALLOC(S->u, Cw); DALLOC(S->v, Cw); DALLOC(S->utmp, Cw); DALLOC(S->vtmp, Cw);
  DALLOC(S->solid, Cw); DALLOC(S->source, Cw); DALLOC(S->sink, Cw); DALLOC(S->count, Cw); DALLOC(S->prev_count, Cw);
  DALLOC(S->count32, Cw);
  if (!S->slab_on) { DALLOC(S->blockedT, Cw); DALLOC(S->uT, Cw); DALLOC(S->vT, Cw); DALLOC(S->countT, Cw); DALLOC(S->solidT, Cw); }      // (the marker stage's column-major copies)
  eu_state_replaced(S);
  DALLOC(S->sys_m, Cw); DALLOC(S->sys_div, Cw);
  if (S->cfg.rainbow) for (float*& d : S->dye) DALLOC(d, Cw);      // (the window like every row-major field; the whole grid without slabs)
  // MAX_MARKER_COUNT = 4 X Y (main.c:92) is the GLOBAL cap; a slab holds the markers inside its rows: room for 6 per owned cell
  S->max_markers = 4 * C;
  if (S->slab_on) { const size_t cap = 6 * (size_t)(S->row_hi - S->row_lo) * S->X + 65536; if (cap < S->max_markers) S->max_markers = cap; }
  DALLOC(S->markers[0], S->max_markers); DALLOC(S->markers[1], S->max_markers);
  if (S->slab_on) { DALLOC(S->keys[0], S->max_markers); DALLOC(S->keys[1], S->max_markers); }
  DALLOC(S->ms, 1);
  const size_t mwords = (S->max_markers + 63) / 64;
  DALLOC(S->evmask, mwords);
  if (!S->slab_on) DALLOC(S->delmask, mwords);      // (the advection pass that bins as well keeps the delete ballot apart from the collision ballot)
  if (!S->slab_on) {      // the tile map (euler_dev.h): flags of the count grid, then of the previous one, a ring of one tile around each (the passes look right of and above a tile without asking where the grid ends)
    S->tmap_nx = (S->X + 63) / 64 + 1; S->tmap_n = S->tmap_nx * ((S->Y + 63) / 64 + 2);
    DALLOC(S->tmap, 3 * (size_t)S->tmap_n);      // (+ "a marker entered the tile since the last refresh": k_markers.hip mk_touch)
  }
  DALLOC(S->ev_theta, S->max_markers); DALLOC(S->ev_delta, S->max_markers);
  S->sel_cap = S->max_markers;
  DALLOC(S->sel_idx, S->sel_cap);
  DALLOC(S->act_idx, S->max_markers); DALLOC(S->act_dt, S->max_markers);
  DALLOC(S->cellmask64, (Cw + 63) / 64 + 1);
  S->sel.capacity_blocks = (mwords + 2047) / 2048 + 1;
  DALLOC(S->sel.block_sums, S->sel.capacity_blocks);
  // skewed arrays carry EU_SKEW_SLACK zeroed elements in front (the backward sweep prefetches below record 0)
  // Every array starts EU_ARRAY_STAGGER bytes further into its allocation than the one before: eight arrays of the same size, allocated back to
  // back, otherwise sit at the same offset modulo every power of two, and the four to six streams a PCG pass reads and writes at the same
  // element index land in the same HBM channel at the same time (EU_ARRAY_STAGGER, bytes; measured in rounds 2-4: 0 .. 1 MB make no difference on this chip)
  {
    const size_t stagger = (size_t)EU_ARRAY_STAGGER / 8 * 8;
    int k = 0;
    for (double** d : {&S->b, &S->p, &S->r, &S->z, &S->s, &S->s2, &S->q, &S->precon}) {
      DALLOC(*d, SS + EU_SKEW_SLACK + 8 * stagger / 8);
      S->skew_alloc[k] = *d;
      *d += (size_t)k * stagger / 8;
      ++k;
    }
    DALLOC(S->cellmask, SS + EU_SKEW_SLACK);
    S->skew_alloc[8] = S->cellmask;
  }
  S->fb_stride = 12 * (((S->geom.T + 7) / 8 + 11) / 12) + 4;   // whole groups of 3 and of 4 blocks + the blocks the prefetch runs ahead
  DALLOC(S->fbits_fwd, (size_t)(S->ab_hi - S->ab_lo) * S->fb_stride * 64);
  DALLOC(S->fbits_bwd, (size_t)(S->ab_hi - S->ab_lo) * S->fb_stride * 64);
  DALLOC(S->sc, 1);
  DALLOC(S->band_ranges, (size_t)S->geom.nbands);
  { int rc = eu_set_tiles(S, S->cfg.precond_tile_records); if (rc) { euler_destroy(S); return rc; } }
  S->red_blocks = (int)eu_blocks(S->e_cnt, EU_RED_ELEMS, 2048);
  DALLOC(S->partial, (size_t)S->red_blocks > 2048 ? (size_t)S->red_blocks : 2048);
  DALLOC(S->partial2, 2048);
  DALLOC(S->pair_buf, 2 * 64);
  DALLOC(S->alpha_buf, 64);
  S->xrow_len = ((size_t)S->X + 15) / 16 * 16 + 64;
  DALLOC(S->xrows, 8 * S->xrow_len);
  S->chunk_cap = (size_t)(S->band_hi - S->band_lo) * (S->geom.T / 16);
  S->chunk_words = (S->chunk_cap + 63) / 64;
  DALLOC(S->chunk_flag, S->chunk_cap + 64);
  DALLOC(S->chunk_prev, S->chunk_cap + 64);
  DALLOC(S->chunk_part, S->chunk_cap + 64);
  DALLOC(S->tile_table, 8 * 64 * 2);
  DALLOC(S->chunk_bits, S->chunk_words + 1);
  DALLOC(S->chunk_list, S->chunk_cap + 64);
  {   // xorshift64* jump-ahead: M^(2^i) as the images of the 64 basis vectors, i < EU_RNG_JUMPS (euler_dev.h)
    RngJump* J = (RngJump*)malloc(sizeof(RngJump));
    if (!J) { euler_destroy(S); return EULER_ENOMEM; }
    auto step = [](unsigned long long st) { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st; };
    for (int b = 0; b < 64; ++b) J->col[0][b] = step(1ull << b);
    for (int i = 1; i < EU_RNG_JUMPS; ++i)
      for (int b = 0; b < 64; ++b) {       // M^(2^i) e_b = M^(2^(i-1)) (M^(2^(i-1)) e_b)
        unsigned long long x = J->col[i - 1][b], y = 0;
        for (int k = 0; k < 64; ++k) if ((x >> k) & 1) y ^= J->col[i - 1][k];
        J->col[i][b] = y;
      }
    int rc = dalloc(&S->rng_jump, 1);
    if (!rc && hipMemcpy(S->rng_jump, J, sizeof(RngJump), hipMemcpyHostToDevice) != hipSuccess) rc = EULER_EHIP;
    free(J);
    if (rc) { euler_destroy(S); return rc; }
  }
  DALLOC(S->red_counter, 1);
  DALLOC(S->halo_buf, (size_t)8 * S->X);   // 4 buffers (send / recv, below / above) of up to two grid rows of doubles
  S->gran_stride = (S->geom.T - 63 + 7) / 8 * 8;   // hand-off columns [0, T - 63)
  DALLOC(S->granules, (size_t)S->geom.nbands * S->gran_stride * 2);
  DALLOC(S->ticket, 1);
  DALLOC(S->sweep_timeline, (size_t)S->geom.nbands * 8);
  S->ticket_base = 0; S->epoch = 0;
  // every device allocation is in place: shift the base pointers to global indexing (euler_dev.h "Row slabs")
  for (float** f : {&S->u, &S->v, &S->utmp, &S->vtmp}) *f -= S->win_off;
  for (float*& d : S->dye) if (d) d -= S->win_off;
  for (uint8_t** g : {&S->solid, &S->source, &S->sink, &S->count, &S->prev_count}) *g -= S->win_off;
  for (double** d : {&S->b, &S->p, &S->r, &S->z, &S->s, &S->s2, &S->q, &S->precon}) *d += EU_SKEW_SLACK - S->skew_off;
  S->cellmask += EU_SKEW_SLACK - S->skew_off;
  S->fbits_fwd -= (size_t)S->ab_lo * S->fb_stride * 64; S->fbits_bwd -= (size_t)S->ab_lo * S->fb_stride * 64;
  S->shifted = 1;
  CREATECHK(hipHostMalloc((void**)&S->ms_host, sizeof(MarkerState), hipHostMallocDefault));
  CREATECHK(hipHostMalloc((void**)&S->sc_host, sizeof(PcgScalars), hipHostMallocDefault));
  CREATECHK(hipHostMalloc((void**)&S->poll_host, 2 * sizeof(PcgScalars), hipHostMallocDefault));
  CREATECHK(hipHostMalloc((void**)&S->res_err, sizeof(int), hipHostMallocDefault));      // (the resident solver raises it from the device: k_resident.hip)
  *S->res_err = 0;
  DALLOC(S->res_gran, 2 * 2 * (2 * 768) * 2);      // (k_resident.hip: [2 generations][2 groups][2 RS_MAX_WG granules] x 2 words)
  S->res_tag = 1;
  memset(S->poll_host, 0, 2 * sizeof(PcgScalars));
  for (hipEvent_t& e : S->poll_event) CREATECHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  memset(S->ms_host, 0, sizeof(MarkerState)); memset(S->sc_host, 0, sizeof(PcgScalars));

  S->interp_lim[0] = nextafterf((float)(S->X - 2), 0.f);   // U extent (X-1, Y): size-1
  S->interp_lim[1] = nextafterf((float)(S->Y - 1), 0.f);
  S->interp_lim[2] = nextafterf((float)(S->X - 1), 0.f);   // V extent (X, Y-1)
  S->interp_lim[3] = nextafterf((float)(S->Y - 2), 0.f);

  S->ev_cap = 16384;
  S->ev_pool = (hipEvent_t*)calloc((size_t)S->ev_cap, sizeof(hipEvent_t));
  S->ev_cls = (int*)calloc((size_t)S->ev_cap / 2, sizeof(int));
  S->ev_solve = (int*)calloc((size_t)S->ev_cap / 2, sizeof(int));
  S->ev_iter = (int*)calloc((size_t)S->ev_cap / 2, sizeof(int));
  S->prof_iter = -2;
  if (!S->ev_pool || !S->ev_cls || !S->ev_solve || !S->ev_iter) { euler_destroy(S); return EULER_ENOMEM; }
  for (int k = 0; k < S->ev_cap; ++k) CREATECHK(hipEventCreate(&S->ev_pool[k]));
  if (S->slab_on) { int rc = eu_slab_alloc(S); if (rc) { euler_destroy(S); return rc; } }
  if (S->cfg.precond == EULER_PRECOND_IC0_TILE2 || S->cfg.precond == EULER_PRECOND_IC0_TILE_MG) {
    if ((S->slab_on && S->cfg.precond != EULER_PRECOND_IC0_TILE_MG) || S->tile_w != 16 || S->cfg.sweep_mode == EULER_SWEEP_SIMPLE) {
      eu_set_error("EULER_PRECOND_IC0_TILE2: one GPU; _MG: one GPU or row slabs; both: tiles of 16 records, the band schedule"); euler_destroy(S); return EULER_EINVAL;
    }
    S->cfg.dot_mode = EULER_DOT_TREE;
    int rc = eu_coarse_alloc(S, S->cfg.precond == EULER_PRECOND_IC0_TILE_MG);
    if (rc) { euler_destroy(S); return rc; }
  }
  S->hbm_bytes = g_alloc_bytes + (S->slab_on ? eu_slab_bytes(S) : 0);
  eu_launch_tile_table(S);      // E^-1 of an interior tile (k_pcg.hip), once per handle
  if (S->cfg.pcg_precision != EULER_PCG_F64 && S->cfg.pcg_precision != EULER_PCG_F32) { eu_set_error("euler_create: pcg_precision %d", S->cfg.pcg_precision); euler_destroy(S); return EULER_EINVAL; }
  if (S->cfg.pcg_precision == EULER_PCG_F32 && !eu_resident_eligible(S)) {
    eu_set_error("euler_create: EULER_PCG_F32 runs in the resident solver: one GPU, EULER_PRECOND_IC0_TILE with tiles of 16 records, EULER_DOT_TREE, "
                 "EULER_RESIDENT_AUTO, and a grid of at most %d 16-record chunks (this one: %d)", 4 * eu_resident_capacity(S, 1), S->geom.nbands * (S->geom.T / 16));
    euler_destroy(S); return EULER_EINVAL;
  }
  CREATECHK(hipStreamSynchronize(S->stream));
#undef CREATECHK
  *out = S;
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// scenario upload = the device side of sim_init (main.c:209-274)
static int upload_scenario(euler_sim* S, const uint8_t* solid, const uint8_t* source, const uint8_t* sink,
                           const uint8_t* fluid) {
  const size_t C = S->C;
  if (S->slab_on && !S->has_comm) { eu_set_error("row-slab handle: install the communicator (euler_set_comm / euler_set_comm_rccl) before loading a scenario"); return EULER_ESTATE; }
  // The marker array comes from ONE sequential RNG stream (main.c:255-266).  A row-slab handle walks the whole stream and keeps
  // the markers of its own rows with their positions in the array (keys); nobody stores the whole array.
  std::vector<float> mk;
  std::vector<unsigned int> keys;
  uint64_t rng = EULER_RNG_SEED, n = 0, n_loc = 0;
  int rc;
  if (S->slab_on) {
    // room: a cell's four markers land in its own row (or, when a jitter rounds up to the cell's edge, the next one)
    uint64_t cells = 0;
    for (int y = S->row_lo > 0 ? S->row_lo - 1 : 0; y < S->row_hi; ++y)
      for (int x = 0; x < S->X; ++x) cells += fluid[(size_t)y * S->X + x] != 0;
    const uint64_t room = 4 * cells;
    if (room > S->max_markers) { eu_set_error("row slab %d holds up to %llu markers, more than its capacity %zu", S->cfg.slab_rank, (unsigned long long)room, S->max_markers); return EULER_ENOMEM; }
    try { mk.resize(2 * (size_t)room + 2); keys.resize((size_t)room + 1); } catch (...) { return EULER_ENOMEM; }
    rc = euler_seed_markers_rows(fluid, S->X, S->Y, S->row_lo, S->row_hi, &rng, mk.data(), keys.data(), room, &n, &n_loc);   // one walk over the stream
  } else {
    try { mk.resize(2 * 4 * C); } catch (...) { return EULER_ENOMEM; }
    rc = euler_seed_markers(fluid, S->X, S->Y, &rng, mk.data(), &n);
    n_loc = n;
  }
  if (rc) return rc;
  size_t nsrc = 0;
  for (size_t i = (size_t)S->row_lo * S->X; i < (size_t)S->row_hi * S->X; ++i) nsrc += source[i] != 0;
  S->n_source_cells = nsrc;
  if (S->draws) { (void)hipFree(S->draws); S->draws = nullptr; }
  if (nsrc) HIPCHK(hipMalloc((void**)&S->draws, 2 * nsrc * sizeof(float)));

  hipStream_t st = S->stream;
  const size_t wo = S->win_off, Cw = S->Cw;            // the window's rows of the static grids
  HIPCHK(hipMemcpyAsync(S->solid + wo, solid + wo, Cw, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S->source + wo, source + wo, Cw, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(S->sink + wo, sink + wo, Cw, hipMemcpyHostToDevice, st));
  eu_state_replaced(S);
  for (float* f : {S->u, S->v, S->utmp, S->vtmp}) HIPCHK(hipMemsetAsync(f + wo, 0, Cw * sizeof(float), st));
  for (double* d : {S->b, S->p, S->r, S->z, S->s, S->s2, S->q, S->precon}) HIPCHK(hipMemsetAsync(d + S->skew_off, 0, S->Sw * sizeof(double), st));
  for (float* d : S->dye) if (d) HIPCHK(hipMemsetAsync(d + wo, 0, Cw * sizeof(float), st));
  HIPCHK(hipMemsetAsync(S->count + wo, 0, Cw, st));
  HIPCHK(hipMemsetAsync(S->prev_count + wo, 0, Cw, st));
  HIPCHK(hipMemsetAsync(S->cellmask + S->skew_off, 0, S->Sw, st));
  S->cur = 0;
  HIPCHK(hipMemcpyAsync(S->markers[0], mk.data(), n_loc * sizeof(float2), hipMemcpyHostToDevice, st));
  if (S->slab_on && n_loc) HIPCHK(hipMemcpyAsync(S->keys[0], keys.data(), n_loc * sizeof(unsigned int), hipMemcpyHostToDevice, st));
  MarkerState m0;
  memset(&m0, 0, sizeof(m0));
  m0.n = n; m0.n_loc = n_loc; m0.max_markers = 4 * C; m0.rng_state = rng;     // n and the cap are the reference's GLOBAL ones
  HIPCHK(hipMemcpyAsync(S->ms, &m0, sizeof(m0), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  S->n_markers_host = n_loc;
  // sim_init ends with refresh_marker_counts() (main.c:268): prev <- 0, counts <- bins
  rc = S->slab_on ? eu_slab_after_load(S) : eu_launch_refresh_counts(S);
  if (rc) return rc;
  rc = eu_launch_colorize(S);   // main.c:270-273 (only with cfg.rainbow)
  if (rc) return rc;
  if (S->slab_on && S->dye[0] && (rc = eu_slab_exchange_dye(S))) return rc;
  rc = eu_sync_marker_state(S);
  if (rc) return rc;
  memset(&S->stats, 0, sizeof(S->stats));
  S->loaded = 1;
  return EULER_OK;
}

static int load_from_grids(euler_sim* S, int (*fill)(void*, uint8_t*, uint8_t*, uint8_t*, uint8_t*), void* ctx) {
  const size_t C = S->C;
  uint8_t* g = (uint8_t*)malloc(4 * C);
  if (!g) return EULER_ENOMEM;
  int rc = fill(ctx, g, g + C, g + 2 * C, g + 3 * C);
  if (!rc) rc = upload_scenario(S, g, g + C, g + 2 * C, g + 3 * C);
  free(g);
  return rc;
}

struct TextCtx { const char* text; int len; int upscale; int X, Y; };
static int fill_text(void* c, uint8_t* so, uint8_t* sr, uint8_t* si, uint8_t* fl) {
  TextCtx* t = (TextCtx*)c;
  return euler_parse_scenario(t->text, t->len, t->X, t->Y, t->upscale, so, sr, si, fl);
}
struct TankCtx { euler_sim* S; int tanks; };
static int fill_tank(void* c, uint8_t* so, uint8_t* sr, uint8_t* si, uint8_t* fl) {
  TankCtx* t = (TankCtx*)c;
  int rc = euler_half_tanks_grids(t->S->X, t->S->Y, t->tanks, so, sr, si, fl);
  if (rc) eu_set_error("half tanks: %d tanks do not fit a %d x %d grid (Y must be a multiple, each tank at least 6 rows)", t->tanks, t->S->X, t->S->Y);
  return rc;
}

extern "C" int euler_load_scenario_mem(euler_sim* S, const char* text, int32_t len, int32_t upscale) {
  if (!S || !text || len < 0) { eu_set_error("euler_load_scenario_mem: bad argument"); return EULER_EINVAL; }
  TextCtx t{text, len, upscale, S->X, S->Y};
  return load_from_grids(S, fill_text, &t);
}

extern "C" int euler_load_scenario_file(euler_sim* S, const char* path, int32_t upscale) {
  if (!S || !path) return EULER_EINVAL;
  FILE* f = fopen(path, "rb");
  if (!f) { eu_set_error("Could not load %s!", path); return EULER_EIO; }   // main.c:213
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)n + 1);
  if (!buf || (n > 0 && fread(buf, (size_t)n, 1, f) != 1)) { fclose(f); free(buf); eu_set_error("Could not load %s!", path); return EULER_EIO; }
  fclose(f);
  int rc = euler_load_scenario_mem(S, buf, (int32_t)n, upscale);
  free(buf);
  return rc;
}

extern "C" int euler_load_half_tanks(euler_sim* S, int32_t tanks) {
  if (!S) return EULER_EINVAL;
  TankCtx t{S, tanks};
  return load_from_grids(S, fill_tank, &t);
}
extern "C" int euler_load_half_tank(euler_sim* S) { return euler_load_half_tanks(S, 1); }

// ------------------------------------------------------------------------------------------
// multi-GPU plumbing
extern "C" int euler_set_stream(euler_sim* S, void* hip_stream) {
  if (!S) return EULER_EINVAL;
  HIPCHK(hipStreamSynchronize(S->stream));
  if (S->own_stream) (void)hipStreamDestroy(S->stream);
  S->stream = (hipStream_t)hip_stream;
  S->own_stream = 0;
  return EULER_OK;
}

// where the reductions' FIN_TO_COMM epilogue leaves this rank's value for the small all-gather (k_pcg.hip "ghost rows")
static int eu_set_comm_slot(euler_sim* S) {
  if (S->comm.nranks > 64) { eu_set_error("euler_set_comm: at most 64 ranks"); return EULER_EINVAL; }
  double* slot = S->alpha_buf + S->comm.rank;
  HIPCHK(hipMemcpyAsync(&S->sc->comm_slot, &slot, sizeof slot, hipMemcpyHostToDevice, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return EULER_OK;
}

// allow_single: keep the communicator code path with one rank (the RCCL self-test on a 1-GPU box)
int eu_install_comm(euler_sim* S, const euler_comm_ops* ops, int32_t coupling, int allow_single) {
  if (!S) return EULER_EINVAL;
  if (ops && ops->nranks >= 1 && !(ops->nranks == 1 && !allow_single && !S->slab_on))
    if (const char* why = eu_precond_combination(S, S->cfg.precond, S->tile_w, 1, S->slab_on, 0)) { eu_set_error("euler_set_comm: %s", why); return EULER_EINVAL; }
  if (S->slab_on) {   // the handle IS one slab: the communicator must be the one it was created for; the partition stays
    if (!ops) { S->has_comm = 0; return EULER_OK; }
    if (!ops->allreduce || !ops->halo || !ops->chain || !ops->allgather || ops->rank != S->cfg.slab_rank || ops->nranks != S->cfg.slab_nranks) {
      eu_set_error("euler_set_comm: this handle is slab %d of %d; the communicator says %d of %d", S->cfg.slab_rank, S->cfg.slab_nranks,
                   ops->rank, ops->nranks);
      return EULER_EINVAL;
    }
    if (coupling == EULER_SLAB_EXACT && S->cfg.precond == EULER_PRECOND_IC0) {
      eu_set_error("euler_set_comm: row-slab handles run the tile-local preconditioner or slab-local IC(0) (the exact band pipeline "
                   "needs the neighbours' rows)");
      return EULER_EINVAL;
    }
    S->comm = *ops; S->bulk = *ops; S->has_comm = 1; S->couple = 0;
    { int rc = eu_set_comm_slot(S); if (rc) return rc; }
    return eu_slab_check_partition(S);      // the ranks' band ranges tile the grid (explicit partitions: euler_config.slab_band_lo / hi)
  }
  if (!ops || ops->nranks < 1 || (ops->nranks == 1 && !allow_single)) {
    S->has_comm = 0; S->band_lo = 0; S->band_hi = S->geom.nbands; S->e_lo = 0; S->e_cnt = S->geom.S;
    return EULER_OK;
  }
  if (!ops->allreduce || !ops->halo || !ops->chain || !ops->allgather || ops->rank < 0 || ops->rank >= ops->nranks) {
    eu_set_error("euler_set_comm: incomplete communicator"); return EULER_EINVAL;
  }
  const int nb = S->geom.nbands;
  if (ops->nranks > nb) { eu_set_error("euler_set_comm: %d ranks for %d bands (64 rows each)", ops->nranks, nb); return EULER_EINVAL; }
  if (S->cfg.sweep_mode == EULER_SWEEP_SIMPLE) { eu_set_error("euler_set_comm: EULER_SWEEP_SIMPLE is single-rank only"); return EULER_EINVAL; }
  S->comm = *ops; S->bulk = *ops;
  S->has_comm = 1;
  S->couple = coupling == EULER_SLAB_EXACT;
  S->band_lo = (int)((int64_t)nb * ops->rank / ops->nranks);
  S->band_hi = (int)((int64_t)nb * (ops->rank + 1) / ops->nranks);
  S->e_lo = (size_t)S->band_lo * S->geom.TS * 64;
  S->e_cnt = (size_t)(S->band_hi - S->band_lo) * S->geom.TS * 64;
  if (S->cfg.dot_mode == EULER_DOT_SEQUENTIAL) S->cfg.dot_mode = EULER_DOT_TREE;   // the replay order is a 1-rank notion
  return eu_set_comm_slot(S);
}

extern "C" int euler_set_comm(euler_sim* S, const euler_comm_ops* ops, int32_t coupling) {
  if (!S) return EULER_EINVAL;
  HIPCHK(hipStreamSynchronize(S->stream));
  eu_p2p_release(S);
  eu_rccl_release(S);   // a caller-supplied communicator replaces the built-in one
  return eu_install_comm(S, ops, coupling, 0);
}

extern "C" int euler_slab_info(euler_sim* S, int32_t* lo, int32_t* hi, int32_t* nb) {
  if (!S) return EULER_EINVAL;
  if (lo) *lo = S->band_lo;
  if (hi) *hi = S->band_hi;
  if (nb) *nb = S->geom.nbands;
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// stepping
int eu_sync_marker_state(euler_sim* S) {
  HIPCHK(hipMemcpyAsync(S->ms_host, S->ms, sizeof(MarkerState), hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  S->n_markers_host = S->slab_on ? S->ms_host->n_loc : S->ms_host->n;
  if (S->ms_host->error) {
    if (S->ms_host->error >= 16) { eu_set_error("row slabs (seen on rank %d): exchange buffer overflow on some rank (code %d: 16 dt-chain candidates, 17 migration, 18 deletions, 19 marker capacity)", S->cfg.slab_rank, S->ms_host->error); return EULER_ESTATE; }
    eu_set_error("device-side bounded wait expired (band pipeline), error=%d", S->ms_host->error);
    return EULER_ETIMEOUT;
  }
  return EULER_OK;
}

extern "C" int euler_timestep(euler_sim* S, float frame_time_left, float* dt) {
  if (!S || !S->loaded) { eu_set_error("euler_timestep: no scenario loaded"); return EULER_ESTATE; }
  int rc = eu_launch_timestep(S, frame_time_left);
  if (rc) return rc;
  rc = eu_sync_marker_state(S);   // one host round trip per substep: dt, marker count, error flag
  if (rc) return rc;
  if (dt) *dt = S->ms_host->dt;
  return EULER_OK;
}

// somebody other than the substep's own sequence is about to write the state: what one stage prepared for the next no longer describes it
static void eu_state_edited(euler_sim* S) {
  S->prebin_valid = 0;                               // (k_advect_bin_a2's counts and delete ballot)
  if (S->maxsq_state == 2) S->maxsq_state = 1;       // (k_velocity_update_para's maxima of u, v: stale, cleared before the next accumulation)
  S->uv_clean = 0; S->uv_zb = 0; S->tmap_valid = 0; S->utmp_clean = 0; S->countT_clean = 0;                     // (what the lean zero_bounds / the velocity update's skipped zero stores rely on)
}
static int run_stage(euler_sim* S, int stage, float dt) {
  if (stage != EULER_STAGE_REFRESH_COUNTS) S->prebin_valid = 0;      // (what k_advect_bin_a2 binned belongs to the refresh that follows it DIRECTLY)
  if (stage == EULER_STAGE_EXTRAPOLATE && S->maxsq_state == 2) S->maxsq_state = 1;      // (writes u, v; inside a substep the timestep has consumed the maxima long before)
  if (stage == EULER_STAGE_REFRESH_COUNTS) { S->uv_clean = S->uv_clean == 1 ? 2 : 0; S->uv_zb = 0; }      // (prev <- cur: once is what k_zero_bounds4<true> expects)
  if (stage == EULER_STAGE_REFRESH_COUNTS) { S->utmp_clean = S->utmp_clean == 1 ? 2 : 0; S->countT_clean = S->countT_clean == 1 ? 2 : 0; }      // (the same for what the tile map's readers left)
  if (stage == EULER_STAGE_SOURCES) S->uv_zb = 0;                                                        // (the count grid changes)
  switch (stage) {
    case EULER_STAGE_ADVECT_MARKERS: return eu_launch_advect_markers(S, dt);
    case EULER_STAGE_REFRESH_COUNTS: return eu_launch_refresh_counts(S);
    case EULER_STAGE_SOURCES: {   // with the dye: extrapolate(g_r/g/b, P) comes first (main.c:859-864)
      int rc = eu_launch_dye_extrapolate(S);
      if (!rc) rc = eu_launch_sources(S);
      if (!rc) rc = eu_launch_dye_sources(S);
      return rc;
    }
    case EULER_STAGE_EXTRAPOLATE: return eu_launch_extrapolate(S);
    case EULER_STAGE_ADVECT_VELOCITY: {   // advect_p reads g_u, g_v before anything overwrites them (main.c:871-882)
      int rc = eu_launch_dye_advect(S, dt);
      if (!rc) rc = eu_launch_advect_velocity(S, dt);
      return rc;
    }
    case EULER_STAGE_PROJECT: return eu_launch_project(S, dt);
    default: eu_set_error("unknown stage %d", stage); return EULER_EINVAL;
  }
}

extern "C" int euler_stage(euler_sim* S, int32_t stage, float dt) {
  if (!S || !S->loaded) { eu_set_error("euler_stage: no scenario loaded"); return EULER_ESTATE; }
  int rc = S->slab_on ? eu_slab_stage(S, stage, dt) : run_stage(S, stage, dt);      // (row slabs: collective - the stage and the exchanges that belong to it)
  if (rc) return rc;
  if (S->slab_on && (rc = eu_slab_error_sync(S))) return rc;
  rc = eu_sync_marker_state(S);
  if (!rc && stage == EULER_STAGE_PROJECT) {      // the solve's outcome (not the totals: a stage is not a substep)
    S->stats.last_pcg_iterations = S->sc_host->nonzero ? S->sc_host->iters : 0;
    S->stats.last_residual = S->sc_host->rnorm;
  }
  return rc;
}

static int substep_async(euler_sim* S, float dt) {
  if (S->slab_on) return eu_slab_substep(S, dt);       // the same stages over this rank's rows, with their exchanges in between
  for (int st = 0; st < EULER_STAGE__COUNT; ++st) {
    int rc = run_stage(S, st, dt);
    if (rc) return rc;
  }
  return EULER_OK;
}

static void account_substep(euler_sim* S, float dt) {
  // sc_host was copied at the end of eu_launch_project and is valid after the next sync
  S->stats.total_substeps += 1;
  S->stats.last_dt = dt;
}

extern "C" int euler_substep(euler_sim* S, float dt) {
  if (!S || !S->loaded) { eu_set_error("euler_substep: no scenario loaded"); return EULER_ESTATE; }
  int rc = substep_async(S, dt);
  if (rc) return rc;
  if (S->slab_on && (rc = eu_slab_error_sync(S))) return rc;
  rc = eu_sync_marker_state(S);
  if (rc) return rc;
  account_substep(S, dt);
  S->stats.total_pcg_iterations += (uint64_t)S->sc_host->iters;
  S->stats.last_pcg_iterations = S->sc_host->iters;
  S->stats.last_residual = S->sc_host->rnorm;
  return EULER_OK;
}

// sim_step (main.c:843-900): split the 0.1 s frame into <= 8 CFL-limited substeps
extern "C" int euler_step(euler_sim* S) {
  if (!S || !S->loaded) { eu_set_error("euler_step: no scenario loaded"); return EULER_ESTATE; }
  float frame_time = S->cfg.frame_time;
  int nsub = 0, iters = 0;
  for (int step = 0; frame_time > 0.f && step < S->cfg.max_substeps; ++step) {
    float dt = 0.f;
    int rc = euler_timestep(S, frame_time, &dt);   // syncs: also finalises the previous substep's scalars
    if (rc) return rc;
    if (step > 0) { iters += S->sc_host->iters; S->stats.last_residual = S->sc_host->rnorm; }
    frame_time -= dt;
    rc = substep_async(S, dt);
    if (rc) return rc;
    account_substep(S, dt);
    ++nsub;
  }
  int rc = S->slab_on ? eu_slab_error_sync(S) : EULER_OK;      // an overflow on one rank fails the frame on every rank
  if (rc) return rc;
  rc = eu_sync_marker_state(S);
  if (rc) return rc;
  if (nsub) { iters += S->sc_host->iters; S->stats.last_residual = S->sc_host->rnorm; }
  S->stats.total_pcg_iterations += (uint64_t)iters;
  S->stats.last_pcg_iterations = iters;
  S->stats.last_substeps = nsub;
  S->stats.frames += 1;
  return EULER_OK;
}

extern "C" int euler_pcg_op(euler_sim* S, int32_t op, float dt, double a, double* out) {
  if (!S || !S->loaded) { eu_set_error("euler_pcg_op: no scenario loaded"); return EULER_ESTATE; }
  // (the single building blocks address the whole grid; a row-slab handle holds a window of it)
  if (S->slab_on) { eu_set_error("euler_pcg_op: single PCG operations are not exposed on a row-slab handle"); return EULER_ESTATE; }
  S->pcg_fields_resident = 0;      // (single operations run the multi-kernel path: z, s, q are in memory again)
  int rcp = eu_pressure_current(S);
  if (rcp) return rcp;
  return eu_launch_pcg_op(S, op, dt, a, out);
}

// ------------------------------------------------------------------------------------------
// state access
// A field as the caller sees it: this handle's OWN rows [row_lo, row_hi) (the whole grid without slabs).
static int field_ptr(euler_sim* S, int f, void** p, size_t* bytes) {
  const size_t C = (size_t)(S->row_hi - S->row_lo) * S->X, o = (size_t)S->row_lo * S->X;
  switch (f) {
    case EULER_F_U: *p = S->u + o; *bytes = C * 4; break;
    case EULER_F_V: *p = S->v + o; *bytes = C * 4; break;
    case EULER_F_UTMP: *p = S->utmp + o; *bytes = C * 4; break;
    case EULER_F_VTMP: *p = S->vtmp + o; *bytes = C * 4; break;
    case EULER_F_SOLID: *p = S->solid + o; *bytes = C; break;
    case EULER_F_SOURCE: *p = S->source + o; *bytes = C; break;
    case EULER_F_SINK: *p = S->sink + o; *bytes = C; break;
    case EULER_F_COUNT: *p = S->count + o; *bytes = C; break;
    case EULER_F_PREV_COUNT: *p = S->prev_count + o; *bytes = C; break;
    case EULER_F_MARKERS: *p = S->markers[S->cur]; *bytes = (size_t)S->n_markers_host * 8; break;
    case EULER_F_MARKER_KEYS:
      if (!S->slab_on) { eu_set_error("EULER_F_MARKER_KEYS: row-slab handles only (elsewhere a marker's key is its index)"); return EULER_ESTATE; }
      *p = S->keys[S->cur]; *bytes = (size_t)S->n_markers_host * 4; break;
    case EULER_F_PRECON: *p = S->precon; *bytes = C * 8; break;
    case EULER_F_PRESSURE: *p = S->p; *bytes = C * 8; break;
    case EULER_F_PCG_B: *p = S->b; *bytes = C * 8; break;
    case EULER_F_PCG_R: *p = S->r; *bytes = C * 8; break;
    case EULER_F_PCG_Z: *p = S->z; *bytes = C * 8; break;
    case EULER_F_PCG_S: *p = eu_current_search_direction(S); *bytes = C * 8; break;
    case EULER_F_PCG_Q: *p = S->q; *bytes = C * 8; break;
    case EULER_F_CELLMASK: *p = S->cellmask; *bytes = C; break;
    case EULER_F_DYE_R: case EULER_F_DYE_G: case EULER_F_DYE_B: case EULER_F_DYE_RTMP: case EULER_F_DYE_GTMP: case EULER_F_DYE_BTMP:
      if (!S->dye[0]) { eu_set_error("field %d needs euler_config.rainbow", f); return EULER_ESTATE; }
      *p = S->dye[f - EULER_F_DYE_R] + o; *bytes = C * 4; break;
    default: eu_set_error("unknown field %d", f); return EULER_EINVAL;
  }
  return EULER_OK;
}

extern "C" size_t euler_field_bytes(const euler_sim* S, int32_t f) {
  void* p; size_t b = 0;
  if (!S || field_ptr((euler_sim*)S, f, &p, &b)) return 0;
  return b;
}

static bool field_is_skewed(int f) {
  return f == EULER_F_PRECON || f == EULER_F_PRESSURE || (f >= EULER_F_PCG_B && f <= EULER_F_PCG_Q) || f == EULER_F_CELLMASK;
}
static int ensure_rowmajor_tmp(euler_sim* S) {
  if (!S->rowmajor_tmp) HIPCHK(hipMalloc((void**)&S->rowmajor_tmp, (size_t)(S->row_hi - S->row_lo) * S->X * sizeof(double)));
  return EULER_OK;
}

extern "C" int euler_get_field(euler_sim* S, int32_t f, void* dst, size_t dst_bytes) {
  if (!S || !dst) return EULER_EINVAL;
  void* p; size_t b;
  int rc = field_ptr(S, f, &p, &b);
  if (rc) return rc;
  if (dst_bytes < b) { eu_set_error("euler_get_field(%d): buffer %zu < %zu bytes", f, dst_bytes, b); return EULER_EINVAL; }
  if (b == 0) return EULER_OK;
  if ((f == EULER_F_PCG_Z || f == EULER_F_PCG_S || f == EULER_F_PCG_Q) && S->pcg_fields_resident) {
    eu_set_error("euler_get_field(%d): the last solve ran in the resident kernel, which keeps z, s and A s in registers (euler_config.resident = EULER_RESIDENT_OFF shows them)", f);
    return EULER_ESTATE;
  }
  if (f == EULER_F_PRESSURE && (rc = eu_pressure_current(S))) return rc;      // (k_velocity_update_para leaves the last fmadds and the clamp to whoever looks)
  if (field_is_skewed(f)) {   // the solver's arrays are band-skewed in HBM: gather to row-major first
    rc = ensure_rowmajor_tmp(S);
    if (rc) return rc;
    if (f == EULER_F_PCG_S && S->s_none) HIPCHK(hipMemsetAsync(S->rowmajor_tmp, 0, b, S->stream));
    else if (f == EULER_F_PCG_S && S->s_stale) eu_unskew_fluid(S, (const double*)p, (double*)S->rowmajor_tmp);
    else eu_unskew(S, p, S->rowmajor_tmp, f == EULER_F_CELLMASK ? 1 : 8);
    p = S->rowmajor_tmp;
  }
  HIPCHK(hipMemcpyAsync(dst, p, b, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  return EULER_OK;
}

// the source cells of the own rows changed: the buffer of their random draws follows (k_source_fill)
int eu_set_source_count(euler_sim* S, size_t nsrc) {
  S->n_source_cells = nsrc;
  if (S->draws) { (void)hipFree(S->draws); S->draws = nullptr; }
  if (nsrc) HIPCHK(hipMalloc((void**)&S->draws, 2 * nsrc * sizeof(float)));
  return EULER_OK;
}

extern "C" int euler_set_field(euler_sim* S, int32_t f, const void* src, size_t src_bytes) {
  if (!S || !src) return EULER_EINVAL;
  if (field_is_skewed(f)) { int rcp = eu_pressure_current(S); if (rcp) return rcp; }      // (the pressure is finished from the solver's arrays before a caller overwrites any of them)
  eu_state_edited(S);
  if (S->slab_on && (f == EULER_F_MARKERS || f == EULER_F_MARKER_KEYS || f == EULER_F_COUNT || f == EULER_F_PREV_COUNT ||
                     f == EULER_F_SOLID || f == EULER_F_SOURCE)) {
    // (solid cells are read through ghost rows, the source cells of ALL ranks decide whether the source stage runs at all: facts
    // that only a scenario / snapshot load sets up; the sink grid is read on own rows only and may be edited)
    eu_set_error("euler_set_field(%d): a row-slab handle takes its markers, counts, solid and source cells from euler_load_scenario_* / euler_load_state only", f); return EULER_ESTATE;
  }
  if (f == EULER_F_MARKERS) return euler_set_markers(S, (const float*)src, src_bytes / 8);
  void* p; size_t b;
  int rc = field_ptr(S, f, &p, &b);
  if (rc) return rc;
  if (src_bytes != b) { eu_set_error("euler_set_field(%d): %zu bytes given, %zu expected", f, src_bytes, b); return EULER_EINVAL; }
  if (field_is_skewed(f)) {
    S->lean_ok = 0;      // a caller's values in the solver arrays: the next assembly writes them whole
    if (f == EULER_F_PCG_S) { S->s_stale = 0; S->s_launched = 0; S->s_none = 0; }      // (every element of S->s is the caller's now)
    rc = ensure_rowmajor_tmp(S);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(S->rowmajor_tmp, src, b, hipMemcpyHostToDevice, S->stream));
    eu_skew(S, S->rowmajor_tmp, p, f == EULER_F_CELLMASK ? 1 : 8);
  } else {
    HIPCHK(hipMemcpyAsync(p, src, b, hipMemcpyHostToDevice, S->stream));
  }
  HIPCHK(hipStreamSynchronize(S->stream));
  if (S->slab_on && (f == EULER_F_U || f == EULER_F_V)) {   // COLLECTIVE on a row-slab handle: the neighbours' ghost rows follow at once
    rc = eu_slab_exchange_uv(S);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(S->stream));
  }
  if (f == EULER_F_SOLID || f == EULER_F_SINK) { S->blocked_dirty = 1; S->solidT_dirty = 1; }      // (the marker stage's column-major copies follow at their next use)
  if (f == EULER_F_SOURCE) {
    const uint8_t* s = (const uint8_t*)src;
    size_t nsrc = 0;
    for (size_t i = 0; i < (size_t)(S->row_hi - S->row_lo) * S->X; ++i) nsrc += s[i] != 0;
    if ((rc = eu_set_source_count(S, nsrc))) return rc;
  }
  S->loaded = 1;
  return EULER_OK;
}

__global__ void k_set_marker_state(MarkerState* ms, unsigned long long n, unsigned long long maxm, int set_n,
                                   unsigned long long rng, int exhausted, int set_rng) {
  if (set_n) { ms->n = n; ms->max_markers = maxm; }
  if (set_rng) { ms->rng_state = rng; ms->exhausted = exhausted; }
}

extern "C" int euler_set_markers(euler_sim* S, const float* xy, uint64_t n) {
  if (!S || (!xy && n) || n > S->max_markers) return EULER_EINVAL;
  if (S->slab_on) { eu_set_error("euler_set_markers: not on a row-slab handle"); return EULER_ESTATE; }
  eu_state_edited(S);
  if (n) HIPCHK(hipMemcpyAsync(S->markers[S->cur], xy, n * 8, hipMemcpyHostToDevice, S->stream));
  hipLaunchKernelGGL(k_set_marker_state, dim3(1), dim3(1), 0, S->stream, S->ms, (unsigned long long)n,
                     (unsigned long long)S->max_markers, 1, 0ull, 0, 0);
  S->loaded = 1;
  return eu_sync_marker_state(S);
}

extern "C" int euler_set_rng(euler_sim* S, uint64_t rng, int32_t exhausted) {
  if (!S) return EULER_EINVAL;
  hipLaunchKernelGGL(k_set_marker_state, dim3(1), dim3(1), 0, S->stream, S->ms, 0ull, 0ull, 0, (unsigned long long)rng,
                     (int)exhausted, 1);
  return eu_sync_marker_state(S);
}

__global__ __launch_bounds__(256) void k_count_fluid(const uint8_t* __restrict__ count, size_t C, unsigned long long* out) {
  unsigned int c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (size_t)gridDim.x * blockDim.x) c += count[i] != 0;
  c = (unsigned int)eu_wave_sum((double)c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

extern "C" int euler_get_stats(euler_sim* S, euler_stats* out) {
  if (!S || !out) return EULER_EINVAL;
  int rc = eu_sync_marker_state(S);
  if (rc) return rc;
  // fluid-cell census reuses the (idle) select total word as a 64-bit scratch
  unsigned long long* scratch = (unsigned long long*)S->partial;
  HIPCHK(hipMemsetAsync(scratch, 0, 8, S->stream));
  // (a row-slab handle counts its own rows; n_markers below is the job's total either way)
  hipLaunchKernelGGL(k_count_fluid, dim3(eu_blocks((size_t)(S->row_hi - S->row_lo) * S->X, 256 * 8, 1024)), dim3(256), 0, S->stream,
                     S->count + (size_t)S->row_lo * S->X, (size_t)(S->row_hi - S->row_lo) * S->X, scratch);
  unsigned long long nf = 0;
  HIPCHK(hipMemcpyAsync(&nf, scratch, 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  S->stats.n_markers = S->ms_host->n;
  S->stats.source_exhausted = S->ms_host->exhausted;
  S->stats.rng_state = S->ms_host->rng_state;
  S->stats.marker_dt_events = S->ms_host->total_dt_events;
  S->stats.marker_multi_events = S->ms_host->multi_events;
  S->stats.fluid_cells = nf;
  *out = S->stats;
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// render: fetch only the visible window rows of the count grid (draw_rows reads y in
// [max(Y-1-wy,1), Y-2] and x in [1, min(X-2, wx)], main.c:917-920)
extern "C" int euler_render(euler_sim* S, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len) {
  if (!S || !len) return EULER_EINVAL;
  if (!S->loaded) return EULER_ESTATE;
  if (S->slab_on) return eu_slab_render(S, wx, wy, out, cap, len);      // collective: the visible rows are gathered from the ranks that own them
  const int X = S->X, Y = S->Y;
  int cutoff = Y - 1 - wy;
  if (cutoff < 1) cutoff = 1;
  const size_t C = S->C;
  const bool dye = S->dye[0] != nullptr;
  uint8_t* g = (uint8_t*)calloc(3, C);
  float* col = dye ? (float*)calloc(3 * C, sizeof(float)) : nullptr;
  if (!g || (dye && !col)) { free(g); free(col); return EULER_ENOMEM; }
  const size_t off = (size_t)cutoff * X, bytes = (size_t)(Y - 1 - cutoff) * X;
  hipError_t e = hipMemcpyAsync(g + off, S->solid + off, bytes, hipMemcpyDeviceToHost, S->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(g + C + off, S->sink + off, bytes, hipMemcpyDeviceToHost, S->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(g + 2 * C + off, S->count + off, bytes, hipMemcpyDeviceToHost, S->stream);
  for (int k = 0; k < 3 && dye && e == hipSuccess; ++k)
    e = hipMemcpyAsync(col + k * C + off, S->dye[k] + off, bytes * sizeof(float), hipMemcpyDeviceToHost, S->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(S->stream);
  int rc = e != hipSuccess ? eu_hip_fail(e, "render copy", __FILE__, __LINE__)
           : dye ? euler_render_grids_rgb(g, g + C, g + 2 * C, col, col + C, col + 2 * C, X, Y, wx, wy, out, cap, len)
                 : euler_render_grids(g, g + C, g + 2 * C, X, Y, wx, wy, out, cap, len);
  free(g);
  free(col);
  return rc;
}

// the 'r' key of the reference's main loop (main.c:970-973): colour the current fluid afresh
extern "C" int euler_colorize(euler_sim* S) {
  if (!S || !S->loaded) return EULER_ESTATE;
  if (!S->dye[0]) { eu_set_error("euler_colorize needs euler_config.rainbow"); return EULER_ESTATE; }
  int rc = eu_launch_colorize(S);
  if (rc) return rc;
  if (S->slab_on && (rc = eu_slab_exchange_dye(S))) return rc;      // (collective on a row-slab handle: the neighbours' ghost rows follow)
  HIPCHK(hipStreamSynchronize(S->stream));
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// measurement helpers
// the fastest plain copy of tools/micro/copy_bench on MI355X: 16 bytes per lane, 8 loads in flight per thread, non-temporal
// (streaming) loads and stores, a grid-stride of 65536 blocks
typedef float cp_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const cp_f4* __restrict__ src, cp_f4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    cp_f4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(&src[i + k * stride]);
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], &dst[i + k * stride]);
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

extern "C" int euler_measure_copy_bandwidth(euler_sim* S, size_t bytes, int32_t reps, double* gbps) {
  if (!S || !gbps || reps < 1) return EULER_EINVAL;
  bytes &= ~(size_t)15;
  cp_f4 *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc((void**)&a, bytes));
  if (hipMalloc((void**)&b, bytes) != hipSuccess) { (void)hipFree(a); return EULER_ENOMEM; }
  (void)hipMemsetAsync(a, 1, bytes, S->stream);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t n = bytes / 16;
  hipLaunchKernelGGL(k_copy16, dim3(65536), dim3(256), 0, S->stream, a, b, n);   // warm-up
  (void)hipEventRecord(e0, S->stream);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy16, dim3(65536), dim3(256), 0, S->stream, a, b, n);
  (void)hipEventRecord(e1, S->stream);
  hipError_t e = hipStreamSynchronize(S->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (e != hipSuccess) return eu_hip_fail(e, "copy probe", __FILE__, __LINE__);
  *gbps = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
  return EULER_OK;
}

// The latency of ONE exchange point of a distributed PCG iteration over the installed communicator - `row_doubles` doubles to / from each neighbour and `nsmall` doubles
// of every rank to every rank, euler_comm_ops.exchange (or halo + all-gather where the communicator has no fused operation) - measured with HIP events on the handle's stream
// over `reps` back-to-back calls.  Collective.  The model of DESIGN.md section 7 has this one unknown (L); tools/node_first_contact.sh measures it before anything else on a node.
extern "C" int euler_measure_exchange(euler_sim* S, int32_t reps, int32_t row_doubles, int32_t nsmall, double* us_per_exchange) {
  if (!S || !us_per_exchange || reps < 1 || row_doubles < 0 || nsmall < 0) return EULER_EINVAL;
  if (!S->has_comm) { eu_set_error("euler_measure_exchange: install a communicator first"); return EULER_ESTATE; }
  const int R = S->bulk.nranks;
  double* buf = nullptr;
  const size_t n = 4 * (size_t)(row_doubles > 0 ? row_doubles : 1) + (size_t)R * (nsmall > 0 ? nsmall : 1);
  HIPCHK(hipMalloc((void**)&buf, n * sizeof(double)));
  (void)hipMemsetAsync(buf, 0, n * sizeof(double), S->stream);
  double *slo = buf, *shi = buf + row_doubles, *rlo = buf + 2 * (size_t)row_doubles, *rhi = buf + 3 * (size_t)row_doubles, *small = buf + 4 * (size_t)(row_doubles > 0 ? row_doubles : 1);
  auto once = [&]() -> int {
    if (S->bulk.exchange) return S->bulk.exchange(S->bulk.ctx, slo, shi, rlo, rhi, row_doubles, small, nsmall);
    int rc = row_doubles > 0 ? S->bulk.halo(S->bulk.ctx, slo, shi, rlo, rhi, row_doubles) : 0;
    if (!rc && nsmall > 0) {
      int64_t off[64], cnt[64];
      for (int r = 0; r < R && r < 64; ++r) { off[r] = (int64_t)8 * nsmall * r; cnt[r] = (int64_t)8 * nsmall; }
      rc = S->bulk.allgather(S->bulk.ctx, small, off, cnt);
    }
    return rc;
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int rc = 0;
  for (int r = 0; r < 3 && !rc; ++r) rc = once();      // warm-up (connections, proxies)
  (void)hipEventRecord(e0, S->stream);
  for (int r = 0; r < reps && !rc; ++r) rc = once();
  (void)hipEventRecord(e1, S->stream);
  hipError_t e = hipStreamSynchronize(S->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  if (rc) { eu_set_error("euler_measure_exchange: the communicator's exchange failed"); return EULER_ECOMM; }
  if (e != hipSuccess) return eu_hip_fail(e, "exchange probe", __FILE__, __LINE__);
  *us_per_exchange = 1e3 * (double)ms / reps;
  return EULER_OK;
}

extern "C" int euler_sweep_timeline(euler_sim* S, uint64_t* out, int32_t cap_bands) {
  if (!S || !out || cap_bands < 0) return EULER_EINVAL;
  const int n = S->band_hi - S->band_lo < cap_bands ? S->band_hi - S->band_lo : cap_bands;
  HIPCHK(hipStreamSynchronize(S->stream));
  HIPCHK(hipMemcpy(out, S->sweep_timeline, (size_t)n * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return n;
}

extern "C" uint64_t euler_hbm_bytes(const euler_sim* S) { return S ? (uint64_t)S->hbm_bytes : 0; }

extern "C" int euler_device_name(euler_sim* S, char* out, int32_t cap) {
  if (!S || !out || cap < 1) return EULER_EINVAL;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, S->cfg.device));
  snprintf(out, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return EULER_OK;
}
