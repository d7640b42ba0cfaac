// k_pcg.hip — the pressure solve: preconditioned conjugate gradient on the masked 5-point
// Laplacian (reference project(), main.c:709-767, and its kernels main.c:580-702).
//
// Device-resident control: alpha, beta, sigma, the residual norm, the iteration count and the
// `done` flag live in PcgScalars in HBM; every kernel starts by reading them and returns at once
// after convergence, so the host enqueues iterations without a round trip and polls `done` every
// few iterations.
//
// Bit-exactness: every element-wise kernel evaluates the reference's expression in the
// reference's association order (compiled with -ffp-contract=off); the IC(0) triangular sweeps
// have no reduction, so any dependency-respecting schedule gives the sequential sweep's bits;
// dot() is either replayed sequentially (EULER_DOT_SEQUENTIAL) or reduced in a fixed tree.
#include "euler_dev.h"

#define RED_THREADS 256

__device__ __forceinline__ bool pcg_idle(const PcgScalars* sc) { return sc->done || !sc->nonzero; }

// ------------------------------------------------------------------------------------------
// scalar epilogues of the reductions
enum { FIN_SIGMA_INIT = 0, FIN_ALPHA, FIN_RNORM, FIN_BETA, FIN_STORE_ONLY };

__device__ __forceinline__ void pcg_scalar_step(PcgScalars* sc, int op, double v) {
  switch (op) {
    case FIN_SIGMA_INIT: sc->sigma = v; break;                                        // main.c:748
    case FIN_ALPHA: sc->zs = v; sc->alpha = sc->sigma / v; sc->iters += 1; break;     // main.c:750-752
    case FIN_RNORM: sc->rnorm = v; if (v <= sc->tol) sc->done = 1; break;             // main.c:756
    case FIN_BETA: sc->sigma_new = v; sc->beta = v / sc->sigma; sc->sigma = v; break; // main.c:762-765
    default: sc->sigma_new = v; break;
  }
}

// fixed-shape block reduction; result valid in thread 0
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double sw[RED_THREADS / 64];
  v = eu_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int k = 0; k < RED_THREADS / 64; ++k) t += sw[k];
  return t;
}
__device__ __forceinline__ double block_max(double v) {
  __shared__ double sm[RED_THREADS / 64];
  v = eu_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) for (int k = 0; k < RED_THREADS / 64; ++k) t = sm[k] > t ? sm[k] : t;
  return t;
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

// dot(a,b) over fluid cells -> per-block partials (tree mode)
__global__ __launch_bounds__(RED_THREADS) void k_dot_partial(const double* __restrict__ a, const double* __restrict__ b,
                                                             const uint8_t* __restrict__ mask, size_t C,
                                                             double* __restrict__ partial, const PcgScalars* sc, int force) {
  if (!force && pcg_idle(sc)) return;
  const size_t chunk = (C + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < C ? lo + chunk : C;
  double t = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += RED_THREADS)
    if (mask[i] & CM_FLUID) t += a[i] * b[i];
  t = block_sum(t);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// dot(a,b) replayed in the reference's order (main.c:629-639): products are formed by the whole
// workgroup, the running sum by one thread.  Bit-identical; meant for small grids.
#define SEQ_TILE 2048
__global__ __launch_bounds__(256) void k_dot_sequential(const double* __restrict__ a, const double* __restrict__ b,
                                                        const uint8_t* __restrict__ mask, size_t C, PcgScalars* sc,
                                                        int op, int force) {
  if (!force && pcg_idle(sc)) return;
  __shared__ double prod[SEQ_TILE];
  __shared__ uint8_t fl[SEQ_TILE];
  double total = 0.0;   // `double total = 0.f`
  for (size_t base = 0; base < C; base += SEQ_TILE) {
    for (int k = threadIdx.x; k < SEQ_TILE; k += 256) {
      const size_t i = base + k;
      const bool f = i < C && (mask[i] & CM_FLUID);
      fl[k] = f;
      prod[k] = f ? a[i] * b[i] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < SEQ_TILE; ++k)
        if (fl[k]) total += prod[k];
    __syncthreads();
  }
  if (threadIdx.x == 0) pcg_scalar_step(sc, op, total);
}

// ------------------------------------------------------------------------------------------
// apply_a (main.c:679-691): z = A s on fluid cells (other entries of z are already 0).
// In tree mode the block also leaves its partial of dot(z,s).
__global__ __launch_bounds__(RED_THREADS) void k_apply_a(const double* __restrict__ s, double* __restrict__ z,
                                                         const uint8_t* __restrict__ mask, int X, size_t C,
                                                         double* __restrict__ partial, const PcgScalars* sc, int force) {
  if (!force && pcg_idle(sc)) return;
  const size_t chunk = (C + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < C ? lo + chunk : C;
  double t = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += RED_THREADS) {
    const uint8_t m = mask[i];
    if (!(m & CM_FLUID)) continue;
    const double si = s[i];
    double o = (double)(int)(m >> CM_DIAG_SHIFT) * si;
    o = o - ((m & CM_RIGHT) ? s[i + 1] : 0.0);
    o = o - ((m & CM_UP) ? s[i + X] : 0.0);
    o = o - ((m & CM_LEFT) ? s[i - 1] : 0.0);
    o = o - ((m & CM_DOWN) ? s[i - X] : 0.0);
    z[i] = o;
    t += o * si;
  }
  t = block_sum(t);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// p += alpha s ; r -= alpha z (fmadd x2, main.c:753-754) ; per-block max |r| (inf_norm, main.c:654-667)
__global__ __launch_bounds__(RED_THREADS) void k_update_pr(double* __restrict__ p, double* __restrict__ r,
                                                           const double* __restrict__ s, const double* __restrict__ z,
                                                           const uint8_t* __restrict__ mask, size_t C,
                                                           double* __restrict__ partial, const PcgScalars* sc, int force,
                                                           double alpha_arg) {
  if (!force && pcg_idle(sc)) return;
  const double alpha = force ? alpha_arg : sc->alpha;
  const double nalpha = -alpha;
  const size_t chunk = (C + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < C ? lo + chunk : C;
  double mx = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += RED_THREADS) {
    if (!(mask[i] & CM_FLUID)) continue;
    p[i] = p[i] + s[i] * alpha;
    const double rn = r[i] + z[i] * nalpha;
    r[i] = rn;
    const double a = fabs(rn);
    if (a > mx) mx = a;
  }
  mx = block_max(mx);
  if (threadIdx.x == 0) partial[blockIdx.x] = mx;
}

// max |r| only (EULER_OP_INF_NORM_R)
__global__ __launch_bounds__(RED_THREADS) void k_inf_norm(const double* __restrict__ r, const uint8_t* __restrict__ mask,
                                                          size_t C, double* __restrict__ partial) {
  const size_t chunk = (C + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < C ? lo + chunk : C;
  double mx = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += RED_THREADS)
    if (mask[i] & CM_FLUID) { const double a = fabs(r[i]); if (a > mx) mx = a; }
  mx = block_max(mx);
  if (threadIdx.x == 0) partial[blockIdx.x] = mx;
}

// s = z + beta s (update_search, main.c:669-677); with COPY: s = z (the memcpy at main.c:746)
template <bool COPY>
__global__ __launch_bounds__(256) void k_update_search(double* __restrict__ s, const double* __restrict__ z,
                                                       const uint8_t* __restrict__ mask, size_t C, const PcgScalars* sc,
                                                       int force, double beta_arg) {
  if (!force && pcg_idle(sc)) return;
  const double beta = force ? beta_arg : sc->beta;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (size_t)gridDim.x * blockDim.x) {
    if (COPY) s[i] = z[i];
    else if (mask[i] & CM_FLUID) s[i] = z[i] + beta * s[i];
  }
}

// Jacobi stand-in preconditioner (not the reference's iterates; roofline comparison only)
__global__ __launch_bounds__(256) void k_jacobi(const double* __restrict__ r, double* __restrict__ z,
                                                const uint8_t* __restrict__ mask, size_t C, const PcgScalars* sc, int force) {
  if (!force && pcg_idle(sc)) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (size_t)gridDim.x * blockDim.x) {
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
  int X, Y, nbands;
  const uint8_t* mask;
  double* pre;            // precon: in/out for SW_FACTOR, in otherwise
  const double* in;       // r (forward) / q (backward); unused for factor
  double* out;            // q (forward) / z (backward); unused for factor
  unsigned long long* granules;   // [nbands][X][2] tagged hand-off of a band's last row
  unsigned int* ticket;
  unsigned int ticket_base;
  unsigned int epoch;
  const PcgScalars* sc;
  int force;
  int* error;
};

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
  const int X = a.X, Y = a.Y;
  constexpr bool BWD = OP == SW_BACKWARD;
  double* dst = OP == SW_FACTOR ? a.pre : a.out;
  for (int d = 0; d < X + Y - 1; ++d) {
    for (int yl = threadIdx.x; yl < Y; yl += 1024) {
      const int xl = d - yl;
      if (xl < 0 || xl >= X) continue;
      const int x = BWD ? X - 1 - xl : xl, y = BWD ? Y - 1 - yl : yl;
      const size_t i = (size_t)y * X + x;
      const uint8_t m = a.mask[i];
      if (OP == SW_FACTOR && !(m & CM_FLUID)) continue;
      double r = 0.0;
      if (m & CM_FLUID) {
        const size_t io = BWD ? i + 1 : i - 1, in_ = BWD ? i + X : i - X;   // fluid cells are interior
        const double own_val = OP == SW_FACTOR ? a.pre[io] : dst[io];
        const double nb_val = OP == SW_FACTOR ? a.pre[in_] : dst[in_];
        const double own_pre = OP == SW_FORWARD ? a.pre[io] : 0.0, nb_pre = OP == SW_FORWARD ? a.pre[in_] : 0.0;
        r = sweep_cell<OP>(m, OP == SW_FACTOR ? 0.0 : a.in[i], a.pre[i], own_val, own_pre, nb_val, nb_pre);
      }
      dst[i] = r;
    }
    __syncthreads();
  }
}

// --- production schedule: 64-row bands, one wave each, skewed along the row -------------------
// Lane l of band b owns row 64 b + l (in sweep order) and visits column tau - l at step tau, so
// the value of the row below arrives from lane l-1 one step later by a wave shift, and the
// previous column's value is the lane's own register.  Bands pipeline through HBM: the last lane
// publishes each value as two 8-byte {epoch, half} granules (one write-through store each); the
// next band's lane 0 polls them 16 columns at a time.  Bands take their index from a ticket, so a
// band can only wait on a band that is already running: no residency assumption, no deadlock.
#define SW_PF 8         // prefetch distance (steps) of the per-lane row streams
#define SW_POLL 16      // boundary columns fetched per poll
#define SW_SPIN_LIMIT (1u << 22)

__device__ __forceinline__ double wave_shift_up(double v) {   // lane l receives lane l-1's value
  return __shfl_up(v, 1, 64);
}

template <int OP>
__global__ __launch_bounds__(64) void k_sweep_band(SweepArgs a) {
  const int lane = threadIdx.x;
  unsigned int tk = 0;
  if (lane == 0) tk = atomicAdd(a.ticket, 1u);
  tk = __shfl(tk, 0, 64);
  const int band = (int)(tk - a.ticket_base);
  if (!a.force && pcg_idle(a.sc)) return;
  constexpr bool BWD = OP == SW_BACKWARD;
  const int X = a.X, Y = a.Y;
  const int yl = band * 64 + lane;
  const bool row_ok = yl < Y;
  const int y = BWD ? Y - 1 - yl : yl;
  const size_t row = (size_t)(row_ok ? y : 0) * X;
  const bool has_below = band > 0;                       // a previous band exists
  const bool publish = band + 1 < a.nbands && lane == 63;
  const size_t row_prev = has_below ? (size_t)(BWD ? y + 1 : y - 1) * X : 0;   // lane 0's neighbour row
  unsigned long long* gr_out = a.granules + (size_t)band * X * 2;
  const unsigned long long* gr_in = a.granules + (size_t)(has_below ? band - 1 : 0) * X * 2;
  const unsigned long long tag = (unsigned long long)a.epoch << 32;
  double* dst = OP == SW_FACTOR ? a.pre : a.out;

  // per-lane row streams, prefetched SW_PF steps ahead
  double in_buf[SW_PF], pre_buf[SW_PF], npre_buf[SW_PF];
  uint8_t m_buf[SW_PF];
  auto fetch = [&](int tau, int slot) {
    const int xl = tau - lane;
    const bool ok = row_ok && xl >= 0 && xl < X;
    const int x = BWD ? X - 1 - xl : xl;
    in_buf[slot] = (ok && OP != SW_FACTOR) ? a.in[row + x] : 0.0;
    pre_buf[slot] = ok ? a.pre[row + x] : 0.0;
    m_buf[slot] = ok ? a.mask[row + x] : (uint8_t)0;
    // lane 0 also needs precon of the row below it, which is static data in forward solves
    npre_buf[slot] = (OP == SW_FORWARD && ok && lane == 0 && has_below) ? a.pre[row_prev + x] : 0.0;
  };
#pragma unroll
  for (int j = 0; j < SW_PF; ++j) fetch(j, j);

  double own_val = 0.0, own_pre = 0.0;     // previous column of this row
  double out_val = 0.0, out_pre = 0.0;     // this lane's result of the previous step (for lane+1)
  double bnd = 0.0;                        // lane j: boundary value of column poll_base + j
  const int nsteps = X + 63;
  for (int t0 = 0; t0 < nsteps; t0 += SW_PF) {
#pragma unroll
    for (int j = 0; j < SW_PF; ++j) {
      const int tau = t0 + j;
      // ---- boundary row of the previous band (lane 0 consumes column tau at step tau)
      if (has_below && (tau % SW_POLL) == 0 && tau < X) {
        const int xl = tau + lane;
        const bool want = lane < SW_POLL && xl < X;
        const int x = BWD ? X - 1 - xl : xl;
        unsigned long long g0 = 0, g1 = 0;
        unsigned int spins = 0;
        while (true) {
          bool ready = true;
          if (want) {
            g0 = __hip_atomic_load(&gr_in[(size_t)x * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g1 = __hip_atomic_load(&gr_in[(size_t)x * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ready = ((g0 >> 32) == a.epoch) && ((g1 >> 32) == a.epoch);
          }
          if (__all(ready)) break;
          if (++spins > SW_SPIN_LIMIT) { if (lane == 0) atomicExch(a.error, 1); break; }
          __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long bits = (g0 & 0xffffffffull) | (g1 << 32);
        bnd = __longlong_as_double((long long)bits);
      }
      // ---- this step's operands
      const double cin = in_buf[j], cpre = pre_buf[j], cnpre = npre_buf[j];
      const uint8_t cm = m_buf[j];
      fetch(tau + SW_PF, j);
      const int xl = tau - lane;
      const bool act = row_ok && xl >= 0 && xl < X;
      const int x = BWD ? X - 1 - xl : xl;
      double nb_val = wave_shift_up(out_val);
      double nb_pre = OP == SW_FORWARD ? wave_shift_up(out_pre) : 0.0;
      const double bsel = __shfl(bnd, tau % SW_POLL, 64);   // executed by the whole wave
      if (lane == 0) {
        nb_val = has_below ? bsel : 0.0;
        nb_pre = cnpre;
      }
      double res = 0.0;
      if (act) {
        res = sweep_cell<OP>(cm, cin, cpre, own_val, own_pre, nb_val, nb_pre);
        if (OP != SW_FACTOR || (cm & CM_FLUID)) dst[row + x] = res;
        own_val = res;
        own_pre = cpre;
        if (publish) {
          const unsigned long long bits = (unsigned long long)__double_as_longlong(res);
          __hip_atomic_store(&gr_out[(size_t)x * 2], tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&gr_out[(size_t)x * 2 + 1], tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      out_val = res;
      out_pre = cpre;
    }
  }
}

// ==========================================================================================
// host-side launch helpers
static SweepArgs make_sweep_args(euler_sim* S, int op, int force) {
  SweepArgs a;
  a.X = S->X; a.Y = S->Y; a.nbands = S->nbands;
  a.mask = S->cellmask; a.pre = S->precon;
  a.in = op == SW_FORWARD ? S->r : S->q;
  a.out = op == SW_FORWARD ? S->q : S->z;
  a.granules = S->granules; a.ticket = S->ticket;
  a.ticket_base = S->ticket_base; a.epoch = S->epoch;
  a.sc = S->sc; a.force = force; a.error = &S->ms->error;
  return a;
}

static bool use_band(const euler_sim* S) {
  if (S->cfg.sweep_mode == EULER_SWEEP_SIMPLE) return false;
  return true;
}

template <int OP>
static int launch_sweep(euler_sim* S, int cls, int force) {
  if (use_band(S)) {
    S->epoch += 1;
    SweepArgs a = make_sweep_args(S, OP, force);
    LAUNCH(S, cls, k_sweep_band<OP>, dim3(S->nbands), dim3(64), a);
    S->ticket_base += (unsigned)S->nbands;
  } else {
    SweepArgs a = make_sweep_args(S, OP, force);
    LAUNCH(S, cls, k_sweep_simple<OP>, dim3(1), dim3(1024), a);
  }
  return EULER_OK;
}

static int launch_dot(euler_sim* S, const double* a, const double* b, int fin_op, int force) {
  const bool seq = S->cfg.dot_mode == EULER_DOT_SEQUENTIAL;
  if (seq) {
    LAUNCH(S, KC_DOT, k_dot_sequential, dim3(1), dim3(256), a, b, S->cellmask, S->C, S->sc, fin_op, force);
  } else {
    LAUNCH(S, KC_DOT, k_dot_partial, dim3(S->red_blocks), dim3(RED_THREADS), a, b, S->cellmask, S->C, S->partial, S->sc, force);
    LAUNCH(S, KC_REDUCE_FINAL, k_reduce_final<false>, dim3(1), dim3(RED_THREADS), S->partial, S->red_blocks, S->sc, fin_op, force);
  }
  return EULER_OK;
}

static int launch_precondition(euler_sim* S, int force) {   // z = M^-1 r
  if (S->cfg.precond == EULER_PRECOND_JACOBI) {
    LAUNCH(S, KC_JACOBI, k_jacobi, dim3(eu_blocks(S->C, 256 * 4, 4096)), dim3(256), S->r, S->z, S->cellmask, S->C, S->sc, force);
    return EULER_OK;
  }
  launch_sweep<SW_FORWARD>(S, KC_FORWARD_SOLVE, force);
  launch_sweep<SW_BACKWARD>(S, KC_BACKWARD_SOLVE, force);
  return EULER_OK;
}

static int launch_apply_a_and_alpha(euler_sim* S, int force) {
  const bool seq = S->cfg.dot_mode == EULER_DOT_SEQUENTIAL;
  LAUNCH(S, KC_APPLY_A, k_apply_a, dim3(S->red_blocks), dim3(RED_THREADS), S->s, S->z, S->cellmask, S->X, S->C, S->partial,
         S->sc, force);
  if (seq) {
    LAUNCH(S, KC_DOT, k_dot_sequential, dim3(1), dim3(256), S->z, S->s, S->cellmask, S->C, S->sc, (int)FIN_ALPHA, force);
  } else {
    LAUNCH(S, KC_REDUCE_FINAL, k_reduce_final<false>, dim3(1), dim3(RED_THREADS), S->partial, S->red_blocks, S->sc,
           (int)FIN_ALPHA, force);
  }
  return EULER_OK;
}

int eu_launch_build_system(euler_sim* S, float dt);
int eu_launch_velocity_update(euler_sim* S, float dt);

__global__ void k_pcg_reset(PcgScalars* sc, double tol, int max_iters) {
  sc->sigma = sc->zs = sc->sigma_new = sc->alpha = sc->beta = sc->rnorm = 0.0;
  sc->tol = tol; sc->nonzero = 0; sc->done = 0; sc->iters = 0; sc->max_iters = max_iters;
}

// project() (main.c:709-806)
int eu_launch_project(euler_sim* S, float dt) {
  LAUNCH(S, KC_MISC, k_pcg_reset, dim3(1), dim3(1), S->sc, S->cfg.tol, S->cfg.max_iterations);
  eu_launch_build_system(S, dt);
  // if (!all_zero(r)) { ... }: every kernel below is a no-op when sc->nonzero == 0
  if (S->cfg.precond == EULER_PRECOND_IC0) launch_sweep<SW_FACTOR>(S, KC_PRECON_FACTOR, 0);   // once per solve: A is fixed
  launch_precondition(S, 0);
  LAUNCH(S, KC_UPDATE_SEARCH, k_update_search<true>, dim3(eu_blocks(S->C, 256 * 4, 4096)), dim3(256), S->s, S->z,
         S->cellmask, S->C, S->sc, 0, 0.0);
  launch_dot(S, S->z, S->r, FIN_SIGMA_INIT, 0);
  const int poll = S->cfg.pcg_poll_interval > 0 ? S->cfg.pcg_poll_interval : 8;
  const int max_it = S->cfg.max_iterations;
  int it = 0;
  bool stop = false;
  while (it < max_it && !stop) {
    const int chunk_end = it + poll < max_it ? it + poll : max_it;
    for (; it < chunk_end; ++it) {
      launch_apply_a_and_alpha(S, 0);
      LAUNCH(S, KC_UPDATE_PR, k_update_pr, dim3(S->red_blocks), dim3(RED_THREADS), S->p, S->r, S->s, S->z, S->cellmask,
             S->C, S->partial, S->sc, 0, 0.0);
      LAUNCH(S, KC_REDUCE_FINAL, k_reduce_final<true>, dim3(1), dim3(RED_THREADS), S->partial, S->red_blocks, S->sc,
             (int)FIN_RNORM, 0);
      if (it + 1 < max_it) {   // the tail of the last iteration (main.c:760-765) is never consumed
        launch_precondition(S, 0);
        launch_dot(S, S->z, S->r, FIN_BETA, 0);
        LAUNCH(S, KC_UPDATE_SEARCH, k_update_search<false>, dim3(eu_blocks(S->C, 256 * 4, 4096)), dim3(256), S->s, S->z,
               S->cellmask, S->C, S->sc, 0, 0.0);
      }
    }
    if (it < max_it) {   // poll the device-side convergence flag
      HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
      HIPCHK(hipStreamSynchronize(S->stream));
      stop = S->sc_host->done || !S->sc_host->nonzero;
    }
  }
  eu_launch_velocity_update(S, dt);
  HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
  return EULER_OK;
}

// single building blocks for kernel-level parity tests (euler_pcg_op)
int eu_launch_pcg_op(euler_sim* S, int op, float dt, double a, double* out) {
  const bool want_scalar = op == EULER_OP_DOT_ZR || op == EULER_OP_DOT_ZS || op == EULER_OP_INF_NORM_R;
  switch (op) {
    case EULER_OP_BUILD_SYSTEM:
      LAUNCH(S, KC_MISC, k_pcg_reset, dim3(1), dim3(1), S->sc, S->cfg.tol, S->cfg.max_iterations);
      eu_launch_build_system(S, dt);
      break;
    case EULER_OP_PRECON_FACTOR: launch_sweep<SW_FACTOR>(S, KC_PRECON_FACTOR, 1); break;
    case EULER_OP_FORWARD_SOLVE: launch_sweep<SW_FORWARD>(S, KC_FORWARD_SOLVE, 1); break;
    case EULER_OP_BACKWARD_SOLVE: launch_sweep<SW_BACKWARD>(S, KC_BACKWARD_SOLVE, 1); break;
    case EULER_OP_APPLY_A:
      LAUNCH(S, KC_APPLY_A, k_apply_a, dim3(S->red_blocks), dim3(RED_THREADS), S->s, S->z, S->cellmask, S->X, S->C,
             S->partial, S->sc, 1);
      break;
    case EULER_OP_DOT_ZR: launch_dot(S, S->z, S->r, FIN_STORE_ONLY, 1); break;
    case EULER_OP_DOT_ZS: launch_dot(S, S->z, S->s, FIN_STORE_ONLY, 1); break;
    case EULER_OP_INF_NORM_R:
      LAUNCH(S, KC_UPDATE_PR, k_inf_norm, dim3(S->red_blocks), dim3(RED_THREADS), S->r, S->cellmask, S->C, S->partial);
      LAUNCH(S, KC_REDUCE_FINAL, k_reduce_final<true>, dim3(1), dim3(RED_THREADS), S->partial, S->red_blocks, S->sc,
             (int)FIN_STORE_ONLY, 1);
      break;
    case EULER_OP_UPDATE_PR:
      LAUNCH(S, KC_UPDATE_PR, k_update_pr, dim3(S->red_blocks), dim3(RED_THREADS), S->p, S->r, S->s, S->z, S->cellmask,
             S->C, S->partial, S->sc, 1, a);
      break;
    case EULER_OP_UPDATE_SEARCH:
      LAUNCH(S, KC_UPDATE_SEARCH, k_update_search<false>, dim3(eu_blocks(S->C, 256 * 4, 4096)), dim3(256), S->s, S->z,
             S->cellmask, S->C, S->sc, 1, a);
      break;
    default: eu_set_error("unknown pcg op %d", op); return EULER_EINVAL;
  }
  HIPCHK(hipMemcpyAsync(S->sc_host, S->sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, S->stream));
  HIPCHK(hipStreamSynchronize(S->stream));
  if (want_scalar && out) *out = S->sc_host->sigma_new;
  return EULER_OK;
}
