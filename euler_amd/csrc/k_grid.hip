// k_grid.hip — grid-stage kernels of one substep: CFL timestep, extrapolation into new fluid,
// boundary zeroing, semi-Lagrangian velocity advection (+ gravity + bounds), pressure-system
// assembly and the post-solve velocity update.  All are embarrassingly parallel gathers with
// coalesced row-major access; none has a floating-point reduction except the exact max.
#include "euler_dev.h"

#include <stdlib.h>

// ------------------------------------------------------------------------------------------
// calculate_timestep / maxsq (main.c:808-841): max over the typed extents, including zeros.
// A block walks whole rows (no division per element; float4 loads where the rows are 16-byte aligned), the maxima are exact and order-independent.  FUSE_DT
// (whole-grid handles): the block that delivers last forms dt as well - k_dt's arithmetic - instead of a second launch.
__device__ __forceinline__ void dt_from_maxima(MarkerState* ms, float frame_time_left, unsigned int mu_bits, unsigned int mv_bits) {
  const float mu = __uint_as_float(mu_bits), mv = __uint_as_float(mv_bits);
  const float max_distance = 0.75f * EU_H;
  const float max_velocity = sqrtf(mu + mv);
  ms->dt = fminf(max_distance / max_velocity, frame_time_left);   // 0.75/0 = +inf -> frame time
  ms->max_u2_bits = 0u;
  ms->max_v2_bits = 0u;
}
template <bool FUSE_DT>
__global__ __launch_bounds__(256) void k_maxsq(const float* __restrict__ u, const float* __restrict__ v,
                                               int X, int Y, MarkerState* ms, int y0, int y1, float frame_time_left) {   // rows [y0, y1) of this rank
  float mu = 0.f, mv = 0.f;
  const bool vec = (X & 3) == 0;
  for (int yq = y0 + 4 * (int)blockIdx.x; yq < y1; yq += 4 * (int)gridDim.x) {      // four rows at a time: eight independent loads in flight per thread
    if (vec) {
      for (int x = 4 * (int)threadIdx.x; x < X; x += 4 * 256) {
        float4 a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = yq + r < y1 ? yq + r : y1 - 1;      // (a row counted twice changes no maximum)
          a[r] = *reinterpret_cast<const float4*>(u + (size_t)y * X + x);
          b[r] = *reinterpret_cast<const float4*>(v + (size_t)y * X + x);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int y = yq + r < y1 ? yq + r : y1 - 1;
          float s = a[r].x * a[r].x; if (s > mu) mu = s;
          s = a[r].y * a[r].y; if (s > mu) mu = s;
          s = a[r].z * a[r].z; if (s > mu) mu = s;
          if (x + 3 < X - 1) { s = a[r].w * a[r].w; if (s > mu) mu = s; }
          if (y < Y - 1) {
            s = b[r].x * b[r].x; if (s > mv) mv = s;
            s = b[r].y * b[r].y; if (s > mv) mv = s;
            s = b[r].z * b[r].z; if (s > mv) mv = s;
            s = b[r].w * b[r].w; if (s > mv) mv = s;
          }
        }
      }
    } else {
      for (int r = 0; r < 4 && yq + r < y1; ++r) {
        const int y = yq + r;
        const float* ur = u + (size_t)y * X;
        const float* vr = v + (size_t)y * X;
        for (int x = (int)threadIdx.x; x < X; x += 256) {
          if (x < X - 1) { const float s = ur[x] * ur[x]; if (s > mu) mu = s; }
          if (y < Y - 1) { const float s = vr[x] * vr[x]; if (s > mv) mv = s; }
        }
      }
    }
  }
  mu = eu_wave_maxf(mu);
  mv = eu_wave_maxf(mv);
  __shared__ float su[4], sv[4];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) { su[w] = mu; sv[w] = mv; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { if (su[k] > mu) mu = su[k]; if (sv[k] > mv) mv = sv[k]; }
    // non-negative floats order like their bit patterns; NaN never gets here (s > m is false)
    atomicMax(&ms->max_u2_bits, __float_as_uint(mu));
    atomicMax(&ms->max_v2_bits, __float_as_uint(mv));
    if (FUSE_DT) {
      __threadfence();
      if (atomicAdd(&ms->dt_ticket, 1u) == gridDim.x - 1) {      // every block's maxima are in
        __threadfence();
        dt_from_maxima(ms, frame_time_left, __hip_atomic_load(&ms->max_u2_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(&ms->max_v2_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store(&ms->dt_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

__global__ void k_dt(MarkerState* ms, float frame_time_left) { dt_from_maxima(ms, frame_time_left, ms->max_u2_bits, ms->max_v2_bits); }

int eu_launch_dt(euler_sim* S, float frame_time_left) {
  LAUNCH(S, KC_TIMESTEP, k_dt, dim3(1), dim3(1), S->ms, frame_time_left);
  return EULER_OK;
}

int eu_launch_timestep(euler_sim* S, float frame_time_left) {
  const int quads = (S->row_hi - S->row_lo + 3) / 4;      // a block walks four rows at a time
  const unsigned nb = (unsigned)(quads < 1024 ? quads : 1024);
  if (S->slab_on) {      // max over the slabs first
    LAUNCH(S, KC_TIMESTEP, k_maxsq<false>, dim3(nb), dim3(256), S->u, S->v, S->X, S->Y, S->ms, S->row_lo, S->row_hi, frame_time_left);
    return eu_slab_timestep(S, frame_time_left);
  }
  if (S->maxsq_state == 2) {      // k_velocity_update_para left the maxima of exactly these u, v (nothing has written them since: driver.hip drops the state otherwise)
    S->maxsq_state = 0;
    return eu_launch_dt(S, frame_time_left);
  }
  if (S->maxsq_state == 1) { HIPCHK(hipMemsetAsync(&S->ms->max_u2_bits, 0, 2 * sizeof(unsigned int), S->stream)); S->maxsq_state = 0; }      // (maxima of a state edited since)
  LAUNCH(S, KC_TIMESTEP, k_maxsq<true>, dim3(nb), dim3(256), S->u, S->v, S->X, S->Y, S->ms, S->row_lo, S->row_hi, frame_time_left);
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// extrapolate (main.c:158-185) for U and V in one launch.  In place is race-free: only samples
// WITHOUT the prev-fluid property are written, only samples WITH it are read.
template <int TYPE>
__device__ __forceinline__ bool typed_prop(const uint8_t* g, size_t i, int X) {
  return TYPE == 1 ? eu_prop_u(g, i) : eu_prop_v(g, i, X);
}

template <int TYPE>
__device__ __forceinline__ void extrapolate_sample(float* q, const uint8_t* prev, const uint8_t* cur,
                                                   int x, int y, int X, int ex, int ey) {
  const size_t i = (size_t)y * X + x;
  if (typed_prop<TYPE>(prev, i, X) || !typed_prop<TYPE>(cur, i, X)) return;
  const int x0 = x > 0 ? x - 1 : 0, x1 = x + 1 < ex ? x + 1 : ex - 1;
  const int y0 = y > 0 ? y - 1 : 0, y1 = y + 1 < ey ? y + 1 : ey - 1;
  float total = 0.f;
  int n = 0;
  for (int yy = y0; yy <= y1; ++yy)       // y outer, x inner, float adds in this order (main.c:161-168)
    for (int xx = x0; xx <= x1; ++xx) {
      const size_t j = (size_t)yy * X + xx;
      if (typed_prop<TYPE>(prev, j, X)) { total += q[j]; ++n; }
    }
  q[i] = total / n;   // n == 0 -> 0/0 like the Release reference
}

__global__ __launch_bounds__(256) void k_extrapolate(float* u, float* v, const uint8_t* __restrict__ prev,
                                                     const uint8_t* __restrict__ cur, int X, int Y, int y0, int y1) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  if (x < X - 1) extrapolate_sample<1>(u, prev, cur, x, y, X, X - 1, Y);
  if (y < Y - 1) extrapolate_sample<2>(v, prev, cur, x, y, X, X, Y - 1);
}

// zero_bounds (main.c:822-832) for U and V in one launch.
__global__ __launch_bounds__(256) void k_zero_bounds(float* u, float* v, const uint8_t* __restrict__ cur,
                                                     const uint8_t* __restrict__ solid, int X, int Y, int y0, int y1) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  const size_t i = (size_t)y * X + x;
  if (x < X - 1 && (!eu_prop_u(cur, i) || eu_prop_u(solid, i))) u[i] = 0.f;
  if (y < Y - 1 && (!eu_prop_v(cur, i, X) || eu_prop_v(solid, i, X))) v[i] = 0.f;
}

// The same two stages, FOUR cells per thread (X a multiple of 4: rows are 4-byte aligned in the byte grids, 16-byte aligned in u and v).  A thread per cell
// is a wave of a dozen instructions per 64 cells - at 8192^2 a million waves that the dispatcher, not the memory system, takes 200 us to start.  Here a thread
// reads its four cells' bytes (and the cell to their right, and the row above) as words, decides the four samples, and writes - zero_bounds - one float4 where
// all four go to zero (air, the bulk of what it writes); extrapolate falls back to the per-sample routine for the rare sample that has just become fluid.
__device__ __forceinline__ unsigned int eu_nonzero_bytes(unsigned int w) {      // bit k set <=> byte k of w is non-zero
  return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
}
// per cell k of the four at (x, y): bit k of .x = u_property, of .y = v_property of grid g (main.c:119-138); the caller masks the grid's last column / row
__device__ __forceinline__ uint2 eu_props4(const uint8_t* __restrict__ g, size_t i, int x, int y, int X, int Y) {
  const unsigned int c = eu_nonzero_bytes(*reinterpret_cast<const unsigned int*>(g + i));
  const unsigned int right = (x + 4 < X && g[i + 4]) ? 1u : 0u;
  const unsigned int up = y + 1 < Y ? eu_nonzero_bytes(*reinterpret_cast<const unsigned int*>(g + i + X)) : 0u;
  return make_uint2(c | (c >> 1) | (right << 3), c | up);
}
// LEAN (round 6, whole-grid handles): the air is not zeroed again.  At the end of a substep every sample without the typed fluid property, and every one with the typed solid
// property, IS zero (the velocity update sets the air's and the walls' faces to 0, main.c:784-790, 797-803; in the first substep u = v = 0).  Of the samples zero_bounds has to
// zero now - no fluid property in the NEW count grid, or solid - only two kinds can hold anything else: those that had the fluid property in the previous grid and no wall
// (a velocity that the water has left), and walls' samples that extrapolate has just written (fluid now, not before).  16384^2 dam break: 1.5 GB of zeros per substep not written.
// (Valid while exactly one refresh_marker_counts lies between the last velocity update and this call and nobody edited the state: euler_sim.uv_clean.)
template <bool LEAN>
__global__ __launch_bounds__(256) void k_zero_bounds4(float* u, float* v, const uint8_t* __restrict__ cur,
                                                      const uint8_t* __restrict__ solid, int X, int Y, int y0, int y1, const uint8_t* __restrict__ prev,
                                                      const uint8_t* __restrict__ tmap, int tnx, int tn, int rep) {
  // (a workgroup: 256 columns x 4 rep rows inside one row of tiles; no fluid property now or before - nothing to zero)
  if (LEAN && tmap && eu_tiles_idle(tmap, tnx, tn, (y0 + (int)blockIdx.y * 4 * rep) >> 6, (int)blockIdx.x * 4, 4)) return;
  const int x = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
  if (x >= X) return;
  for (int r = 0; r < rep; ++r) {
    const int y = y0 + ((int)blockIdx.y * rep + r) * 4 + (int)(threadIdx.x >> 6);
    if (y >= y1) return;
    const size_t i = (size_t)y * X + x;
    const uint2 pc = eu_props4(cur, i, x, y, X, Y), ps = eu_props4(solid, i, x, y, X, Y);
    unsigned int zu = (~pc.x | ps.x) & 0xfu, zv = (~pc.y | ps.y) & 0xfu;      // samples that go to zero
    if (LEAN) {
      const uint2 pp = eu_props4(prev, i, x, y, X, Y);
      zu &= (pp.x & ~ps.x) | (pc.x & ~pp.x);
      zv &= (pp.y & ~ps.y) | (pc.y & ~pp.y);
    }
    if (x + 4 >= X) zu &= 0x7u;                                               // (the grid's last column holds no U sample)
    if (y >= Y - 1) zv = 0u;                                                  // (nor its last row a V sample)
    if (zu == 0xfu) *reinterpret_cast<float4*>(u + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    else { for (int k = 0; k < 4; ++k) if (zu & (1u << k)) u[i + k] = 0.f; }
    if (zv == 0xfu) *reinterpret_cast<float4*>(v + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    else { for (int k = 0; k < 4; ++k) if (zv & (1u << k)) v[i + k] = 0.f; }
  }
}
__global__ __launch_bounds__(256) void k_extrapolate4(float* u, float* v, const uint8_t* __restrict__ prev,
                                                      const uint8_t* __restrict__ cur, int X, int Y, int y0, int y1,
                                                      const uint8_t* __restrict__ tmap, int tnx, int tn, int rep) {
  if (tmap && eu_tiles_idle(tmap, tnx, tn, (y0 + (int)blockIdx.y * 4 * rep) >> 6, (int)blockIdx.x * 4, 4)) return;      // (no sample has just become fluid where there is no water)
  const int x = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
  if (x >= X) return;
  for (int r = 0; r < rep; ++r) {
    const int y = y0 + ((int)blockIdx.y * rep + r) * 4 + (int)(threadIdx.x >> 6);
    if (y >= y1) return;
    const size_t i = (size_t)y * X + x;
    const uint2 pp = eu_props4(prev, i, x, y, X, Y), pc = eu_props4(cur, i, x, y, X, Y);
    unsigned int eu = ~pp.x & pc.x & 0xfu, ev = ~pp.y & pc.y & 0xfu;          // samples that have just become fluid (main.c:158-160)
    if (x + 4 >= X) eu &= 0x7u;
    if (y >= Y - 1) ev = 0u;
    for (int k = 0; k < 4; ++k) {
      if (eu & (1u << k)) extrapolate_sample<1>(u, prev, cur, x + k, y, X, X - 1, Y);
      if (ev & (1u << k)) extrapolate_sample<2>(v, prev, cur, x + k, y, X, X, Y - 1);
    }
  }
}

int eu_launch_extrapolate(euler_sim* S) {
  // (a small grid has too few waves to hide the longer thread: 1024^2 extrapolates in 5.7 us a cell per thread, in 11.4 four per thread; 8192^2: 186 -> 79 us,
  // zero_bounds 217 -> 149 us)
  const size_t min_cells = (size_t)S->opt[EULER_OPT_GRID4_MIN_CELLS];      // (EULER_OPT_GRID4_MIN_CELLS; tests: 0 selects the four-cell kernels on any grid)
  if ((S->X & 3) == 0 && (size_t)S->X * (S->row_hi - S->row_lo) >= min_cells) {
    dim3 grid4((S->X / 4 + 63) / 64, (S->row_hi - S->row_lo + 3) / 4);
    const uint8_t* tm = eu_tile_map_on(S) ? S->tmap : nullptr;
    int rep = 1;
    if (tm) while (rep < 16 && ((size_t)S->X * S->Y >> 24) >= (size_t)(2 * rep)) rep *= 2;      // (8192^2: 4, 16384^2: 16 - at least 4 K workgroups)
    grid4.y = (unsigned)((S->row_hi - S->row_lo + 4 * rep - 1) / (4 * rep));
    LAUNCH(S, KC_EXTRAPOLATE, k_extrapolate4, grid4, dim3(256), S->u, S->v, S->prev_count, S->count, S->X, S->Y, S->row_lo, S->row_hi, tm, S->tmap_nx, S->tmap_n, rep);
    if (!S->slab_on && S->uv_clean == 2 && S->opt[EULER_OPT_VELOCITY_TWO_PASS] == 0)
      LAUNCH(S, KC_EXTRAPOLATE, k_zero_bounds4<true>, grid4, dim3(256), S->u, S->v, S->count, S->solid, S->X, S->Y, S->row_lo, S->row_hi, S->prev_count, tm, S->tmap_nx, S->tmap_n, rep);
    else
      LAUNCH(S, KC_EXTRAPOLATE, k_zero_bounds4<false>, grid4, dim3(256), S->u, S->v, S->count, S->solid, S->X, S->Y, S->row_lo, S->row_hi, (const uint8_t*)nullptr,
             (const uint8_t*)nullptr, 0, 0, rep);
    S->uv_zb = 1;      // (zero_bounds has run with the count grid as it stands)
    return EULER_OK;
  }
  dim3 grid((S->X + 63) / 64, (S->row_hi - S->row_lo + 3) / 4);   // this rank's rows (all of them without slabs)
  LAUNCH(S, KC_EXTRAPOLATE, k_extrapolate, grid, dim3(256), S->u, S->v, S->prev_count, S->count, S->X, S->Y, S->row_lo, S->row_hi);
  LAUNCH(S, KC_EXTRAPOLATE, k_zero_bounds, grid, dim3(256), S->u, S->v, S->count, S->solid, S->X, S->Y, S->row_lo, S->row_hi);
  S->uv_zb = 1;
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// advect_u + advect_v (main.c:378-422) + apply_body_forces (main.c:539-545) + zero_bounds on the
// outputs (main.c:888-889), fused: what reaches utmp/vtmp equals the reference's arrays after its
// second zero_bounds pair (non-fluid or solid faces are 0, so the reference's stale entries and the
// gravity it adds to non-fluid faces never survive).
__global__ __launch_bounds__(256) void k_advect_velocity(const float* __restrict__ u, const float* __restrict__ v,
                                                         float* __restrict__ uout, float* __restrict__ vout,
                                                         const uint8_t* __restrict__ solid, GridRef g, float dt, int y0, int y1,
                                                         const uint8_t* __restrict__ tmap, int tnx, int tn, int rep) {
  const int X = g.X, Y = g.Y;
  // round 6 (tmap: whole-grid handles whose utmp / vtmp this kernel wrote one refresh ago): a tile with no water in or next to it now and then holds the zeros it would get.
  // A workgroup takes 64 columns x 4 rep rows (rep = 16 on large grids: the tile; a wave that only looks at the map and leaves still costs its microsecond - 16384^2: 2.9 M of them)
  if (tmap && eu_tiles_idle(tmap, tnx, tn, (y0 + (int)blockIdx.y * 4 * rep) >> 6, (int)blockIdx.x, 1)) return;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  if (x >= X) return;
  for (int r = 0; r < rep; ++r) {
    const int y = y0 + ((int)blockIdx.y * rep + r) * 4 + (int)(threadIdx.x >> 6);
    if (y >= y1) return;
    const size_t i = (size_t)y * X + x;
    // (both faces' interpolations issued together - the v face not waiting for the u face - measured: no difference; the pass is bound by the number of its gathers)
    if (x < X - 1) {
      float out = 0.f;
      if (eu_prop_u(g.count, i) && !eu_prop_u(solid, i)) {
        const float dx = u[i];
        const float dy = eu_interp<2>(g, v, x + 0.5f, y - 0.5f);          // vidx_from_u, main.c:378-380
        const float px = x - dx * dt / EU_H, py = y - dy * dt / EU_H;     // main.c:392-393
        out = eu_interp<1>(g, u, px, py);
      }
      uout[i] = out;
    }
    if (y < Y - 1) {
      float out = 0.f;
      if (eu_prop_v(g.count, i, X) && !eu_prop_v(solid, i, X)) {
        const float dy = v[i];
        const float dx = eu_interp<1>(g, u, x - 0.5f, y + 0.5f);          // uidx_from_v, main.c:401-403
        const float px = x - dx * dt / EU_H, py = y - dy * dt / EU_H;
        out = eu_interp<2>(g, v, px, py);
        out += EU_G * dt;                                                 // main.c:542
      }
      vout[i] = out;
    }
  }
}

// EXTENSION (the reference is inviscid): one explicit diffusion step of the advected
// velocities over their live faces, out = q + nu*dt/h^2 * sum over live 4-neighbours (q_n - q), float,
// neighbours visited left, right, down, up - the arithmetic of oracle eo_diffuse.  Jacobi from the old
// values: reads utmp/vtmp, writes u/v, which are dead between advection and projection.
__global__ __launch_bounds__(256) void k_diffuse_velocity(const float* __restrict__ uin, const float* __restrict__ vin,
                                                          float* __restrict__ uout, float* __restrict__ vout,
                                                          const uint8_t* __restrict__ count, const uint8_t* __restrict__ solid,
                                                          int X, int Y, float c, int y0, int y1) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  const size_t i = (size_t)y * X + x;
  auto live_u = [&](size_t k) { return eu_prop_u(count, k) && !eu_prop_u(solid, k); };
  auto live_v = [&](size_t k) { return eu_prop_v(count, k, X) && !eu_prop_v(solid, k, X); };
  if (x < X - 1) {
    float r = uin[i];
    if (live_u(i)) {
      float acc = 0.f;
      if (x > 0 && live_u(i - 1)) acc += uin[i - 1] - uin[i];
      if (x + 1 < X - 1 && live_u(i + 1)) acc += uin[i + 1] - uin[i];
      if (y > 0 && live_u(i - X)) acc += uin[i - X] - uin[i];
      if (y + 1 < Y && live_u(i + X)) acc += uin[i + X] - uin[i];
      r = uin[i] + c * acc;
    }
    uout[i] = r;
  }
  if (y < Y - 1) {
    float r = vin[i];
    if (live_v(i)) {
      float acc = 0.f;
      if (x > 0 && live_v(i - 1)) acc += vin[i - 1] - vin[i];
      if (x + 1 < X && live_v(i + 1)) acc += vin[i + 1] - vin[i];
      if (y > 0 && live_v(i - X)) acc += vin[i - X] - vin[i];
      if (y + 1 < Y - 1 && live_v(i + X)) acc += vin[i + X] - vin[i];
      r = vin[i] + c * acc;
    }
    vout[i] = r;
  }
}
__global__ __launch_bounds__(256) void k_copy_typed(const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ uo,
                                                    float* __restrict__ vo, int X, int Y, int y0, int y1) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  const size_t i = (size_t)y * X + x;
  if (x < X - 1) uo[i] = u[i];
  if (y < Y - 1) vo[i] = v[i];
}

int eu_launch_diffuse(euler_sim* S, float dt);
int eu_launch_advect_velocity(euler_sim* S, float dt) {
  GridRef g{S->X, S->Y, S->count, S->interp_lim[0], S->interp_lim[1], S->interp_lim[2], S->interp_lim[3]};
  dim3 grid((S->X + 63) / 64, (S->row_hi - S->row_lo + 3) / 4);
  // (the tile map: utmp / vtmp must be what this kernel left one refresh ago, and nothing but the refresh may have touched the count grids since)
  const bool lean = eu_tile_map_on(S) && S->utmp_clean == 2 && !(S->cfg.viscosity > 0.f);
  int rep = 1;
  if (lean) while (rep < 16 && ((size_t)S->X * S->Y >> 22) >= (size_t)(2 * rep)) rep *= 2;      // (2048^2: 1, 4096^2: 4, 8192^2 and beyond: 16 - at least 16 K workgroups)
  grid.y = (unsigned)((S->row_hi - S->row_lo + 4 * rep - 1) / (4 * rep));
  LAUNCH(S, KC_ADVECT_VELOCITY, k_advect_velocity, grid, dim3(256), S->u, S->v, S->utmp, S->vtmp, S->solid, g, dt, S->row_lo, S->row_hi,
         lean ? (const uint8_t*)S->tmap : (const uint8_t*)nullptr, S->tmap_nx, S->tmap_n, rep);
  S->utmp_clean = (S->cfg.viscosity > 0.f) ? 0 : 1;      // (the diffusion extension writes utmp / vtmp behind this)
  // (a row-slab handle exchanges the ghost rows of utmp / vtmp between the two: eu_slab_substep calls eu_launch_diffuse itself)
  return S->slab_on ? EULER_OK : eu_launch_diffuse(S, dt);
}
// the diffusion extension over the own rows: reads utmp / vtmp one row below and above them
int eu_launch_diffuse(euler_sim* S, float dt) {
  if (!(S->cfg.viscosity > 0.f)) return EULER_OK;   // extension stage; absent (bit-identical to the reference) at viscosity 0
  dim3 grid((S->X + 63) / 64, (S->row_hi - S->row_lo + 3) / 4);
  const float c = S->cfg.viscosity * dt / (EU_H * EU_H);
  LAUNCH(S, KC_ADVECT_VELOCITY, k_diffuse_velocity, grid, dim3(256), S->utmp, S->vtmp, S->u, S->v, S->count, S->solid, S->X, S->Y, c, S->row_lo, S->row_hi);
  LAUNCH(S, KC_ADVECT_VELOCITY, k_copy_typed, grid, dim3(256), S->u, S->v, S->utmp, S->vtmp, S->X, S->Y, S->row_lo, S->row_hi);
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// b, A and the initial PCG vectors (main.c:713-741).  One byte per cell carries everything the
// solver needs to know about A: fluid flag, the four neighbour-fluid flags (the implicit -1
// off-diagonals, main.c:561-575) and a_diag = 4 - #solid neighbours (main.c:554-559).
// Threads run in the solver's band-skewed order (coalesced writes of b, r, p, mask; the row-major
// velocity/count reads are a diagonal gather, once per substep).  Padding entries are skipped:
// they keep mask 0 and value 0 from allocation.
// TILE (the tile-local preconditioner): q and z need no zeroing (k_precond_tile writes z over whole active tiles and reads A s = q
// on fluid cells only), and a wave without fluid in a chunk that held none in the previous solve either has nothing to write at
// all - its masks, p and r are still the zeros the last solve that touched them (or the allocation) left.  8192^2 half tank:
// 41 B x all cells -> 25 B x the cells of active chunks.
template <bool TILE>
__global__ __launch_bounds__(256) void k_build_system(const float* __restrict__ u, const float* __restrict__ v,
                                                      const uint8_t* __restrict__ count, const uint8_t* __restrict__ solid,
                                                      double* __restrict__ b, double* __restrict__ r, double* __restrict__ p,
                                                      double* __restrict__ q, double* __restrict__ z,
                                                      uint8_t* __restrict__ cellmask,
                                                      PcgScalars* sc, SkewGeom g, float dt, size_t e_lo, size_t e_cnt,
                                                      uint8_t* __restrict__ chunk_flag, uint8_t* __restrict__ chunk_part,
                                                      const uint8_t* __restrict__ chunk_prev) {
  // eu_xcd_block: the 16 diagonal steps that share a 64-byte line of a row-major field meet in ONE L2 instead of in all eight
  const size_t e = e_lo + eu_xcd_block() * blockDim.x + threadIdx.x;   // this rank's bands only
  // the wave's 64 elements are two half records of one 16-record chunk (euler_dev.h "Active chunks")
  const size_t we = e_lo + eu_xcd_block() * blockDim.x + (threadIdx.x & ~63u);      // the wave's first element
  const size_t per_band = (size_t)g.TS * 64;
  const size_t chunk = ((we - e_lo) / per_band) * (size_t)(g.T / 16) + ((we % per_band) / 64) / 16;
  const bool in_chunks = (we % per_band) / 64 < (size_t)g.T;      // (the TS - T records behind a band's last chunk are padding)
  bool nz = false, fl = false, inside = false;
  uint8_t m = 0;
  double bv = 0.0;
  if (e < e_lo + e_cnt) {
    const int X = g.X;
    int band, t, l;
    skew_decode(g, e, band, t, l);
    const int x = t - l, y = band * 64 + l;
    if (x >= 0 && x < X && y < g.Y) {
      inside = true;
      const size_t i = (size_t)y * X + x;
      if (count[i] != 0) {   // never true on the border ring (sinks), so the +-1 / +-X reads are in range
        m = CM_FLUID;
        fl = true;
        if (count[i + 1]) m |= CM_RIGHT;
        if (count[i + X]) m |= CM_UP;
        if (count[i - 1]) m |= CM_LEFT;
        if (count[i - X]) m |= CM_DOWN;
        const int diag = 4 - solid[i - 1] - solid[i + 1] - solid[i - X] - solid[i + X];
        m |= (uint8_t)(diag << CM_DIAG_SHIFT);
        const float k_inv_scale_f = (EU_H * EU_H) * EU_RHO / dt;                      // float expression, main.c:713
        const float div_f = (u[i] - u[i - 1] + v[i] - v[i - X]) / EU_H;               // float expression, main.c:720
        bv = -(double)div_f * (double)k_inv_scale_f;
        nz = bv != 0.0;
      }
    }
  }
  const bool wave_fluid = __any(fl);
  const bool touch = !TILE || wave_fluid || (in_chunks && chunk_prev[chunk] != 0);
  if (inside && touch) {
    cellmask[e] = m;
    b[e] = bv;
    r[e] = bv;
    p[e] = 0.0;
    if (!TILE) {
      q[e] = 0.0;   // the sweeps only visit the records that hold fluid (k_band_ranges): what they skip must be +0
      z[e] = 0.0;
    }
  }
  // !all_zero(r), main.c:742.  Once the flag is up nobody has to raise it again: in a moving fluid EVERY wave finds a nonzero b, and
  // half a million atomic ORs on one address took 3.4 of the kernel's 5.1 ms at 8192^2 (at rest, where b is mostly zero, none)
  if (__any(nz) && (threadIdx.x & 63) == 0 && __hip_atomic_load(&sc->nonzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    atomicOr(&sc->nonzero, 1);
  // one byte says the chunk holds fluid, a second one that it is not an INTERIOR chunk (euler_dev.h: every cell fluid with four
  // fluid neighbours and a_diag 4, mask CM_INTERIOR).  Plain stores: every wave writes the same value; an atomic OR into shared
  // words cost 3 ms
  if ((threadIdx.x & 63) == 0 && in_chunks) {
    if (wave_fluid) chunk_flag[chunk] = 1;
  }
  if (in_chunks && __any(m != CM_INTERIOR) && (threadIdx.x & 63) == 0) chunk_part[chunk] = 1;
}

// The assembly in two steps (round 3).  k_build_system above gathers along the diagonal: a wave's 64 lanes sit in 64 different rows, so
// each of the 13 loads of a fluid cell touches 64 cache lines - the texture path, not HBM, set its 1.27 ms at 8192^2 (1.2 GB of traffic).
// Here the neighbourhood work runs where it coalesces - k_cell_system, a thread per cell of the row-major grids, leaves the mask byte and
// the float divergence (main.c:720) per cell - and the skewed pass gathers just those two (one for a cell without fluid): same bits.
// (Measured and rejected: staging the row-major fields of a chunk's parallelogram through LDS - 81 columns for 16 records, 52 KB per 1024
// cells: 3.1 ms.)
__global__ __launch_bounds__(256) void k_cell_system(const float* __restrict__ u, const float* __restrict__ v, const uint8_t* __restrict__ count,
                                                     const uint8_t* __restrict__ solid, uint8_t* __restrict__ sys_m, float* __restrict__ sys_div,
                                                     int X, int y0, int y1, size_t win_off) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (size_t)X * (y1 - y0)) return;
  const size_t i = (size_t)y0 * X + k;
  uint8_t m = 0;
  float div_f = 0.f;
  if (count[i] != 0) {   // never true on the border ring (sinks), so the +-1 / +-X reads are in range
    m = CM_FLUID;
    if (count[i + 1]) m |= CM_RIGHT;
    if (count[i + X]) m |= CM_UP;
    if (count[i - 1]) m |= CM_LEFT;
    if (count[i - X]) m |= CM_DOWN;
    const int diag = 4 - solid[i - 1] - solid[i + 1] - solid[i - X] - solid[i + X];
    m |= (uint8_t)(diag << CM_DIAG_SHIFT);
    div_f = (u[i] - u[i - 1] + v[i] - v[i - X]) / EU_H;               // float expression, main.c:720
  }
  sys_m[i - win_off] = m;
  sys_div[i - win_off] = div_f;
}

template <bool TILE>
__global__ __launch_bounds__(256) void k_build_system2(const uint8_t* __restrict__ sys_m, const float* __restrict__ sys_div, size_t win_off,
                                                       double* __restrict__ b, double* __restrict__ r, double* __restrict__ p,
                                                       double* __restrict__ q, double* __restrict__ z,
                                                       uint8_t* __restrict__ cellmask,
                                                       PcgScalars* sc, SkewGeom g, float dt, size_t e_lo, size_t e_cnt,
                                                       uint8_t* __restrict__ chunk_flag, uint8_t* __restrict__ chunk_part,
                                                       const uint8_t* __restrict__ chunk_prev) {
  const size_t e = e_lo + eu_xcd_block() * blockDim.x + threadIdx.x;   // this rank's bands only
  const size_t we = e_lo + eu_xcd_block() * blockDim.x + (threadIdx.x & ~63u);      // the wave's first element
  const size_t per_band = (size_t)g.TS * 64;
  const size_t chunk = ((we - e_lo) / per_band) * (size_t)(g.T / 16) + ((we % per_band) / 64) / 16;
  const bool in_chunks = (we % per_band) / 64 < (size_t)g.T;      // (the TS - T records behind a band's last chunk are padding)
  bool nz = false, fl = false, inside = false;
  uint8_t m = 0;
  double bv = 0.0;
  if (e < e_lo + e_cnt) {
    const int X = g.X;
    int band, t, l;
    skew_decode(g, e, band, t, l);
    const int x = t - l, y = band * 64 + l;
    if (x >= 0 && x < X && y < g.Y) {
      inside = true;
      const size_t il = (size_t)y * X + x - win_off;
      m = sys_m[il];
      if (m) {
        fl = true;
        const float k_inv_scale_f = (EU_H * EU_H) * EU_RHO / dt;                      // float expression, main.c:713
        bv = -(double)sys_div[il] * (double)k_inv_scale_f;
        nz = bv != 0.0;
      }
    }
  }
  const bool wave_fluid = __any(fl);
  const bool touch = !TILE || wave_fluid || (in_chunks && chunk_prev[chunk] != 0);
  if (inside && touch) {
    cellmask[e] = m;
    b[e] = bv;
    r[e] = bv;
    p[e] = 0.0;
    if (!TILE) {
      q[e] = 0.0;   // the sweeps only visit the records that hold fluid (k_band_ranges): what they skip must be +0
      z[e] = 0.0;
    }
  }
  if (__any(nz) && (threadIdx.x & 63) == 0 && __hip_atomic_load(&sc->nonzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    atomicOr(&sc->nonzero, 1);
  if ((threadIdx.x & 63) == 0 && in_chunks) {
    if (wave_fluid) chunk_flag[chunk] = 1;
  }
  if (in_chunks && __any(m != CM_INTERIOR) && (threadIdx.x & 63) == 0) chunk_part[chunk] = 1;
}

// The assembly in ONE pass (round 6).  The two-step form above writes 5 bytes per cell of scratch and gathers them back along the diagonal (0.22 + 0.60 ms at 8192^2).  A run of
// W consecutive records of a band is a PARALLELOGRAM of the grid: row l of the band holds its cells x = t0 - l .. t0 - l + W - 1 - W consecutive cells of a row-major row.  A
// workgroup takes BS_W = 48 records (three chunks; T is a whole number of 96): it walks the 64 row segments with consecutive threads on consecutive cells (coalesced
// 192-byte reads of count, solid, utmp, vtmp; the four neighbours come out of the same lines), leaves mask byte and float divergence (main.c:720) in LDS, and writes the skewed
// arrays from there with consecutive threads on consecutive elements.  The same arithmetic per cell: the same bits; the chunk flags as before.
#define BS_W 48            // records per workgroup (T is a whole number of 96): 15 KB of LDS, ten workgroups per CU
template <bool TILE>
__global__ __launch_bounds__(256) void k_build_system_para(const float* __restrict__ u, const float* __restrict__ v, const uint8_t* __restrict__ count, const uint8_t* __restrict__ solid,
                                                           double* __restrict__ b, double* __restrict__ r, double* __restrict__ p, double* __restrict__ q, double* __restrict__ z,
                                                           uint8_t* __restrict__ cellmask, PcgScalars* sc, SkewGeom g, float dt, int band_lo,
                                                           uint8_t* __restrict__ chunk_flag, uint8_t* __restrict__ chunk_part, const uint8_t* __restrict__ chunk_prev,
                                                           const uint8_t* __restrict__ tmap, int tnx) {
  __shared__ float sd[64][BS_W + 1];
  __shared__ uint8_t sm[64][BS_W + 4];
  __shared__ int s_fl[BS_W / 16], s_part[BS_W / 16], s_prev[BS_W / 16], s_nz;
  const int units = g.T / BS_W;
  const size_t blk = eu_xcd_block();      // neighbouring units share the lines their slanted edges cut: one L2 for a run of them
  const int band = band_lo + (int)(blk / units), t0 = (int)(blk % units) * BS_W;
  const int X = g.X, Y = g.Y, tid = threadIdx.x;
  const size_t chunk0 = (size_t)(band - band_lo) * (size_t)(g.T / 16) + (size_t)(t0 / 16);
  if (tid < BS_W / 16) { s_fl[tid] = 0; s_part[tid] = 0; s_prev[tid] = TILE ? (chunk_prev[chunk0 + tid] != 0) : 1; }
  if (tid == 0) s_nz = 0;
  __syncthreads();
  if (TILE && tmap) {      // the tile map (euler_dev.h): no water in the tiles the parallelogram lies in, none in its chunks at the previous solve - masks, b, r, p are the zeros they would get
    bool prev_any = false;
    for (int k = 0; k < BS_W / 16; ++k) prev_any = prev_any || s_prev[k] != 0;
    if (!prev_any && eu_tiles_idle_cur(tmap, tnx, band, t0 > 63 ? (t0 - 63) >> 6 : 0, ((t0 + BS_W - 1) >> 6) - 1, false)) {
      if (tid < BS_W / 16) chunk_part[chunk0 + tid] = 1;      // (what the pass leaves for chunks of air: not INTERIOR)
      return;
    }
  }
  static_assert((64 * BS_W) % (256 * 4) == 0, "whole rounds of four cells per thread");
  for (int c0 = tid; c0 < 64 * BS_W; c0 += 256 * 4) {
    // four cells per thread; a wave whose cells hold no fluid is done after the count bytes, any other issues every load of the round before the first use
    // (the neighbours of a cell that cannot be fluid - the border ring, beyond the grid - are read at the cell itself: in range, unused)
    size_t ii[4]; int ll[4], jj[4]; uint8_t cc[4]; size_t dx[4], dy[4];
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 256 * u;
      ll[u] = c / BS_W; jj[u] = c % BS_W;
      const int x = t0 - ll[u] + jj[u], y = band * 64 + ll[u];
      const bool in = x >= 0 && x < X && y < Y;
      const bool inner = x > 0 && x < X - 1 && y > 0 && y < Y - 1;
      // (a cell beyond the grid's edge reads the nearest cell of its own row instead - never cell 0: a row-slab handle's arrays hold its window of rows only)
      ii[u] = (size_t)(y < Y ? y : Y - 1) * X + (size_t)(x < 0 ? 0 : x >= X ? X - 1 : x);
      dx[u] = inner && in ? (size_t)1 : (size_t)0; dy[u] = inner && in ? (size_t)X : (size_t)0;
      cc[u] = in ? count[ii[u]] : (uint8_t)0;      // never non-zero on the border ring (sinks)
      any = any || cc[u] != 0;
    }
    uint8_t mm[4] = {0, 0, 0, 0};
    float dv[4] = {0.f, 0.f, 0.f, 0.f};
    if (__any(any)) {
      uint8_t cr[4], cu[4], cl[4], cd[4], sr[4], su[4], sl[4], sdn[4];
      float u0[4], u1[4], v0[4], v1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t i = ii[k];
        cr[k] = count[i + dx[k]]; cu[k] = count[i + dy[k]]; cl[k] = count[i - dx[k]]; cd[k] = count[i - dy[k]];
        sr[k] = solid[i + dx[k]]; su[k] = solid[i + dy[k]]; sl[k] = solid[i - dx[k]]; sdn[k] = solid[i - dy[k]];
        u0[k] = u[i]; u1[k] = u[i - dx[k]]; v0[k] = v[i]; v1[k] = v[i - dy[k]];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (cc[k] == 0) continue;
        uint8_t m = CM_FLUID;
        if (cr[k]) m |= CM_RIGHT;
        if (cu[k]) m |= CM_UP;
        if (cl[k]) m |= CM_LEFT;
        if (cd[k]) m |= CM_DOWN;
        const int diag = 4 - sl[k] - sr[k] - sdn[k] - su[k];
        m |= (uint8_t)(diag << CM_DIAG_SHIFT);
        mm[k] = m;
        dv[k] = (u0[k] - u1[k] + v0[k] - v1[k]) / EU_H;               // float expression, main.c:720
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sm[ll[k]][jj[k]] = mm[k];
      sd[ll[k]][jj[k]] = dv[k];
      if (mm[k] & CM_FLUID) s_fl[jj[k] / 16] = 1;                 // (plain stores of the same value)
      if (mm[k] != CM_INTERIOR) s_part[jj[k] / 16] = 1;
    }
  }
  __syncthreads();
  const float k_inv_scale_f = (EU_H * EU_H) * EU_RHO / dt;                      // float expression, main.c:713
  bool nz = false;
  for (int c = tid; c < 64 * BS_W; c += 256) {          // element c of the unit: pair-record c / 128, lane (c % 128) / 2, parity c % 2
    const int pr = c >> 7, l = (c & 127) >> 1, j = 2 * pr + (c & 1);
    const int ch = j / 16;
    const bool touch = s_fl[ch] || s_prev[ch];      // (TILE: a chunk without fluid now and in the previous solve still holds the zeros that solve's assembly left)
    const int x = t0 + j - l, y = band * 64 + l;
    if (!touch || x < 0 || x >= X || y >= Y) continue;  // (padding entries keep mask 0 and value 0 from allocation)
    const uint8_t m = sm[l][j];
    const double bv = m ? -(double)sd[l][j] * (double)k_inv_scale_f : 0.0;
    nz = nz || bv != 0.0;
    const size_t e = ((size_t)band * g.TS + (size_t)(t0 + 2 * pr)) * 64 + (size_t)(c & 127);
    cellmask[e] = m;
    b[e] = bv;
    r[e] = bv;
    p[e] = 0.0;
    if (!TILE) { q[e] = 0.0; z[e] = 0.0; }
  }
  if (nz) s_nz = 1;
  __syncthreads();
  if (tid < BS_W / 16) {
    if (s_fl[tid]) chunk_flag[chunk0 + tid] = 1;
    if (s_part[tid]) chunk_part[chunk0 + tid] = 1;
  }
  // !all_zero(r), main.c:742 (once the flag is up nobody has to raise it again)
  if (tid == 0 && s_nz && __hip_atomic_load(&sc->nonzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&sc->nonzero, 1);
}

// one bit per chunk from the bytes k_build_system left (the ordered select wants bits)
__global__ __launch_bounds__(256) void k_pack_chunk_bits(const uint8_t* __restrict__ flag, unsigned long long* __restrict__ bits, size_t nwords, size_t nchunks) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned long long v = 0;
  for (int k = 0; k < 64; ++k) {
    const size_t c = w * 64 + k;
    if (c < nchunks && flag[c]) v |= 1ull << k;
  }
  bits[w] = v;
}

// Pressure clamp (main.c:773-779) + velocity update (main.c:782-805), fused.  The clamp is
// idempotent, so applying it on the fly to the neighbour reads races benignly with the write.
// Row-major threads (coalesced velocity writes); p is gathered from the skewed layout.
__global__ __launch_bounds__(256) void k_velocity_update(const float* __restrict__ uin, const float* __restrict__ vin,
                                                         float* __restrict__ uout, float* __restrict__ vout,
                                                         double* p, const uint8_t* __restrict__ count,
                                                         const uint8_t* __restrict__ solid, SkewGeom g, float dt, int y0, int y1) {
  const int X = g.X, Y = g.Y;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  const size_t i = (size_t)y * X + x;
  const size_t e = skew_index(g, x, y);
  const bool f0 = count[i] != 0;
  double p0 = p[e];
  if (f0 && p0 < 0.0) { p0 = 0.0; p[e] = 0.0; }
  const float neg_inv = -(1.f / (EU_RHO * EU_H));   // accel(), main.c:705-707
  if (x < X - 1) {
    const bool f1 = count[i + 1] != 0;
    float o = 0.f;
    if (solid[i] | solid[i + 1]) o = 0.f;
    else if (f0 | f1) {
      double p1 = p[skew_index(g, x + 1, y)];
      if (f1 && p1 < 0.0) p1 = 0.0;
      o = uin[i] + (neg_inv * (float)(p1 - p0)) * dt;
    }
    uout[i] = o;
  }
  if (y < Y - 1) {
    const bool f1 = count[i + X] != 0;
    float o = 0.f;
    if (solid[i] | solid[i + X]) o = 0.f;
    else if (f0 | f1) {
      double p1 = p[skew_index(g, x, y + 1)];
      if (f1 && p1 < 0.0) p1 = 0.0;
      o = vin[i] + (neg_inv * (float)(p1 - p0)) * dt;
    }
    vout[i] = o;
  }
}

// Round 6: the end of project() in ONE pass over parallelograms (k_build_system_para's geometry) - the p += alpha s that are still due (k_finish_p's arithmetic, main.c:753),
// the clamp (main.c:773-779), both velocity updates (main.c:782-805) and the maxima calculate_timestep wants next (main.c:808-841).  A workgroup finishes the pressure of its
// VU_W = 48 records in skewed order (coalesced reads of p, the mask and the pending search directions) into LDS - plus the record to its right and the row above its band, formed
// the same way from the same inputs - and walks the 64 row segments with consecutive threads on consecutive cells: what rounds 1-5 gathered from the skewed array (three
// pressures per cell) comes out of LDS.  The pass READS p and WRITES none of it (no workgroup sees another's result): the finished, clamped pressure is formed in memory
// only when somebody asks for it (eu_pressure_current: EULER_F_PRESSURE, euler_pcg_op), from the same ring - most substeps nobody does, and 8 bytes per fluid cell stay unwritten.
struct VuRing { const double* s[8]; int n, steps, use; };      // use 0: p is final as it stands (the resident solver wrote it; a right-hand side of zeros)
#define VU_W 48           // records per workgroup (T is a whole number of 96): 25 KB of LDS, six workgroups per CU
#define VU_ILP 4          // elements per thread whose loads are in flight together (the pass is a chain mask -> p, ring -> LDS per element: latency, not bytes, at one)
__global__ __launch_bounds__(256) void k_velocity_update_para(const float* __restrict__ uin, const float* __restrict__ vin, float* __restrict__ uout, float* __restrict__ vout,
                                                              const double* __restrict__ p, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ count,
                                                              const uint8_t* __restrict__ solid, SkewGeom g, float dt, const PcgScalars* sc, VuRing ring, MarkerState* ms,
                                                              int band_lo, int do_max, int skip_zero,      // this rank's bands start at band_lo; do_max 0: a row slab's maxima are k_maxsq's (all-reduced)
                                                              const uint8_t* __restrict__ tmap, int tnx) {
  // skip_zero: the faces this pass sets to 0 - the air's, the walls' (main.c:784-790, 797-803) - are not stored: zero_bounds zeroed exactly those samples of u, v earlier in
  // this substep, with the same count grid (euler_sim.uv_zb), and nothing has written them since
  __shared__ double sp[65][VU_W + 1];
  __shared__ float s_mu[4], s_mv[4];
  const int units = g.T / VU_W;
  const size_t blk = eu_xcd_block();
  const int band = band_lo + (int)(blk / units), t0 = (int)(blk % units) * VU_W;
  const int X = g.X, Y = g.Y, tid = threadIdx.x;
  // the tile map (euler_dev.h; with skip_zero): no water in the tiles the parallelogram lies in, to their right or above - every face is the air's, zero already, and no maximum
  if (skip_zero && tmap && eu_tiles_idle_cur(tmap, tnx, band, t0 > 63 ? (t0 - 63) >> 6 : 0, (t0 + VU_W - 1) >> 6, true)) return;
  // the fmadds that are due: iterations from .. n_it - 1 (k_finish_p)
  int cnt = 0, from = 0;
  if (ring.use && sc->nonzero && sc->iters > 0) {
    const int n_it = sc->iters;
    from = n_it - 1 >= ring.steps ? (n_it - 1) / ring.steps * ring.steps : 0;
    cnt = n_it - from;
  }
  double al[8];
  const double* sq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int it = from + (j < cnt ? j : 0); al[j] = sc->alpha_hist[it & 7]; sq[j] = ring.s[it % ring.n]; }
  // the clamped, finished pressure of element e; +0 off the fluid (main.c:739: p starts as zeros and only fluid cells are ever written)
  const size_t ebase = ((size_t)band * g.TS + (size_t)t0) * 64;
  static_assert((64 * VU_W) % (256 * VU_ILP) == 0, "whole rounds of VU_ILP elements per thread");
  for (int c0 = tid; c0 < 64 * VU_W; c0 += 256 * VU_ILP) {      // element c of the unit: pair-record c / 128, lane (c % 128) / 2, parity c % 2
    uint8_t m[VU_ILP];
    double pv[VU_ILP];
    bool any = false;
#pragma unroll
    for (int u = 0; u < VU_ILP; ++u) { m[u] = mask[ebase + (size_t)(c0 + 256 * u)]; any = any || (m[u] & CM_FLUID); }
    if (__any(any)) {      // (uniform: a wave whose elements hold no fluid loads nothing else - the air above a tank)
#pragma unroll
      for (int u = 0; u < VU_ILP; ++u) pv[u] = p[ebase + (size_t)(c0 + 256 * u)];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cnt) {      // (uniform) the fmadds in their order (main.c:753), per element
          double sv[VU_ILP];
#pragma unroll
          for (int u = 0; u < VU_ILP; ++u) sv[u] = sq[j][ebase + (size_t)(c0 + 256 * u)];
#pragma unroll
          for (int u = 0; u < VU_ILP; ++u) pv[u] = pv[u] + sv[u] * al[j];
        }
    } else {
#pragma unroll
      for (int u = 0; u < VU_ILP; ++u) pv[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < VU_ILP; ++u) {
      const int c = c0 + 256 * u;
      const double v = (m[u] & CM_FLUID) ? (pv[u] < 0.0 ? 0.0 : pv[u]) : 0.0;
      sp[(c & 127) >> 1][2 * (c >> 7) + (c & 1)] = v;
    }
  }
  {      // the record to the right (t0 + W <= T: a padding record at worst, mask 0) and the row above the band: lane 0 of band + 1, records t0 - 64 + j (j = 1 .. W)
    size_t e = 0;
    bool want = false, above = false;
    int hl = 0, hj = 0;
    if (tid < 64) { e = ebase + (size_t)VU_W * 64 + 2 * (size_t)tid; want = true; hl = tid; hj = VU_W; }
    else if (tid < 64 + VU_W) {
      const int j = tid - 64 + 1, x = t0 - 64 + j;
      hl = 64; hj = j;
      // (fluid or not is read off the count grid here: on a row slab that row belongs to the rank above - its pressures arrived as a ghost row, its mask bytes did not)
      if (x >= 0 && x < X && band + 1 < g.nbands && count[(size_t)(band + 1) * 64 * X + x] != 0) { e = ((size_t)(band + 1) * g.TS + (size_t)(x & ~1)) * 64 + (size_t)(x & 1); want = true; above = true; }
    }
    if (tid < 64 + VU_W) {
      double pv = 0.0;
      if (want && (above || (mask[e] & CM_FLUID))) {
        pv = p[e];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < cnt) pv = pv + sq[j][e] * al[j];
        pv = pv < 0.0 ? 0.0 : pv;
      }
      sp[hl][hj] = pv;
    }
  }
  __syncthreads();
  const float neg_inv = -(1.f / (EU_RHO * EU_H));   // accel(), main.c:705-707
  float mu = 0.f, mv = 0.f;
  static_assert((64 * VU_W) % (256 * 4) == 0, "whole rounds of four cells per thread");
  for (int c0 = tid; c0 < 64 * VU_W; c0 += 256 * 4) {
    // four cells per thread, every load of the round issued before the first use (cells beyond the grid's edge read cell 0 and write nothing)
    size_t ii[4]; bool in[4]; int ll[4], jj[4], xx[4], yy[4];
    uint8_t c_0[4], c_r[4], c_u[4], s_0[4], s_r[4], s_u[4];
    float ui[4], vi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 256 * u;
      ll[u] = c / VU_W; jj[u] = c % VU_W;
      xx[u] = t0 - ll[u] + jj[u]; yy[u] = band * 64 + ll[u];
      in[u] = xx[u] >= 0 && xx[u] < X && yy[u] < Y;
      ii[u] = (size_t)(yy[u] < Y ? yy[u] : Y - 1) * X + (size_t)(xx[u] < 0 ? 0 : xx[u] >= X ? X - 1 : xx[u]);      // (beyond the edge: a cell of the own row, read and not used)
      const bool hr = in[u] && xx[u] < X - 1, hu = in[u] && yy[u] < Y - 1;
      c_0[u] = count[ii[u]]; s_0[u] = solid[ii[u]];
      c_r[u] = count[ii[u] + (hr ? 1 : 0)]; s_r[u] = solid[ii[u] + (hr ? 1 : 0)];
      c_u[u] = count[ii[u] + (hu ? (size_t)X : (size_t)0)]; s_u[u] = solid[ii[u] + (hu ? (size_t)X : (size_t)0)];
    }
    bool wet = false;      // some face of these cells has fluid on a side: only then is an old velocity read (main.c:786-789, 799-802: the air's faces are set to 0)
#pragma unroll
    for (int u = 0; u < 4; ++u) wet = wet || ((c_0[u] | c_r[u] | c_u[u]) != 0);
    if (__any(wet)) {      // (uniform per wave)
#pragma unroll
      for (int u = 0; u < 4; ++u) { ui[u] = uin[ii[u]]; vi[u] = vin[ii[u]]; }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) { ui[u] = 0.f; vi[u] = 0.f; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!in[u]) continue;
      const int l = ll[u], j = jj[u];
      const bool f0 = c_0[u] != 0;
      const double p0 = sp[l][j];
      if (xx[u] < X - 1) {
        const bool f1 = c_r[u] != 0;
        float o = 0.f;
        bool live = false;
        if (s_0[u] | s_r[u]) o = 0.f;
        else if (f0 | f1) { o = ui[u] + (neg_inv * (float)(sp[l][j + 1] - p0)) * dt; live = true; }
        if (live || !skip_zero) uout[ii[u]] = o;
        const float sqv = o * o; if (sqv > mu) mu = sqv;
      }
      if (yy[u] < Y - 1) {
        const bool f1 = c_u[u] != 0;
        float o = 0.f;
        bool live = false;
        if (s_0[u] | s_u[u]) o = 0.f;
        else if (f0 | f1) { o = vi[u] + (neg_inv * (float)(sp[l + 1][j + 1] - p0)) * dt; live = true; }
        if (live || !skip_zero) vout[ii[u]] = o;
        const float sqv = o * o; if (sqv > mv) mv = sqv;
      }
    }
  }
  // maxsq (main.c:808-820) of what was just written: exact, order-free maxima; k_dt forms dt from them at the next timestep
  mu = eu_wave_maxf(mu);
  mv = eu_wave_maxf(mv);
  if ((tid & 63) == 0) { s_mu[tid >> 6] = mu; s_mv[tid >> 6] = mv; }
  __syncthreads();
  if (tid == 0 && do_max) {
    for (int k = 1; k < 4; ++k) { if (s_mu[k] > mu) mu = s_mu[k]; if (s_mv[k] > mv) mv = s_mv[k]; }
    if (mu > 0.f) atomicMax(&ms->max_u2_bits, __float_as_uint(mu));      // (non-negative floats order like their bit patterns; a NaN never wins: s > m is false)
    if (mv > 0.f) atomicMax(&ms->max_v2_bits, __float_as_uint(mv));
  }
}
// the clamp alone, in memory (eu_pressure_current)
__global__ __launch_bounds__(256) void k_clamp_p(double* __restrict__ p, const uint8_t* __restrict__ mask, size_t e_lo, size_t S) {
  for (size_t i = e_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < e_lo + S; i += (size_t)gridDim.x * blockDim.x)
    if ((mask[i] & CM_FLUID) && p[i] < 0.0) p[i] = 0.0;
}

int eu_launch_band_ranges(euler_sim* S);

// the top bit of a listed chunk says INTERIOR (k_search_apply / k_precond_tile: no mask loads, constant coefficients, E^-1 from the table)
__global__ __launch_bounds__(256) void k_mark_interior(unsigned int* __restrict__ list, const PcgScalars* sc, const uint8_t* __restrict__ part) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sc->n_chunks) return;
  const unsigned int c = list[i];
  if (!part[c]) list[i] = c | EU_CHUNK_INTERIOR;
}

int eu_launch_build_system(euler_sim* S, float dt) {
  // the flags of the previous solve stay readable (which chunks may hold stale masks / p / r): two sets, swapped per solve
  uint8_t* t = S->chunk_flag; S->chunk_flag = S->chunk_prev; S->chunk_prev = t;
  HIPCHK(hipMemsetAsync(S->chunk_flag, 0, S->chunk_cap + 64, S->stream));
  HIPCHK(hipMemsetAsync(S->chunk_part, 0, S->chunk_cap + 64, S->stream));
  // the lean assembly needs every solve since the arrays were last written whole to have been a tile-mode solve of this handle
  const bool lean = eu_is_tile(S) && S->cfg.sweep_mode != EULER_SWEEP_SIMPLE && S->lean_ok;
  const bool gather = S->opt[EULER_OPT_BUILD_GATHER] != 0;      // (experiments: the one-kernel diagonal gather of rounds 1-2)
  const bool two_pass = S->opt[EULER_OPT_BUILD_TWO_PASS] != 0;  // (rounds 3-5: a row-major pass and a skewed gather)
  if (!gather && !two_pass) {
    const unsigned nblk = (unsigned)((size_t)(S->band_hi - S->band_lo) * (S->geom.T / BS_W));
    if (lean)
      LAUNCH(S, KC_BUILD_SYSTEM, k_build_system_para<true>, dim3(nblk), dim3(256), S->utmp, S->vtmp, S->count, S->solid, S->b, S->r, S->p, S->q, S->z, S->cellmask, S->sc, S->geom, dt, S->band_lo,
             S->chunk_flag, S->chunk_part, S->chunk_prev, eu_tile_map_on(S) ? (const uint8_t*)S->tmap : (const uint8_t*)nullptr, S->tmap_nx);
    else
      LAUNCH(S, KC_BUILD_SYSTEM, k_build_system_para<false>, dim3(nblk), dim3(256), S->utmp, S->vtmp, S->count, S->solid, S->b, S->r, S->p, S->q, S->z, S->cellmask, S->sc, S->geom, dt, S->band_lo,
             S->chunk_flag, S->chunk_part, S->chunk_prev, (const uint8_t*)nullptr, 0);
  } else if (!gather) {
    const size_t win_off = (size_t)S->win_lo * S->X;
    LAUNCH(S, KC_BUILD_SYSTEM, k_cell_system, dim3(eu_blocks((size_t)S->X * (S->row_hi - S->row_lo), 256)), dim3(256), S->utmp, S->vtmp, S->count, S->solid, S->sys_m, S->sys_div,
           S->X, S->row_lo, S->row_hi, win_off);
    if (lean)
      LAUNCH(S, KC_BUILD_SYSTEM, k_build_system2<true>, dim3(eu_blocks(S->e_cnt, 256)), dim3(256), S->sys_m, S->sys_div, win_off, S->b, S->r, S->p, S->q, S->z, S->cellmask, S->sc,
             S->geom, dt, S->e_lo, S->e_cnt, S->chunk_flag, S->chunk_part, S->chunk_prev);
    else
      LAUNCH(S, KC_BUILD_SYSTEM, k_build_system2<false>, dim3(eu_blocks(S->e_cnt, 256)), dim3(256), S->sys_m, S->sys_div, win_off, S->b, S->r, S->p, S->q, S->z, S->cellmask, S->sc,
             S->geom, dt, S->e_lo, S->e_cnt, S->chunk_flag, S->chunk_part, S->chunk_prev);
  } else if (lean)
    LAUNCH(S, KC_BUILD_SYSTEM, k_build_system<true>, dim3(eu_blocks(S->e_cnt, 256)), dim3(256), S->utmp, S->vtmp, S->count,
           S->solid, S->b, S->r, S->p, S->q, S->z, S->cellmask, S->sc, S->geom, dt, S->e_lo, S->e_cnt, S->chunk_flag, S->chunk_part, S->chunk_prev);
  else
    LAUNCH(S, KC_BUILD_SYSTEM, k_build_system<false>, dim3(eu_blocks(S->e_cnt, 256)), dim3(256), S->utmp, S->vtmp, S->count,
           S->solid, S->b, S->r, S->p, S->q, S->z, S->cellmask, S->sc, S->geom, dt, S->e_lo, S->e_cnt, S->chunk_flag, S->chunk_part, S->chunk_prev);
  S->lean_ok = 1;      // this solve's flags describe everything that is non-zero from here on
  LAUNCH(S, KC_BUILD_SYSTEM, k_pack_chunk_bits, dim3(eu_blocks(S->chunk_words, 256)), dim3(256), S->chunk_flag, S->chunk_bits, S->chunk_words, S->chunk_cap);
  // the ascending list of this solve's active chunks (k_search_apply, k_precond_tile)
  int rc = eu_ordered_select(S, S->chunk_bits, S->chunk_words, S->chunk_list, &S->sc->n_chunks);
  if (rc) return rc;
  const bool no_interior = S->opt[EULER_OPT_NO_INTERIOR] != 0;      // (experiments: every chunk takes the general path)
  if (!no_interior)
    LAUNCH(S, KC_BUILD_SYSTEM, k_mark_interior, dim3(eu_blocks(S->chunk_cap, 256)), dim3(256), S->chunk_list, S->sc, S->chunk_part);
  return eu_launch_band_ranges(S);
}

int eu_launch_finish_p(euler_sim* S);      // k_pcg.hip: the p += alpha s still due, in memory
// The finished, clamped pressure in memory: k_velocity_update_para forms it in LDS only.  Whoever reads S->p between two solves comes through here first.
int eu_pressure_current(euler_sim* S) {
  if (!S->p_pending) return EULER_OK;
  const int mode = S->p_pending;
  S->p_pending = 0;
  if (mode == 1) { int rc = eu_launch_finish_p(S); if (rc) return rc; }
  LAUNCH(S, KC_VELOCITY_UPDATE, k_clamp_p, dim3(eu_blocks(S->e_cnt, 256 * 4, 4096)), dim3(256), S->p, S->cellmask, S->e_lo, S->e_cnt);
  return EULER_OK;
}
// finish: 1 - the multi-kernel solve left its last p += alpha s undone (the ring of eu_launch_project is current); 0 - p is final (resident solver)
int eu_launch_velocity_update(euler_sim* S, float dt, int finish) {
  // (row slabs: the finished pressure is in memory - the ghost row of the rank above came out of it - so the pass has nothing to finish, only to clamp; the round-1 layout,
  // a communicator without row slabs, keeps the old kernel: its ranks hold the whole grid's p but only their own bands' masks)
  const bool one_pass = (!S->has_comm || S->slab_on) && S->opt[EULER_OPT_VELOCITY_TWO_PASS] == 0 && (!S->slab_on || !finish);
  if (one_pass) {
    VuRing ring;
    for (int k = 0; k < 8; ++k) ring.s[k] = S->s_ring[k < S->s_ring_n ? k : 0];
    ring.n = S->s_ring_n > 0 ? S->s_ring_n : 1; ring.steps = ring.n; ring.use = finish && S->s_ring_n > 0;
    if (!ring.use) for (int k = 0; k < 8; ++k) ring.s[k] = S->p;      // (never read)
    const int do_max = S->slab_on ? 0 : 1;
    if (do_max && S->maxsq_state != 0) HIPCHK(hipMemsetAsync(&S->ms->max_u2_bits, 0, 2 * sizeof(unsigned int), S->stream));      // (maxima no timestep consumed)
    LAUNCH(S, KC_VELOCITY_UPDATE, k_velocity_update_para, dim3((unsigned)((size_t)(S->band_hi - S->band_lo) * (S->geom.T / VU_W))), dim3(256), S->utmp, S->vtmp, S->u, S->v, S->p, S->cellmask,
           S->count, S->solid, S->geom, dt, S->sc, ring, S->ms, S->band_lo, do_max, (!S->slab_on && S->uv_zb) ? 1 : 0,
           eu_tile_map_on(S) ? (const uint8_t*)S->tmap : (const uint8_t*)nullptr, S->tmap_nx);
    S->uv_clean = 1;      // every sample without the fluid property, every wall's sample is zero now (k_zero_bounds4<true> of the next substep relies on it)
    S->p_pending = ring.use ? 1 : 2;
    if (do_max) S->maxsq_state = 2;      // the maxima of u, v as they stand are in ms (eu_launch_timestep: k_dt alone)
    return EULER_OK;
  }
  if (finish) { int rc = eu_launch_finish_p(S); if (rc) return rc; }
  dim3 grid((S->X + 63) / 64, (S->row_hi - S->row_lo + 3) / 4);
  LAUNCH(S, KC_VELOCITY_UPDATE, k_velocity_update, grid, dim3(256), S->utmp, S->vtmp, S->u, S->v, S->p, S->count,
         S->solid, S->geom, dt, S->row_lo, S->row_hi);
  return EULER_OK;
}

// ------------------------------------------------------------------------------------------
// skewed <-> row-major conversion for euler_get_field / euler_set_field (tests, render of p)
// (rowmajor holds this rank's rows only, starting at row y0)
template <typename T>
__global__ __launch_bounds__(256) void k_unskew(const T* __restrict__ skew, T* __restrict__ rowmajor, SkewGeom g, int y0, int y1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)g.X * (y1 - y0)) rowmajor[i] = skew[skew_index(g, (int)(i % g.X), y0 + (int)(i / g.X))];
}
template <typename T>
__global__ __launch_bounds__(256) void k_skew(const T* __restrict__ rowmajor, T* __restrict__ skew, SkewGeom g, int y0, int y1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)g.X * (y1 - y0)) skew[skew_index(g, (int)(i % g.X), y0 + (int)(i / g.X))] = rowmajor[i];
}

// a solver vector whose non-fluid elements are not maintained (k_pcg.hip: the search directions' ring): they read as +0
__global__ __launch_bounds__(256) void k_unskew_fluid(const double* __restrict__ skew, const uint8_t* __restrict__ mask, double* __restrict__ rowmajor, SkewGeom g, int y0, int y1) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)g.X * (y1 - y0)) return;
  const size_t e = skew_index(g, (int)(i % g.X), y0 + (int)(i / g.X));
  rowmajor[i] = (mask[e] & CM_FLUID) ? skew[e] : 0.0;
}
int eu_unskew_fluid(euler_sim* S, const double* skew, double* rowmajor) {
  const unsigned nb = eu_blocks((size_t)S->X * (S->row_hi - S->row_lo), 256);
  hipLaunchKernelGGL(k_unskew_fluid, dim3(nb), dim3(256), 0, S->stream, skew, S->cellmask, rowmajor, S->geom, S->row_lo, S->row_hi);
  return EULER_OK;
}
int eu_unskew(euler_sim* S, const void* skew, void* rowmajor, int elem_bytes) {
  const unsigned nb = eu_blocks((size_t)S->X * (S->row_hi - S->row_lo), 256);
  if (elem_bytes == 8) hipLaunchKernelGGL(k_unskew<double>, dim3(nb), dim3(256), 0, S->stream, (const double*)skew, (double*)rowmajor, S->geom, S->row_lo, S->row_hi);
  else hipLaunchKernelGGL(k_unskew<uint8_t>, dim3(nb), dim3(256), 0, S->stream, (const uint8_t*)skew, (uint8_t*)rowmajor, S->geom, S->row_lo, S->row_hi);
  return EULER_OK;
}
int eu_skew(euler_sim* S, const void* rowmajor, void* skew, int elem_bytes) {
  const unsigned nb = eu_blocks((size_t)S->X * (S->row_hi - S->row_lo), 256);
  if (elem_bytes == 8) hipLaunchKernelGGL(k_skew<double>, dim3(nb), dim3(256), 0, S->stream, (const double*)rowmajor, (double*)skew, S->geom, S->row_lo, S->row_hi);
  else hipLaunchKernelGGL(k_skew<uint8_t>, dim3(nb), dim3(256), 0, S->stream, (const uint8_t*)rowmajor, (uint8_t*)skew, S->geom, S->row_lo, S->row_hi);
  return EULER_OK;
}
