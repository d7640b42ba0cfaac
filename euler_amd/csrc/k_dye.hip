// k_dye.hip — the `--rainbow` dye (reference main.c:75-83): three cell-centred colour fields that are
// coloured at init (colorize, main.c:187-201), extended into new fluid cells (extrapolate(.., P),
// main.c:859-863), refreshed at source cells (main.c:283,292-294) and advected with the flow
// (advect_p + the whole-array memcpy, main.c:424-438, 873-882).  Cosmetic: nothing here feeds back
// into the flow.  Element-wise gathers over row-major arrays, the three channels share one launch.
#include "euler_dev.h"

// misc/color.h:16-34: period 6, values in [0, 1]
__device__ __forceinline__ float hsv_basis(float t) {
  t -= 6.f * floorf(1.f / 6 * t);
  if (t < 0.f) t += 6.f;
  if (t < 1.f) return t;
  if (t < 3.f) return 1.f;
  if (t < 4.f) return 4.f - t;
  return 0.f;
}

// colorize (main.c:187-201): fluid cells only; k_initial_color_period = 60 cells (main.c:83)
__global__ __launch_bounds__(256) void k_colorize(float* r, float* g, float* b, const uint8_t* __restrict__ count,
                                                  const uint8_t* __restrict__ source, int X, int y0, int y1) {   // rows [y0, y1) of this rank
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  const size_t i = (size_t)y * X + x;
  if (!count[i]) return;
  float t = 0.f;
  if (!source[i]) t = (x + y) * 6.f / 60.f;
  r[i] = hsv_basis(t + 2.f);
  g[i] = hsv_basis(t);
  b[i] = hsv_basis(t - 2.f);
}

// extrapolate(q, P) (main.c:158-185) for the three channels: a cell that just became fluid takes the mean
// of its previously-fluid 3x3 neighbours, summed y-major then x.  In place is race-free (writes go to cells
// without the prev-fluid property, reads to cells with it).
__global__ __launch_bounds__(256) void k_extrapolate_dye(float* r, float* g, float* b, const uint8_t* __restrict__ prev,
                                                         const uint8_t* __restrict__ cur, int X, int Y, int ry0, int ry1) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = ry0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= ry1) return;
  const size_t i = (size_t)y * X + x;
  if (prev[i] || !cur[i]) return;
  const int x0 = x > 0 ? x - 1 : 0, x1 = x + 1 < X ? x + 1 : X - 1;
  const int y0 = y > 0 ? y - 1 : 0, y1 = y + 1 < Y ? y + 1 : Y - 1;
  float tr = 0.f, tg = 0.f, tb = 0.f;
  int n = 0;
  for (int yy = y0; yy <= y1; ++yy)
    for (int xx = x0; xx <= x1; ++xx) {
      const size_t j = (size_t)yy * X + xx;
      if (prev[j]) { tr += r[j]; tg += g[j]; tb += b[j]; ++n; }
    }
  r[i] = tr / n; g[i] = tg / n; b[i] = tb / n;   // n == 0 -> 0/0 like the Release reference
}

// the source colour (main.c:283,292-294): every source cell, every substep
__global__ __launch_bounds__(256) void k_dye_sources(float* r, float* g, float* b, const uint8_t* __restrict__ source,
                                                     size_t i0, size_t C, float t) {   // cells [i0, i0 + C): this rank's rows
  const size_t i = i0 + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= i0 + C || !source[i]) return;
  r[i] = hsv_basis(t + 2.f);
  g[i] = hsv_basis(t);
  b[i] = hsv_basis(t - 2.f);
}

// advect_p (main.c:424-438) for the three channels; only fluid cells of the outputs are written
__global__ __launch_bounds__(256) void k_advect_dye(const float* __restrict__ r, const float* __restrict__ g,
                                                    const float* __restrict__ b, float* __restrict__ rout,
                                                    float* __restrict__ gout, float* __restrict__ bout,
                                                    const float* __restrict__ u, const float* __restrict__ v, GridRef gr, float dt, int y0, int y1) {
  const int X = gr.X;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= X || y >= y1) return;
  const size_t i = (size_t)y * X + x;
  if (!gr.count[i]) return;                     // never fluid on the border ring (all sink): i - X, i - 1 exist
  const float dy = (v[i] + v[i - X]) / 2;
  const float dx = (u[i] + u[i - 1]) / 2;
  const float px = x - dx * dt / EU_H, py = y - dy * dt / EU_H;
  rout[i] = eu_interp<0>(gr, r, px, py);
  gout[i] = eu_interp<0>(gr, g, px, py);
  bout[i] = eu_interp<0>(gr, b, px, py);
}

static inline dim3 cell_grid(const euler_sim* S) { return dim3((S->X + 63) / 64, (S->row_hi - S->row_lo + 3) / 4); }   // this rank's rows (all of them without slabs)

int eu_launch_colorize(euler_sim* S) {
  if (!S->dye[0]) return EULER_OK;
  LAUNCH(S, KC_MISC, k_colorize, cell_grid(S), dim3(256), S->dye[0], S->dye[1], S->dye[2], S->count, S->source, S->X, S->row_lo, S->row_hi);
  return EULER_OK;
}

int eu_launch_dye_extrapolate(euler_sim* S) {
  if (!S->dye[0]) return EULER_OK;
  LAUNCH(S, KC_EXTRAPOLATE, k_extrapolate_dye, cell_grid(S), dim3(256), S->dye[0], S->dye[1], S->dye[2], S->prev_count, S->count, S->X, S->Y, S->row_lo, S->row_hi);
  return EULER_OK;
}

int eu_launch_dye_sources(euler_sim* S) {
  if (!S->dye[0] || S->n_source_cells == 0) return EULER_OK;
  // k_source_color_period = 10 s (main.c:82); g_frame_count is a uint16_t (main.c:88) = frames completed so far
  const float t = 0.6f / 10.f * (uint16_t)S->stats.frames;
  const size_t i0 = (size_t)S->row_lo * S->X, n = (size_t)(S->row_hi - S->row_lo) * S->X;
  LAUNCH(S, KC_SOURCES, k_dye_sources, dim3(eu_blocks(n, 256)), dim3(256), S->dye[0], S->dye[1], S->dye[2], S->source, i0, n, t);
  return EULER_OK;
}

int eu_launch_dye_advect(euler_sim* S, float dt) {
  if (!S->dye[0]) return EULER_OK;
  GridRef g{S->X, S->Y, S->count, S->interp_lim[0], S->interp_lim[1], S->interp_lim[2], S->interp_lim[3]};
  LAUNCH(S, KC_ADVECT_VELOCITY, k_advect_dye, cell_grid(S), dim3(256), S->dye[0], S->dye[1], S->dye[2], S->dye[3], S->dye[4], S->dye[5],
         S->u, S->v, g, dt, S->row_lo, S->row_hi);
  // memcpy(g_r, g_rtmp, sizeof(g_r)) x3 (main.c:875,878,881): the WHOLE scratch array, stale non-fluid entries included (a row slab: its own rows)
  const size_t o = (size_t)S->row_lo * S->X, n = (size_t)(S->row_hi - S->row_lo) * S->X;
  for (int k = 0; k < 3; ++k) HIPCHK(hipMemcpyAsync(S->dye[k] + o, S->dye[3 + k] + o, n * sizeof(float), hipMemcpyDeviceToDevice, S->stream));
  return EULER_OK;
}
