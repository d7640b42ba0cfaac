/*
 * euler_oracle.c — TEST ORACLE, not product code (see euler_oracle.h).
 *
 * CPU restatement of the cgmb/euler per-frame simulation path with a run-time grid.  Written
 * from the behaviour of the reference (file:line cited per function), keeping the per-cell
 * operation order so that, compiled without FMA contraction and without -ffast-math, it is
 * bit-identical to the compiled reference at X=100,Y=40 (checked in tests/).
 *
 * Conventions: every field is a flat row-major [Y][X] array, index i = y*X + x.  A U sample
 * (x,y) sits between P cells (x,y) and (x+1,y); a V sample between (x,y) and (x,y+1).
 */
#include "euler_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference constants, main.c:58-60 */
static const float H_CELL  = 1.f;   /* k_side_length */
static const float DENSITY = 1.f;   /* k_density */
static const float GRAVITY = -10.f; /* k_gravity */

#define AT(s, y, x) ((size_t)(y) * (size_t)(s)->X + (size_t)(x))

/* ---------------------------------------------------------------- allocation */

eo_sim* eo_create(int X, int Y) {
  if (X < 4 || Y < 4) return NULL;
  eo_sim* s = (eo_sim*)calloc(1, sizeof(eo_sim));
  if (!s) return NULL;
  size_t C = (size_t)X * (size_t)Y;
  s->X = X; s->Y = Y;
  s->u = calloc(C, sizeof(float));    s->v = calloc(C, sizeof(float));
  s->utmp = calloc(C, sizeof(float)); s->vtmp = calloc(C, sizeof(float));
  s->solid = calloc(C, 1); s->source = calloc(C, 1); s->sink = calloc(C, 1);
  s->count = calloc(C, 1); s->prev_count = calloc(C, 1);
  s->max_markers = 4 * C;
  s->markers = calloc(s->max_markers, sizeof(eo_vec2f));
  s->a_diag = calloc(C, 1);
  s->precon = calloc(C, sizeof(double)); s->q = calloc(C, sizeof(double));
  s->b = calloc(C, sizeof(double)); s->p = calloc(C, sizeof(double));
  s->r = calloc(C, sizeof(double)); s->z = calloc(C, sizeof(double));
  s->s = calloc(C, sizeof(double));
  s->cr = calloc(C, sizeof(float)); s->cg = calloc(C, sizeof(float)); s->cb = calloc(C, sizeof(float));
  s->crtmp = calloc(C, sizeof(float)); s->cgtmp = calloc(C, sizeof(float)); s->cbtmp = calloc(C, sizeof(float));
  s->rng_state = 0x9bd185c449534b91ull; /* main.c:204 */
  s->max_iterations = 100;              /* main.c:735 */
  s->tol = (double)1e-6f;               /* main.c:736: a float literal widened */
  return s;
}

static void mg_free(eo_sim* s);
void eo_destroy(eo_sim* s) {
  if (!s) return;
  free(s->u); free(s->v); free(s->utmp); free(s->vtmp);
  free(s->solid); free(s->source); free(s->sink); free(s->count); free(s->prev_count);
  free(s->markers); free(s->a_diag); free(s->precon); free(s->q);
  free(s->b); free(s->p); free(s->r); free(s->z); free(s->s);
  free(s->cr); free(s->cg); free(s->cb); free(s->crtmp); free(s->cgtmp); free(s->cbtmp);
  free(s->coarse_chol); free(s->coarse_null);
  mg_free(s);
  free(s);
}

/* ---------------------------------------------------------------- rng (misc/rng.c:5-20, main.c:203-207) */

static float eo_randf(eo_sim* s) {
  uint64_t x = s->rng_state;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  s->rng_state = x;
  uint32_t hi = (uint32_t)((x * 0x2545F4914F6CDD1Dull) >> 32);
  return (float)(hi / (double)UINT32_MAX); /* closed interval [0,1] */
}

/* ---------------------------------------------------------------- typed cell properties (main.c:119-156) */

static inline int prop(const eo_sim* s, const uint8_t* g, int x, int y, int type) {
  size_t i = AT(s, y, x);
  switch (type) {
    case EO_U: return (g[i] != 0) | (g[i + 1] != 0);
    case EO_V: return (g[i] != 0) | (g[i + (size_t)s->X] != 0);
    default:   return g[i] != 0;
  }
}
static inline int ext_x(const eo_sim* s, int type) { return type == EO_U ? s->X - 1 : s->X; }
static inline int ext_y(const eo_sim* s, int type) { return type == EO_V ? s->Y - 1 : s->Y; }

/* ---------------------------------------------------------------- marker binning (main.c:102-117) */

void eo_refresh_marker_counts(eo_sim* s) {
  size_t C = (size_t)s->X * (size_t)s->Y;
  memcpy(s->prev_count, s->count, C);
  memset(s->count, 0, C);
  size_t i = 0;
  while (i < s->n_markers) {
    int x = (int)floorf(s->markers[i].x / H_CELL);
    int y = (int)floorf(s->markers[i].y / H_CELL);
    size_t c = AT(s, y, x);
    if (s->sink[c] || s->solid[c]) {
      /* swap-with-last; the swapped-in marker is examined next (main.c:112) */
      s->markers[i] = s->markers[--s->n_markers];
    } else {
      s->count[c]++; /* uint8 wrap is part of the semantics */
      ++i;
    }
  }
}

/* ---------------------------------------------------------------- scenario (main.c:209-274) */

static void finish_init(eo_sim* s, const uint8_t* fluid) {
  int X = s->X, Y = s->Y;
  for (int y = 0; y < Y; ++y) { s->sink[AT(s, y, 0)] = 1; s->sink[AT(s, y, X - 1)] = 1; }
  for (int x = 0; x < X; ++x) { s->sink[AT(s, 0, x)] = 1; s->sink[AT(s, Y - 1, x)] = 1; }
  /* 4 jittered markers per fluid cell, x outer / y inner, x drawn before y (main.c:255-266) */
  size_t n = 0;
  for (int i = 0; i < X; ++i)
    for (int j = 0; j < Y; ++j)
      if (fluid[AT(s, j, i)])
        for (int k = 0; k < 4; ++k) {
          float fx = i + (k < 2 ? 0 : 0.5f) + (eo_randf(s) / 2);
          float fy = j + (k % 2 ? 0 : 0.5f) + (eo_randf(s) / 2);
          s->markers[n].x = H_CELL * fx;
          s->markers[n].y = H_CELL * fy;
          ++n;
        }
  s->n_markers = n;
  eo_refresh_marker_counts(s);
  if (s->rainbow) eo_colorize(s);   /* main.c:270-273; set s->rainbow before loading */
}

static void set_cell(eo_sim* s, uint8_t* fluid, int x, int y, char c) {
  size_t i = AT(s, y, x);
  if (c == 'X') s->solid[i] = 1;
  else if (c == '0') fluid[i] = 1;
  else if (c == '?') { fluid[i] = 1; s->source[i] = 1; }
  else if (c == '=') s->sink[i] = 1;
}

int eo_load_scenario_mem(eo_sim* s, const char* text, int len, int upscale) {
  int X = s->X, Y = s->Y;
  size_t C = (size_t)X * (size_t)Y;
  uint8_t* fluid = calloc(C, 1);
  if (!fluid) return -1;
  if (!upscale) {
    /* streaming parse: first line -> y=Y-2, first char -> x=1; over-long lines discarded */
    int i = 0;
    for (int y = Y - 2; y > 0 && i < len; --y) {
      int x;
      for (x = 1; x < X - 1 && i < len; ++x) {
        char c = text[i++];
        if (c == '\n') break;
        set_cell(s, fluid, x, y, c);
      }
      if (x == X - 1) { while (i < len && text[i++] != '\n') {} }
    }
  } else {
    /* ch(x,y) = file[floor((Y-2-y)*Hf/(Y-2))][floor((x-1)*Wf/(X-2))] */
    int nlines = 0, Wf = 0;
    for (int i = 0, w = 0; i <= len; ++i) {
      if (i == len || text[i] == '\n') {
        if (i < len || w > 0) { nlines++; if (w > Wf) Wf = w; }
        w = 0;
      } else w++;
    }
    if (nlines == 0 || Wf == 0) { free(fluid); return -2; }
    const char** line = malloc(sizeof(char*) * (size_t)nlines);
    int* linelen = malloc(sizeof(int) * (size_t)nlines);
    int k = 0, st = 0;
    for (int i = 0; i <= len; ++i)
      if (i == len || text[i] == '\n') {
        if (i < len || i > st) { line[k] = text + st; linelen[k] = i - st; k++; }
        st = i + 1;
      }
    for (int y = 1; y <= Y - 2; ++y) {
      long fr = ((long)(Y - 2 - y) * nlines) / (Y - 2);
      for (int x = 1; x <= X - 2; ++x) {
        long fc = ((long)(x - 1) * Wf) / (X - 2);
        char c = (fc < linelen[fr]) ? line[fr][fc] : ' ';
        set_cell(s, fluid, x, y, c);
      }
    }
    free(line); free(linelen);
  }
  finish_init(s, fluid);
  free(fluid);
  return 0;
}

int eo_load_scenario_file(eo_sim* s, const char* path, int upscale) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char* buf = malloc((size_t)n + 1);
  if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return -1; }
  fclose(f);
  int rc = eo_load_scenario_mem(s, buf, (int)n, upscale);
  free(buf);
  return rc;
}

int eo_load_half_tank(eo_sim* s) {
  int X = s->X, Y = s->Y;
  uint8_t* fluid = calloc((size_t)X * (size_t)Y, 1);
  if (!fluid) return -1;
  for (int y = 1; y <= Y - 2; ++y)
    for (int x = 1; x <= X - 2; ++x) {
      size_t i = AT(s, y, x);
      if (y == 1 || y == Y - 2 || x == 1 || x == X - 2) s->solid[i] = 1;
      else if (y < Y / 2) fluid[i] = 1;
    }
  finish_init(s, fluid);
  free(fluid);
  return 0;
}

/* ---------------------------------------------------------------- dye colours (misc/color.h:16-34, main.c:187-201) */

static float hsv_basis(float t) {   /* misc/color.h:16-34: period 6, values in [0,1] */
  t -= 6.f * floorf(1.f / 6 * t);
  if (t < 0.f) t += 6.f;
  if (t < 1.f) return t;
  if (t < 3.f) return 1.f;
  if (t < 4.f) return 4.f - t;
  return 0.f;
}

void eo_colorize(eo_sim* s) {   /* main.c:187-201; k_initial_color_period = 60 grid cells (main.c:83) */
  for (int y = 0; y < s->Y; ++y)
    for (int x = 0; x < s->X; ++x) {
      size_t i = AT(s, y, x);
      if (!s->count[i]) continue;
      float t = 0.f;
      if (!s->source[i]) t = (x + y) * 6.f / 60.f;
      s->cr[i] = hsv_basis(t + 2.f);
      s->cg[i] = hsv_basis(t);
      s->cb[i] = hsv_basis(t - 2.f);
    }
}

/* ---------------------------------------------------------------- sources (main.c:276-298) */

void eo_update_fluid_sources(eo_sim* s) {
  int X = s->X, Y = s->Y;
  s->source_exhausted |= (s->n_markers == s->max_markers - 1);
  /* main.c:283: k_source_color_period = 10 s (main.c:82); g_frame_count is a uint16_t (main.c:88) */
  float t = 0.6f / 10.f * (uint16_t)s->frame_count;
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      size_t i = AT(s, y, x);
      if (s->source[i]) {   /* main.c:292-294: written whether or not the dye is shown */
        s->cr[i] = hsv_basis(t + 2.f);
        s->cg[i] = hsv_basis(t);
        s->cb[i] = hsv_basis(t - 2.f);
      }
      if (s->source[i] && !s->source_exhausted && s->count[i] < 4) {
        /* main.c:288 v2f(x+randf(), y+randf()): argument evaluation order is unspecified in C;
         * the compiled reference (gcc, x86-64) evaluates right-to-left, so y is drawn FIRST.
         * The golden fixtures pin this order. */
        float ry = eo_randf(s), rx = eo_randf(s);
        s->markers[s->n_markers].x = H_CELL * (x + rx);
        s->markers[s->n_markers].y = H_CELL * (y + ry);
        s->n_markers++;
        s->count[i]++;
        s->source_exhausted |= (s->n_markers == s->max_markers - 1);
      }
    }
}

/* ---------------------------------------------------------------- extrapolation (main.c:158-185) */

void eo_extrapolate(eo_sim* s, float* q, int type) {
  int ex = ext_x(s, type), ey = ext_y(s, type);
  for (int y = 0; y < ey; ++y)
    for (int x = 0; x < ex; ++x) {
      if (prop(s, s->prev_count, x, y, type) || !prop(s, s->count, x, y, type)) continue;
      int x0 = x > 0 ? x - 1 : 0, x1 = x + 1 < ex ? x + 1 : ex - 1;
      int y0 = y > 0 ? y - 1 : 0, y1 = y + 1 < ey ? y + 1 : ey - 1;
      float total = 0.f; int n = 0;
      for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx)
          if (prop(s, s->prev_count, xx, yy, type)) { total += q[AT(s, yy, xx)]; n++; }
      q[AT(s, y, x)] = total / n; /* n==0 -> 0/0, as the Release reference (assert off) */
    }
}

/* ---------------------------------------------------------------- masked bilinear (main.c:300-364) */

static inline float lerp1(float x0, float x1, float f) { return (1.f - f) * x0 + f * x1; }
static inline float pick_frac(float f, int start_ok, int end_ok) {
  return !start_ok ? 1.f : (!end_ok ? 0.f : f);
}
static inline float clampf_(float lo, float x, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

float eo_interpolate(const eo_sim* s, const float* q, float ix, float iy, int type) {
  ix = clampf_(0, ix, nextafterf((float)(ext_x(s, type) - 1), 0));
  iy = clampf_(0, iy, nextafterf((float)(ext_y(s, type) - 1), 0));
  float wx, wy;
  float fx = modff(ix, &wx), fy = modff(iy, &wy);
  int bx = (int)wx, by = (int)wy;
  int v00 = prop(s, s->count, bx, by, type),     v01 = prop(s, s->count, bx + 1, by, type);
  int v10 = prop(s, s->count, bx, by + 1, type), v11 = prop(s, s->count, bx + 1, by + 1, type);
  float q00 = v00 ? q[AT(s, by, bx)] : 0.f,     q01 = v01 ? q[AT(s, by, bx + 1)] : 0.f;
  float q10 = v10 ? q[AT(s, by + 1, bx)] : 0.f, q11 = v11 ? q[AT(s, by + 1, bx + 1)] : 0.f;
  /* vertical lerp per column, then horizontal (main.c:322-330) */
  float lf = pick_frac(fy, v00, v10), rf = pick_frac(fy, v01, v11);
  float lv = lerp1(q00, q10, lf), rv = lerp1(q01, q11, rf);
  float hf = pick_frac(fx, v00 | v10, v01 | v11);
  return lerp1(lv, rv, hf);
}

/* ---------------------------------------------------------------- velocity advection (main.c:378-422) */

void eo_advect_u(const eo_sim* s, const float* u, const float* v, float dt, float* out) {
  for (int y = 0; y < s->Y; ++y)
    for (int x = 0; x < s->X - 1; ++x) {
      if (!prop(s, s->count, x, y, EO_U)) continue;
      float dx = u[AT(s, y, x)];
      float dy = eo_interpolate(s, v, x + 0.5f, y - 0.5f, EO_V);
      float px = x - dx * dt / H_CELL, py = y - dy * dt / H_CELL;
      out[AT(s, y, x)] = eo_interpolate(s, u, px, py, EO_U);
    }
}

void eo_advect_v(const eo_sim* s, const float* u, const float* v, float dt, float* out) {
  for (int y = 0; y < s->Y - 1; ++y)
    for (int x = 0; x < s->X; ++x) {
      if (!prop(s, s->count, x, y, EO_V)) continue;
      float dy = v[AT(s, y, x)];
      float dx = eo_interpolate(s, u, x - 0.5f, y + 0.5f, EO_U);
      float px = x - dx * dt / H_CELL, py = y - dy * dt / H_CELL;
      out[AT(s, y, x)] = eo_interpolate(s, v, px, py, EO_V);
    }
}

/* advect_p (main.c:424-438): cell-centred quantity, fluid cells only, the rest of `out` keeps its old content */
void eo_advect_p(const eo_sim* s, const float* q, const float* u, const float* v, float dt, float* out) {
  for (int y = 0; y < s->Y; ++y)
    for (int x = 0; x < s->X; ++x) {
      if (!s->count[AT(s, y, x)]) continue;   /* never true on the border ring: it is all sink */
      float dy = (v[AT(s, y, x)] + v[AT(s, y - 1, x)]) / 2;
      float dx = (u[AT(s, y, x)] + u[AT(s, y, x - 1)]) / 2;
      float px = x - dx * dt / H_CELL, py = y - dy * dt / H_CELL;
      out[AT(s, y, x)] = eo_interpolate(s, q, px, py, EO_P);
    }
}

/* ---------------------------------------------------------------- marker advection (main.c:440-537) */

static inline float time_to(float p0, float p1, float vel) {
  return fabsf(vel) > 0.f ? (p1 - p0) / vel : FLT_MAX;
}

/* NOTE (main.c:464,501,518): the reference subtracts t_prev from its *parameter* dt inside the
 * marker loop, so a collision that happens after a cell crossing shortens dt for every LATER
 * marker in the array.  Marker order is therefore observable; it is reproduced exactly here
 * (init order main.c:255-266, swap-with-last main.c:112, source appends main.c:288). */
void eo_advect_markers(eo_sim* s, float dt) {
  for (size_t m = 0; m < s->n_markers; ++m) {
    float px = s->markers[m].x, py = s->markers[m].y;
    float vx = eo_interpolate(s, s->u, px / H_CELL - 1.f, py / H_CELL - 0.5f, EO_U);
    float vy = eo_interpolate(s, s->v, px / H_CELL - 0.5f, py / H_CELL - 1.f, EO_V);
    int xi = (int)floorf(px / H_CELL), yi = (int)floorf(py / H_CELL);

    int xdir = vx > 0 ? 1 : -1, nxi = xi + (vx > 0 ? 1 : 0);
    float npx = nxi * H_CELL;
    float tx = time_to(px, npx, vx);
    int xoff = vx < 0 ? -1 : 0;

    int ydir = vy > 0 ? 1 : -1, nyi = yi + (vy > 0 ? 1 : 0);
    float npy = nyi * H_CELL;
    float ty = time_to(py, npy, vy);
    int yoff = vy < 0 ? -1 : 0;

    float t_prev = 0.f, t_near = fminf(tx, ty);
    while (t_near < dt) {
      if (tx < ty) {
        if (s->solid[AT(s, yi, nxi + xoff)]) {
          px = px + t_prev * vx; py = py + t_prev * vy;
          dt -= t_prev; t_near = 0; vx = 0.f; tx = FLT_MAX;
          ty = time_to(py, npy, vy);
        } else {
          xi = nxi; nxi = xi + xdir; npx = nxi * H_CELL;
          tx = time_to(px, npx, vx);
        }
      } else {
        if (s->solid[AT(s, nyi + yoff, xi)]) {
          px = px + t_prev * vx; py = py + t_prev * vy;
          dt -= t_prev; t_near = 0; vy = 0.f; ty = FLT_MAX;
          tx = time_to(px, npx, vx);
        } else {
          yi = nyi; nyi = yi + ydir; npy = nyi * H_CELL;
          ty = time_to(py, npy, vy);
        }
      }
      t_prev = t_near;
      t_near = fminf(tx, ty);
    }
    float t = (t_near < FLT_MAX) ? dt : t_prev;
    s->markers[m].x = px + t * vx;
    s->markers[m].y = py + t * vy;
  }
}

/* ---------------------------------------------------------------- forces, bounds, dt (main.c:539-545, 808-841) */

void eo_apply_body_forces(const eo_sim* s, float* v, float dt) {
  for (int y = 0; y < s->Y - 1; ++y)
    for (int x = 0; x < s->X; ++x) v[AT(s, y, x)] += GRAVITY * dt;
}

void eo_zero_bounds(const eo_sim* s, float* q, int type) {
  int ex = ext_x(s, type), ey = ext_y(s, type);
  for (int y = 0; y < ey; ++y)
    for (int x = 0; x < ex; ++x)
      if (!prop(s, s->count, x, y, type) || prop(s, s->solid, x, y, type)) q[AT(s, y, x)] = 0.f;
}

static float max_square(const eo_sim* s, const float* q, int type) {
  int ex = ext_x(s, type), ey = ext_y(s, type);
  float m = 0;
  for (int y = 0; y < ey; ++y)
    for (int x = 0; x < ex; ++x) { float sq = q[AT(s, y, x)] * q[AT(s, y, x)]; if (sq > m) m = sq; }
  return m;
}

float eo_calculate_timestep(const eo_sim* s, float frame_time) {
  const float max_distance = 0.75f * H_CELL;
  float vmax = sqrtf(max_square(s, s->u, EO_U) + max_square(s, s->v, EO_V));
  return fminf(max_distance / vmax, frame_time); /* 0.75/0 = +inf -> frame_time */
}

/* ---------------------------------------------------------------- pressure projection (main.c:547-806) */

#define FLUID(s, y, x) ((s)->count[AT(s, y, x)] != 0)

void eo_build_system(eo_sim* s, float dt, const float* u, const float* v) {
  free(s->coarse_chol); s->coarse_chol = NULL;      /* a new system: the coarse factor (two-level extension) is recomputed on demand */
  mg_free(s);
  int X = s->X, Y = s->Y;
  size_t C = (size_t)X * (size_t)Y;
  const double k_inv_scale = (H_CELL * H_CELL) * DENSITY / dt; /* float expr widened, main.c:713 */
  memset(s->b, 0, C * sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      double divergence = (u[i] - u[i - 1] + v[i] - v[i - X]) / H_CELL; /* float, then widened */
      s->b[i] = -divergence * k_inv_scale;
      s->a_diag[i] = (int8_t)(4 - s->solid[i - 1] - s->solid[i + 1] - s->solid[i - X] - s->solid[i + X]);
    }
}

/* EXTENSION (this build's, no reference counterpart): where the tile-local preconditioner cuts a band (euler_oracle.h) */
int eo_tile_start(int tile_records, int t) {
  return tile_records > 0 ? t % tile_records == 0 : t == 0;
}

/* Tile-local IC(0): the same three recurrences as below restricted to blocks.  A coupling that is cut contributes
 * the value the product's wavefront carries into a tile: precon 0 in the factor, the term -0.0 in the forward solve
 * (= (-1 * precon) * (+0)), z = +0 in the backward solve. */
static void apply_preconditioner_tiled(eo_sim* s, const double* r, double* z) {
  int X = s->X, Y = s->Y;
  size_t C = (size_t)X * (size_t)Y;
  double* pre = s->precon; double* q = s->q;
  uint8_t* start = (uint8_t*)malloc((size_t)X + 64 + 1);   /* start[t]: record t begins a tile */
  for (int t = 0; t <= X + 63; ++t) start[t] = (uint8_t)eo_tile_start(s->tile_records, t);
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      const int l = y & 63, t = x + l, cut = start[t];
      double a = s->a_diag[i];
      double cl = -1 * (cut ? 0.0 : pre[i - 1]);
      double cb = -1 * ((cut || l == 0) ? 0.0 : pre[i - X]);
      double e = a - cl * cl - cb * cb;
      if (e < 0.25 * a) e = a != 0 ? a : 1;
      pre[i] = 1 / sqrt(e);
    }
  memset(q, 0, C * sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      const int l = y & 63, t = x + l, cut = start[t];
      double t_ = r[i] - (cut ? -0.0 : -1 * pre[i - 1] * q[i - 1])
                       - ((cut || l == 0) ? -0.0 : -1 * pre[i - X] * q[i - X]);
      q[i] = t_ * pre[i];
    }
  memset(z, 0, C * sizeof(double));
  for (int y = Y; y--;)
    for (int x = X; x--;) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      const int l = y & 63, t = x + l, cut = start[t + 1];   /* the right / upper neighbour sits in record t + 1 */
      double t_ = q[i] - (FLUID(s, y, x + 1) ? -1 : 0) * pre[i] * (cut ? 0.0 : z[i + 1])
                       - (FLUID(s, y + 1, x) ? -1 : 0) * pre[i] * ((cut || l == 63) ? 0.0 : z[i + X]);
      z[i] = t_ * pre[i];
    }
  free(start);
}

/* EXTENSION (euler_oracle.h pcg_f32): the same three recurrences in FLOAT arithmetic - every operation's result is rounded to float (a double operation on
 * float operands rounded once more to float IS the float operation: +, -, *, /, sqrt are innocuous under double rounding at these widths). */
static double F32(double x) { return (double)(float)x; }
static void apply_preconditioner_tiled_f32(eo_sim* s, const double* r, double* z) {
  int X = s->X, Y = s->Y;
  size_t C = (size_t)X * (size_t)Y;
  double* pre = s->precon; double* q = s->q;
  uint8_t* start = (uint8_t*)malloc((size_t)X + 64 + 1);
  for (int t = 0; t <= X + 63; ++t) start[t] = (uint8_t)eo_tile_start(s->tile_records, t);
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      const int l = y & 63, t = x + l, cut = start[t];
      double a = s->a_diag[i];
      double cl = -1 * (cut ? 0.0 : F32(pre[i - 1]));      /* (a stale entry of a double solve is read as float) */
      double cb = -1 * ((cut || l == 0) ? 0.0 : F32(pre[i - X]));
      double e = F32(F32(a - F32(cl * cl)) - F32(cb * cb));
      if (e < F32(0.25 * a)) e = a != 0 ? a : 1;
      pre[i] = F32(1 / F32(sqrt(e)));
    }
  memset(q, 0, C * sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      const int l = y & 63, t = x + l, cut = start[t];
      double t_ = F32(F32(r[i] - (cut ? -0.0 : F32(-1 * pre[i - 1] * q[i - 1])))
                           - ((cut || l == 0) ? -0.0 : F32(-1 * pre[i - X] * q[i - X])));
      q[i] = F32(t_ * pre[i]);
    }
  memset(z, 0, C * sizeof(double));
  for (int y = Y; y--;)
    for (int x = X; x--;) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      const int l = y & 63, t = x + l, cut = start[t + 1];
      double t_ = F32(F32(q[i] - F32((FLUID(s, y, x + 1) ? -1 : 0) * pre[i] * (cut ? 0.0 : z[i + 1])))
                              - F32((FLUID(s, y + 1, x) ? -1 : 0) * pre[i] * ((cut || l == 63) ? 0.0 : z[i + X])));
      z[i] = F32(t_ * pre[i]);
    }
  free(start);
}
static void apply_a_f32(const eo_sim* s, const double* in, double* out) {
  int X = s->X, Y = s->Y;
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      double v = F32(s->a_diag[i] * in[i]);
      v = F32(v - (FLUID(s, y, x + 1) ? in[i + 1] : 0));
      v = F32(v - (FLUID(s, y + 1, x) ? in[i + X] : 0));
      v = F32(v - (FLUID(s, y, x - 1) ? in[i - 1] : 0));
      v = F32(v - (FLUID(s, y - 1, x) ? in[i - X] : 0));
      out[i] = v;
    }
}
/* the PCG loop of project() (main.c:742-766) on float vectors; returns the iteration count */
static int pcg_f32(eo_sim* s) {
  size_t C = (size_t)s->X * (size_t)s->Y;
  double *p = s->p, *r = s->r, *z = s->z, *sv = s->s;
  for (size_t i = 0; i < C; ++i) r[i] = F32(r[i]);
  int iters = 0;
  apply_preconditioner_tiled_f32(s, r, z);
  memcpy(sv, z, C * sizeof(double));
  double sigma = eo_dot(s, z, r);
  for (int it = 0; it < s->max_iterations; ++it) {
    apply_a_f32(s, sv, z);
    iters++;
    const double alpha = sigma / eo_dot(s, z, sv);
    const double at = F32(alpha), nat = F32(-alpha);
    for (size_t i = 0; i < C; ++i) if (s->count[i]) { p[i] = F32(p[i] + F32(sv[i] * at)); r[i] = F32(r[i] + F32(z[i] * nat)); }
    s->last_residual = eo_inf_norm(s, r);
    if (s->last_residual <= s->tol) break;
    if (it + 1 == s->max_iterations) break;      /* (the budget's last preconditioner application is never consumed) */
    apply_preconditioner_tiled_f32(s, r, z);
    const double sigma_new = eo_dot(s, z, r);
    const double bt = F32(sigma_new / sigma);
    for (size_t i = 0; i < C; ++i) if (s->count[i]) sv[i] = F32(z[i] + F32(bt * sv[i]));
    sigma = sigma_new;
  }
  return iters;
}

/* EXTENSION (two-level, euler_oracle.h coarse_m): z += P (P^T A P)^-1 P^T r. */
int eo_coarse_m(int X, int Y) {
  int m = 1;
  while (((X + 64 * m - 1) / (64 * m)) * ((Y + 64 * m - 1) / (64 * m)) > 256) m *= 2;
  return m;
}
static void coarse_top_solve(eo_sim* s, double* rc);
static void mg_build(eo_sim* s, double** A_top, int* n_top, int* nx_top);
/* the dense level: A (n x n, symmetric, banded with half-bandwidth bw) -> its Cholesky factor in place, with the pin rule for cut-off fluid and the indicators of such components */
static void dense_factor(eo_sim* s, double* A, int n, int nx, int bw) {
  s->coarse_npinned = 0;
  for (int c = 0; c < n; ++c) if (A[(size_t)c * n + c] == 0.0) A[(size_t)c * n + c] = 1.0;      /* a coarse cell without fluid */
  double* a_kk = (double*)malloc((size_t)n * sizeof(double));
  for (int c = 0; c < n; ++c) a_kk[c] = A[(size_t)c * n + c];
  for (int k = 0; k < n; ++k) {      /* in place, lower triangle (banded: entries beyond column distance bw stay 0) */
    /* fluid cut off from the air makes A and P^T A P singular: the last pivot of such a component is rounding noise - it falls back to the diagonal (that coarse cell is pinned) */
    if (!(A[(size_t)k * n + k] > 1e-8 * a_kk[k])) { A[(size_t)k * n + k] = a_kk[k]; if (s->coarse_npinned < 16 && a_kk[k] != 1.0) s->coarse_pinned[s->coarse_npinned++] = k; }
    const double d = sqrt(A[(size_t)k * n + k]);
    A[(size_t)k * n + k] = d;
    const int hi = k + bw < n - 1 ? k + bw : n - 1;
    for (int i = k + 1; i <= hi; ++i) A[(size_t)i * n + k] /= d;
    for (int j = k + 1; j <= hi; ++j)
      for (int i = j; i <= hi; ++i) A[(size_t)i * n + j] -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
  }
  s->coarse_chol = A; s->coarse_n = n; s->coarse_nx = nx; s->coarse_bw = bw;
  /* the indicator of every cut-off component: the pinned system's answer to its own pin, n = a_kk S e_k (coarse_top_solve projects it out on both sides) */
  free(s->coarse_null);
  s->coarse_null = s->coarse_npinned ? (double*)calloc((size_t)s->coarse_npinned * n, sizeof(double)) : NULL;
  const int np = s->coarse_npinned;
  s->coarse_npinned = 0;      /* (plain pinned solves while the indicators are formed, one after the other as the product does) */
  for (int q = 0; q < np; ++q) {
    double* nv = s->coarse_null + (size_t)q * n;
    nv[s->coarse_pinned[q]] = a_kk[s->coarse_pinned[q]];
    coarse_top_solve(s, nv);
    s->coarse_npinned = q + 1;
  }
  free(a_kk);
}
static void coarse_factor(eo_sim* s) {      /* A_c = P^T A P: sums of A's entries over pairs of coarse cells, then its Cholesky factor */
  if (s->coarse_mg) {      /* multilevel mode: the dense level is the top of the hierarchy (mg_build) */
    double* A; int n, nx;
    mg_build(s, &A, &n, &nx);
    dense_factor(s, A, n, nx, n > nx ? nx + 1 : n - 1 > 0 ? n - 1 : 1);
    return;
  }
  const int X = s->X, Y = s->Y, g = 64 * s->coarse_m;
  const int nx = (X + g - 1) / g, ny = (Y + g - 1) / g, n = nx * ny;
  double* A = (double*)calloc((size_t)n * n, sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      const int c = (y / g) * nx + x / g;
      A[(size_t)c * n + c] += s->a_diag[AT(s, y, x)];
      if (FLUID(s, y, x + 1)) { const int d = (y / g) * nx + (x + 1) / g; A[(size_t)c * n + d] -= 1; A[(size_t)d * n + c] -= 1; }
      if (FLUID(s, y + 1, x)) { const int d = ((y + 1) / g) * nx + x / g; A[(size_t)c * n + d] -= 1; A[(size_t)d * n + c] -= 1; }
    }
  dense_factor(s, A, n, nx, nx);
}
static void coarse_project_null(const eo_sim* s, double* v) {      /* v <- (I - n n^T / n.n) v for the indicator n of every cut-off component */
  const int n = s->coarse_n;
  for (int q = 0; q < s->coarse_npinned; ++q) {
    const double* nv = s->coarse_null + (size_t)q * n;
    double nn = 0.0, nvv = 0.0;
    for (int i = 0; i < n; ++i) { nn += nv[i] * nv[i]; nvv += nv[i] * v[i]; }
    for (int i = 0; i < n; ++i) v[i] -= nv[i] * (nvv / nn);
  }
}
static void coarse_top_solve_pinned(eo_sim* s, double* rc);
static void coarse_top_solve(eo_sim* s, double* rc) {      /* rc <- pseudo-inverse of P^T A P applied to rc */
  coarse_project_null(s, rc);
  coarse_top_solve_pinned(s, rc);
  coarse_project_null(s, rc);
}
static void coarse_top_solve_pinned(eo_sim* s, double* rc) {      /* rc <- S rc, S = the inverse of the (pinned) factored matrix, by the two banded substitutions */
  const int n = s->coarse_n, nx = s->coarse_bw;
  const double* L = s->coarse_chol;
  for (int i = 0; i < n; ++i) {      /* L w = r_c */
    double t = rc[i];
    for (int j = i - nx > 0 ? i - nx : 0; j < i; ++j) t -= L[(size_t)i * n + j] * rc[j];
    rc[i] = t / L[(size_t)i * n + i];
  }
  for (int i = n; i--;) {            /* L^T y = w */
    double t = rc[i];
    const int hi = i + nx < n - 1 ? i + nx : n - 1;
    for (int j = i + 1; j <= hi; ++j) t -= L[(size_t)j * n + i] * rc[j];
    rc[i] = t / L[(size_t)i * n + i];
  }
}
static void coarse_correction(eo_sim* s, const double* r, double* z) {
  const int X = s->X, Y = s->Y, g = 64 * s->coarse_m;
  if (!s->coarse_chol) coarse_factor(s);
  const int nx = s->coarse_nx, n = s->coarse_n;
  double* rc = (double*)calloc((size_t)n, sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x)
      if (FLUID(s, y, x)) rc[(y / g) * nx + x / g] += r[AT(s, y, x)];
  coarse_top_solve(s, rc);
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x)
      if (FLUID(s, y, x)) z[AT(s, y, x)] += rc[(y / g) * nx + x / g];
  free(rc);
}

/* EXTENSION (multilevel, euler_oracle.h coarse_mg; round 5: bilinear coarse spaces): z += P_0 V(P_0^T r).
 * Level 0: a node per G0 x G0 grid cells (G0 = 8), at the centre of cell (G0 J + G0 / 2, G0 I + G0 / 2) (nx0 = ceil(X / G0), ny0 = (64 / G0) ceil(Y / 64): the product's bands);
 * P_0 = BILINEAR interpolation from the four nodes around a cell (weights in 1 / G0), restricted to the fluid, constant beyond the outermost nodes.  Level l + 1: every other node of level l in both
 * directions (its node J sits on node 2 J), bilinear again (weights 1, 1/2: full weighting).  Nodes sit AT cell centres so that a hat is 0 at the neighbouring nodes.  A_0 = P_0^T A P_0, A_(l+1) = P^T A_l P: nine-point stencils, a[k][c] = the entry that couples node c = (I, J) to node (I + k / 3 - 1, J + k % 3 - 1).
 * One symmetric V-cycle: damped Jacobi (omega) from a zero guess, restricted residual, recursion, correction, Jacobi again; the top level (at most MG_TOP_MAX nodes) is
 * solved exactly (dense_factor).  Round 3-4 used piecewise constants over the same blocks (aggregation, 5-point stencils with integer entries, correction scaled by 1.7):
 * 104 / 108 PCG iterations to 1e-6 on the 1024^2 / 2048^2 tank at rest, 109 on a 512^2 dam break at impact - the bilinear spaces need 52 / 52 / 64 with nodes 16 cells
 * apart and 30 / - / 39 with 8 (tools/r05/mg_proto.py). */
typedef struct { int nx, ny; double* a[9]; double *rhs, *x, *t, *x1, *wd; } mg_level;      /* wd: the Jacobi steps' omega_i / d_i (mg_damping) */
typedef struct { int nlev; mg_level lv[20]; int nnull; double* n0[4]; double* m0[4]; } mg_hierarchy;      /* n0 / m0: see mg_gauge */
#define MG_OMEGA 0.8
#define MG_THETA 1.6     /* the Jacobi steps stay below this Gershgorin bound of D~^-1 A (the product: k_mg.h MG_THETA) */
#define MG_TOP_MAX 64
#define MG_G0 8       /* level 0's node spacing in grid cells (the product: k_mg.h MG_G0) */
#define MG_LOG 3
static void mg_free(eo_sim* s) {
  mg_hierarchy* h = (mg_hierarchy*)s->mg;
  if (!h) return;
  for (int l = 0; l < h->nlev; ++l) { free(h->lv[l].a[0]); free(h->lv[l].rhs); }
  for (int q = 0; q < 4; ++q) { free(h->n0[q]); free(h->m0[q]); }
  free(h); s->mg = NULL;
}
static void mg_alloc_level(mg_level* L, int nx, int ny) {
  const size_t n = (size_t)nx * ny;
  L->nx = nx; L->ny = ny;
  L->a[0] = (double*)calloc(9 * n, sizeof(double));
  for (int k = 1; k < 9; ++k) L->a[k] = L->a[0] + k * n;
  L->rhs = (double*)calloc(5 * n, sizeof(double)); L->x = L->rhs + n; L->t = L->rhs + 2 * n; L->x1 = L->rhs + 3 * n; L->wd = L->rhs + 4 * n;
}
/* cell c of a row / column of cells against n nodes, node j AT the centre of cell G0 j + G0 / 2 (so a node's hat is 0 at the neighbouring nodes and the Galerkin operators are
 * exact nine-point stencils): the two nodes j0, j1 around the cell and the weight f of j1 (1 - f of j0), a multiple of 1 / G0; constant beyond the outermost nodes */
static inline void mg_w0(int c, int n, int* j0, int* j1, double* f) {
  const int u = c - MG_G0 / 2;
  int j = u >= 0 ? u >> MG_LOG : -1;
  double w = (double)(u - MG_G0 * j) / (double)MG_G0;
  if (j < 0) { j = 0; w = 0.0; }                  /* left of the first node */
  int k = j + 1;
  if (k > n - 1) { k = n - 1; w = 0.0; }          /* right of the last node */
  *j0 = j; *j1 = k; *f = w;
}
/* node c of a finer level against the n nodes of the next one, whose node j sits ON the finer level's node 2 j: weights 1 or 1/2, 1/2 */
static inline void mg_w1(int c, int n, int* j0, int* j1, double* f) {
  int j = c >> 1;
  double w = (c & 1) ? 0.5 : 0.0;
  int k = j + 1;
  if (k > n - 1) { k = n - 1; w = 0.0; }
  *j0 = j; *j1 = k; *f = w;
}
static inline void mg_add(mg_level* L, int I, int J, int I2, int J2, double v) {      /* A[(I, J), (I2, J2)] += v */
  if (v == 0.0) return;
  L->a[(I2 - I + 1) * 3 + (J2 - J + 1)][(size_t)I * L->nx + J] += v;
}
/* Damped Jacobi, x += (omega_i / d_i) (rhs - A x)_i, converges iff the eigenvalues of D~^-1 A stay below 2.  On a regular node of these nine-point operators the off-diagonal
 * entries add up to the diagonal and omega = 0.8 gives 1.6; but a drop of spray - ONE fluid cell seen by its four nodes - is a rank-one block whose eigenvalue is 4 d, and the
 * step then multiplies that mode by 1 - 3.2 instead of damping it: the cycle stops approximating the coarse solve, and with a few hundred drops on the grid (the waterfall after
 * 100 s) PCG needs as many iterations as without a coarse correction.  So every node damps by omega_i = min(omega, theta / (1 + sum |off-diagonals| / d)): Gershgorin keeps D~^-1 A below theta = 1.6 < 2 everywhere,
 * regular nodes keep 0.8, and the cycle stays symmetric (the same D~ before and behind the correction) and positive. */
static void mg_damping(mg_level* L, double theta) {
  const size_t n = (size_t)L->nx * L->ny;
  for (size_t c = 0; c < n; ++c) {
    const double d = L->a[4][c];
    double off = 0.0;
    for (int k = 0; k < 9; ++k) if (k != 4) off = off + fabs(L->a[k][c]);
    double om = d != 0.0 ? theta / (1.0 + off / d) : 0.0;
    if (om > MG_OMEGA) om = MG_OMEGA;
    L->wd[c] = d != 0.0 ? om / d : 0.0;
  }
}
static void mg_build(eo_sim* s, double** A_top, int* n_top, int* nx_top) {
  const int X = s->X, Y = s->Y;
  mg_free(s);
  mg_hierarchy* h = (mg_hierarchy*)calloc(1, sizeof(mg_hierarchy));
  int nx = (X + MG_G0 - 1) / MG_G0, ny = (64 / MG_G0) * ((Y + 63) / 64), l = 0;
  for (;; ++l) { mg_alloc_level(&h->lv[l], nx, ny); if (nx * ny <= MG_TOP_MAX) break; nx = (nx + 1) / 2; ny = (ny + 1) / 2; }
  h->nlev = l + 1;
  mg_level* L0 = &h->lv[0];
  /* A_0 = P_0^T A P_0: x^T A x = sum_cells a_diag x_i^2 - 2 sum_edges x_i x_j with x = P_0 X; every entry is a multiple of 1 / G0^4 - exact in any order */
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      int jx[2], jy[2]; double wx[2], wy[2], f;
      mg_w0(x, L0->nx, &jx[0], &jx[1], &f); wx[0] = 1.0 - f; wx[1] = f;
      mg_w0(y, L0->ny, &jy[0], &jy[1], &f); wy[0] = 1.0 - f; wy[1] = f;
      const double ad = (double)s->a_diag[AT(s, y, x)];
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
          mg_add(L0, jy[a >> 1], jx[a & 1], jy[b >> 1], jx[b & 1], ad * (wy[a >> 1] * wx[a & 1]) * (wy[b >> 1] * wx[b & 1]));
      for (int e = 0; e < 2; ++e) {      /* the edge to the right (e = 0) / upper (e = 1) neighbour, both directions */
        const int x2 = x + (e == 0), y2 = y + (e == 1);
        if (!FLUID(s, y2, x2)) continue;
        int kx[2], ky[2]; double vx[2], vy[2];
        mg_w0(x2, L0->nx, &kx[0], &kx[1], &f); vx[0] = 1.0 - f; vx[1] = f;
        mg_w0(y2, L0->ny, &ky[0], &ky[1], &f); vy[0] = 1.0 - f; vy[1] = f;
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b) {
            const double v = -((wy[a >> 1] * wx[a & 1]) * (vy[b >> 1] * vx[b & 1]));
            mg_add(L0, jy[a >> 1], jx[a & 1], ky[b >> 1], kx[b & 1], v);
            mg_add(L0, ky[b >> 1], kx[b & 1], jy[a >> 1], jx[a & 1], v);
          }
      }
    }
  for (int k = 1; k < h->nlev; ++k) {      /* A_k = P^T A_(k-1) P */
    const mg_level* F = &h->lv[k - 1]; mg_level* C = &h->lv[k];
    for (int I = 0; I < F->ny; ++I)
      for (int J = 0; J < F->nx; ++J) {
        const size_t c = (size_t)I * F->nx + J;
        int jx[2], jy[2]; double wx[2], wy[2], f;
        mg_w1(J, C->nx, &jx[0], &jx[1], &f); wx[0] = 1.0 - f; wx[1] = f;
        mg_w1(I, C->ny, &jy[0], &jy[1], &f); wy[0] = 1.0 - f; wy[1] = f;
        for (int e = 0; e < 9; ++e) {
          const double av = F->a[e][c];
          if (av == 0.0) continue;
          const int I2 = I + e / 3 - 1, J2 = J + e % 3 - 1;
          int kx[2], ky[2]; double vx[2], vy[2];
          mg_w1(J2, C->nx, &kx[0], &kx[1], &f); vx[0] = 1.0 - f; vx[1] = f;
          mg_w1(I2, C->ny, &ky[0], &ky[1], &f); vy[0] = 1.0 - f; vy[1] = f;
          for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
              mg_add(C, jy[a >> 1], jx[a & 1], ky[b >> 1], kx[b & 1], (wy[a >> 1] * wx[a & 1]) * av * (vy[b >> 1] * vx[b & 1]));
        }
      }
  }
  for (int k = 0; k < h->nlev; ++k) mg_damping(&h->lv[k], s->mg_theta > 0.0 ? s->mg_theta : MG_THETA);
  s->mg = h;
  const mg_level* T = &h->lv[h->nlev - 1];
  const int n = T->nx * T->ny;
  double* A = (double*)calloc((size_t)n * n, sizeof(double));
  for (int I = 0; I < T->ny; ++I)
    for (int J = 0; J < T->nx; ++J)
      for (int e = 0; e < 9; ++e) {
        const int I2 = I + e / 3 - 1, J2 = J + e % 3 - 1;
        if (I2 < 0 || I2 >= T->ny || J2 < 0 || J2 >= T->nx) continue;
        A[(size_t)(I * T->nx + J) * n + I2 * T->nx + J2] = T->a[e][(size_t)I * T->nx + J];
      }
  *A_top = A; *n_top = n; *nx_top = T->nx;
}
/* (A_l v)[c] in the order of the stencil: k = 0 .. 8 */
static double mg_apply(const mg_level* L, const double* v, int I, int J) {
  const size_t c = (size_t)I * L->nx + J;
  double t = 0.0;
  for (int e = 0; e < 9; ++e) {
    const int I2 = I + e / 3 - 1, J2 = J + e % 3 - 1;
    if (I2 < 0 || I2 >= L->ny || J2 < 0 || J2 >= L->nx) continue;
    t = t + L->a[e][c] * v[(size_t)I2 * L->nx + J2];
  }
  return t;
}
static void mg_restrict(const mg_level* F, const double* t, mg_level* C, double* out) {      /* out = P^T t */
  memset(out, 0, (size_t)C->nx * C->ny * sizeof(double));
  for (int I = 0; I < F->ny; ++I)
    for (int J = 0; J < F->nx; ++J) {
      int jx[2], jy[2]; double wx[2], wy[2], f;
      mg_w1(J, C->nx, &jx[0], &jx[1], &f); wx[0] = 1.0 - f; wx[1] = f;
      mg_w1(I, C->ny, &jy[0], &jy[1], &f); wy[0] = 1.0 - f; wy[1] = f;
      const double v = t[(size_t)I * F->nx + J];
      for (int a = 0; a < 4; ++a) out[(size_t)jy[a >> 1] * C->nx + jx[a & 1]] += (wy[a >> 1] * wx[a & 1]) * v;
    }
}
static double mg_interp1(const mg_level* F, const mg_level* C, const double* e, int I, int J) {      /* (P e) at node (I, J) of the finer level */
  int jx[2], jy[2]; double fx, fy;
  mg_w1(J, C->nx, &jx[0], &jx[1], &fx);
  mg_w1(I, C->ny, &jy[0], &jy[1], &fy);
  (void)F;
  const double lo = (1.0 - fx) * e[(size_t)jy[0] * C->nx + jx[0]] + fx * e[(size_t)jy[0] * C->nx + jx[1]];
  const double hi = (1.0 - fx) * e[(size_t)jy[1] * C->nx + jx[0]] + fx * e[(size_t)jy[1] * C->nx + jx[1]];
  return (1.0 - fy) * lo + fy * hi;
}
static double mg_interp0(const mg_level* L0, const double* e, int x, int y) {      /* (P_0 e) at cell (x, y) */
  int jx[2], jy[2]; double fx, fy;
  mg_w0(x, L0->nx, &jx[0], &jx[1], &fx);
  mg_w0(y, L0->ny, &jy[0], &jy[1], &fy);
  const double lo = (1.0 - fy) * e[(size_t)jy[0] * L0->nx + jx[0]] + fy * e[(size_t)jy[1] * L0->nx + jx[0]];
  const double hi = (1.0 - fy) * e[(size_t)jy[0] * L0->nx + jx[1]] + fy * e[(size_t)jy[1] * L0->nx + jx[1]];
  return (1.0 - fx) * lo + fx * hi;
}
static void mg_vcycle(eo_sim* s, mg_hierarchy* h, int l) {      /* lv[l].x = V_l lv[l].rhs */
  mg_level* L = &h->lv[l];
  const size_t n = (size_t)L->nx * L->ny;
  if (l == h->nlev - 1) { memcpy(L->x, L->rhs, n * sizeof(double)); coarse_top_solve(s, L->x); return; }
  mg_level* C = &h->lv[l + 1];
  const double* d = L->a[4];
  for (size_t c = 0; c < n; ++c) L->x1[c] = d[c] != 0.0 ? L->wd[c] * L->rhs[c] : 0.0;      /* Jacobi from a zero guess (the product keeps omega / d per node) */
  for (int I = 0; I < L->ny; ++I)
    for (int J = 0; J < L->nx; ++J) { const size_t c = (size_t)I * L->nx + J; L->t[c] = d[c] != 0.0 ? L->rhs[c] - mg_apply(L, L->x1, I, J) : 0.0; }
  mg_restrict(L, L->t, C, C->rhs);
  mg_vcycle(s, h, l + 1);
  for (int I = 0; I < L->ny; ++I)      /* the correction */
    for (int J = 0; J < L->nx; ++J) { const size_t c = (size_t)I * L->nx + J; L->t[c] = d[c] != 0.0 ? L->x1[c] + mg_interp1(L, C, C->x, I, J) : 0.0; }
  for (int I = 0; I < L->ny; ++I)      /* Jacobi again */
    for (int J = 0; J < L->nx; ++J) { const size_t c = (size_t)I * L->nx + J; L->x[c] = d[c] != 0.0 ? L->t[c] + L->wd[c] * (L->rhs[c] - mg_apply(L, L->t, I, J)) : 0.0; }
}
static double* mg_null_level0(eo_sim* s, const double* nv);
/* Water cut off from the air: its pressure is determined up to a constant, PCG delivers the one with n . M p = 0 (M the preconditioner, n the region's indicator) and the
 * reference's clamp (main.c:773-779) makes that constant observable.  With the tile-local factor alone that is, like the reference's own, very nearly "p has mean 0 over the
 * region"; the dense level's pseudo-inverse keeps it so, the Jacobi steps of the levels in between do not.  So the correction is made mean-free over the CELLS of every such
 * region before it is added: x_0 -= n_0 (m_0 . x_0) / (m_0 . n_0), n_0 = the indicator on level 0, m_0 = P_0^T (the indicator on the cells). */
static void mg_gauge(eo_sim* s, mg_hierarchy* h) {
  const int np = s->coarse_npinned < 4 ? s->coarse_npinned : 4;
  if (np == 0) return;
  mg_level* L0 = &h->lv[0];
  const size_t n = (size_t)L0->nx * L0->ny;
  if (h->nnull != np || !h->n0[0]) {
    for (int q = 0; q < 4; ++q) { free(h->n0[q]); free(h->m0[q]); h->n0[q] = h->m0[q] = NULL; }
    for (int q = 0; q < np; ++q) {
      h->n0[q] = mg_null_level0(s, s->coarse_null + (size_t)q * s->coarse_n);
      h->m0[q] = (double*)calloc(n, sizeof(double));
      for (int y = 0; y < s->Y; ++y)
        for (int x = 0; x < s->X; ++x) {
          if (!FLUID(s, y, x) || !(mg_interp0(L0, h->n0[q], x, y) > 0.5)) continue;
          int jx[2], jy[2]; double fx, fy;
          mg_w0(x, L0->nx, &jx[0], &jx[1], &fx);
          mg_w0(y, L0->ny, &jy[0], &jy[1], &fy);
          h->m0[q][(size_t)jy[0] * L0->nx + jx[0]] += (1.0 - fy) * (1.0 - fx);
          h->m0[q][(size_t)jy[0] * L0->nx + jx[1]] += (1.0 - fy) * fx;
          h->m0[q][(size_t)jy[1] * L0->nx + jx[0]] += fy * (1.0 - fx);
          h->m0[q][(size_t)jy[1] * L0->nx + jx[1]] += fy * fx;
        }
    }
    h->nnull = np;
  }
  for (int q = 0; q < np; ++q) {
    double mx = 0.0, mn = 0.0;
    for (size_t c = 0; c < n; ++c) { mx += h->m0[q][c] * L0->x[c]; mn += h->m0[q][c] * h->n0[q][c]; }
    if (mn > 0.0) for (size_t c = 0; c < n; ++c) L0->x[c] -= h->n0[q][c] * (mx / mn);
  }
}
static void mg_correction(eo_sim* s, const double* r, double* z) {
  const int X = s->X, Y = s->Y;
  if (!s->coarse_chol) coarse_factor(s);
  mg_hierarchy* h = (mg_hierarchy*)s->mg;
  mg_level* L0 = &h->lv[0];
  memset(L0->rhs, 0, (size_t)L0->nx * L0->ny * sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      int jx[2], jy[2]; double fx, fy;
      mg_w0(x, L0->nx, &jx[0], &jx[1], &fx);
      mg_w0(y, L0->ny, &jy[0], &jy[1], &fy);
      const double rv = r[AT(s, y, x)];
      L0->rhs[(size_t)jy[0] * L0->nx + jx[0]] += ((1.0 - fy) * (1.0 - fx)) * rv;
      L0->rhs[(size_t)jy[0] * L0->nx + jx[1]] += ((1.0 - fy) * fx) * rv;
      L0->rhs[(size_t)jy[1] * L0->nx + jx[0]] += (fy * (1.0 - fx)) * rv;
      L0->rhs[(size_t)jy[1] * L0->nx + jx[1]] += (fy * fx) * rv;
    }
  mg_vcycle(s, h, 0);
  mg_gauge(s, h);
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x)
      if (FLUID(s, y, x)) z[AT(s, y, x)] += mg_interp0(L0, L0->x, x, y);
}
/* the indicator of a cut-off component, known on the top level (coarse_null), at the cells: prolonged down the hierarchy, then P_0 */
static double* mg_null_level0(eo_sim* s, const double* nv) {
  mg_hierarchy* h = (mg_hierarchy*)s->mg;
  const int top = h->nlev - 1;
  double* cur = (double*)malloc((size_t)h->lv[top].nx * h->lv[top].ny * sizeof(double));
  memcpy(cur, nv, (size_t)h->lv[top].nx * h->lv[top].ny * sizeof(double));
  for (int l = top - 1; l >= 0; --l) {
    const mg_level* L = &h->lv[l]; const mg_level* C = &h->lv[l + 1];
    double* nxt = (double*)malloc((size_t)L->nx * L->ny * sizeof(double));
    for (int I = 0; I < L->ny; ++I)
      for (int J = 0; J < L->nx; ++J) nxt[(size_t)I * L->nx + J] = L->a[4][(size_t)I * L->nx + J] != 0.0 ? mg_interp1(L, C, cur, I, J) : 0.0;
    free(cur); cur = nxt;
  }
  return cur;
}

void eo_apply_preconditioner(eo_sim* s, const double* r, double* z) {
  if (s->tile_records > 0) {
    apply_preconditioner_tiled(s, r, z);
    if (s->coarse_m > 0 && s->coarse_mg) mg_correction(s, r, z);
    else if (s->coarse_m > 0) coarse_correction(s, r, z);
    return;
  }
  int X = s->X, Y = s->Y;
  size_t C = (size_t)X * (size_t)Y;
  double* pre = s->precon; double* q = s->q;
  /* E^-1 of the IC(0) factor, recomputed every call as the reference does (main.c:586-600).
   * NOTE the reference's "minus" coefficients: get_a_minus_i(y,x) = get_a_plus_i(y,x-1) =
   * is_fluid(y,x) ? -1 : 0 (main.c:561-575), i.e. ALWAYS -1 for the fluid cell being visited,
   * whatever the left/lower neighbour is.  So the stale precon[] of a neighbour that is no
   * longer fluid DOES enter e (precon[] persists across calls and substeps, main.c:577), and
   * in the forward solve the neighbour term is (-1*precon)*q with q=+0 on non-fluid cells. */
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      double a = s->a_diag[i];
      double cl = -1 * pre[i - 1];
      double cb = -1 * pre[i - X];
      double e = a - cl * cl - cb * cb;
      if (e < 0.25 * a) e = a != 0 ? a : 1;
      pre[i] = 1 / sqrt(e);
    }
  /* L q = r (main.c:602-613) */
  memset(q, 0, C * sizeof(double));
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      double t = r[i] - -1 * pre[i - 1] * q[i - 1]
                      - -1 * pre[i - X] * q[i - X];
      q[i] = t * pre[i];
    }
  /* L^T z = q (main.c:615-626) */
  memset(z, 0, C * sizeof(double));
  for (int y = Y; y--;)
    for (int x = X; x--;) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      double t = q[i] - (FLUID(s, y, x + 1) ? -1 : 0) * pre[i] * z[i + 1]
                      - (FLUID(s, y + 1, x) ? -1 : 0) * pre[i] * z[i + X];
      z[i] = t * pre[i];
    }
}

double eo_dot(const eo_sim* s, const double* a, const double* b) {
  double total = 0.f;
  size_t C = (size_t)s->X * (size_t)s->Y;
  for (size_t i = 0; i < C; ++i) if (s->count[i]) total += a[i] * b[i];
  return total;
}

double eo_inf_norm(const eo_sim* s, const double* r) {
  double m = 0.f;
  size_t C = (size_t)s->X * (size_t)s->Y;
  for (size_t i = 0; i < C; ++i) if (s->count[i]) { double a = fabs(r[i]); if (a > m) m = a; }
  return m;
}

void eo_apply_a(const eo_sim* s, const double* in, double* out) {
  int X = s->X, Y = s->Y;
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X; ++x) {
      if (!FLUID(s, y, x)) continue;
      size_t i = AT(s, y, x);
      out[i] = s->a_diag[i] * in[i]
             - (FLUID(s, y, x + 1) ? in[i + 1] : 0)
             - (FLUID(s, y + 1, x) ? in[i + X] : 0)
             - (FLUID(s, y, x - 1) ? in[i - 1] : 0)
             - (FLUID(s, y - 1, x) ? in[i - X] : 0);
    }
}

static void axpy_fluid(const eo_sim* s, const double* a, double k, double* c) { /* fmadd, main.c:694 */
  size_t C = (size_t)s->X * (size_t)s->Y;
  for (size_t i = 0; i < C; ++i) if (s->count[i]) c[i] += a[i] * k;
}

int eo_project(eo_sim* s, float dt, const float* u, const float* v, float* uout, float* vout) {
  int X = s->X, Y = s->Y;
  size_t C = (size_t)X * (size_t)Y;
  eo_build_system(s, dt, u, v);
  double *p = s->p, *r = s->r, *z = s->z, *sv = s->s;
  memset(p, 0, C * sizeof(double));
  memcpy(r, s->b, C * sizeof(double));
  int iters = 0;
  int nonzero = 0;
  for (size_t i = 0; i < C && !nonzero; ++i) if (s->count[i] && r[i] != 0.f) nonzero = 1;
  if (nonzero && s->tile_records > 0 && s->coarse_m > 0) {
    /* EXTENSION (coarse modes): water cut off from the air makes A singular along the indicator n of that region, and b - float divergences - is compatible
     * with that only to rounding (n.b ~ 1e-4, not 0).  CG on a singular, slightly inconsistent system wanders once it gets close; the part of b along n
     * (a few 1e-9 per cell, far below the tolerance, and no A s can touch it anyway) is taken out before the solve starts. */
    if (!s->coarse_chol) coarse_factor(s);
    const int g = 64 * s->coarse_m;
    /* (all the sums from the r the solve starts with, then all the corrections - the indicators of distinct regions do not overlap - as the product's two launches do;
     * the product keeps at most four indicators) */
    double eps[4] = {0, 0, 0, 0}, nn[4] = {0, 0, 0, 0};
    const int np = s->coarse_npinned < 4 ? s->coarse_npinned : 4;
    double* n0[4] = {NULL, NULL, NULL, NULL};      /* multilevel mode: the indicators on level 0 (the cells sample them through P_0) */
    const mg_level* L0 = s->coarse_mg ? &((mg_hierarchy*)s->mg)->lv[0] : NULL;
    for (int q = 0; q < np && s->coarse_mg; ++q) n0[q] = mg_null_level0(s, s->coarse_null + (size_t)q * s->coarse_n);
#define NULL_AT(q, y, x) (s->coarse_mg ? mg_interp0(L0, n0[q], x, y) : (s->coarse_null + (size_t)(q) * s->coarse_n)[((y) / g) * s->coarse_nx + (x) / g])
    for (int q = 0; q < np; ++q) {
      for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x)
          if (FLUID(s, y, x)) { const double w = NULL_AT(q, y, x); eps[q] += r[AT(s, y, x)] * w; nn[q] += w * w; }
    }
    for (int q = 0; q < np; ++q) {
      /* (> 1e-12, not > 0: a pinned node can also be a dependent one - two nodes that see the same lone drop - whose "indicator" is rounding noise on the cells, 1e-17 on one
       * cell; normalised, that would take the WHOLE right-hand side of the cell out.  A real region has cells with indicator ~1) */
      if (nn[q] > 1e-12)
        for (int y = 0; y < Y; ++y)
          for (int x = 0; x < X; ++x)
            if (FLUID(s, y, x)) r[AT(s, y, x)] -= NULL_AT(q, y, x) * (eps[q] / nn[q]);
    }
#undef NULL_AT
    for (int q = 0; q < 4; ++q) free(n0[q]);
  }
  s->last_residual = 0;
  if (nonzero && s->pcg_f32 && s->tile_records > 0 && s->coarse_m == 0) iters = pcg_f32(s);
  else if (nonzero) {
    eo_apply_preconditioner(s, r, z);
    memcpy(sv, z, C * sizeof(double));
    double sigma = eo_dot(s, z, r);
    for (int it = 0; it < s->max_iterations; ++it) {
      eo_apply_a(s, sv, z);
      iters++;
      double alpha = sigma / eo_dot(s, z, sv);
      axpy_fluid(s, sv, alpha, p);
      axpy_fluid(s, z, -alpha, r);
      s->last_residual = eo_inf_norm(s, r);
      if (s->last_residual <= s->tol) break;
      eo_apply_preconditioner(s, r, z);
      double sigma_new = eo_dot(s, z, r);
      double beta = sigma_new / sigma;
      for (size_t i = 0; i < C; ++i) if (s->count[i]) sv[i] = z[i] + beta * sv[i];
      sigma = sigma_new;
    }
  }
  for (size_t i = 0; i < C; ++i) if (s->count[i] && p[i] < 0.f) p[i] = 0.f; /* main.c:773-779 */

  const float neg_inv = -(1.f / (DENSITY * H_CELL)); /* accel(), main.c:705-707 */
  for (int y = 0; y < Y; ++y)
    for (int x = 0; x < X - 1; ++x) {
      size_t i = AT(s, y, x);
      if (prop(s, s->solid, x, y, EO_U)) uout[i] = 0.f;
      else if (prop(s, s->count, x, y, EO_U)) uout[i] = u[i] + (neg_inv * (float)(p[i + 1] - p[i])) * dt;
      else uout[i] = 0.f;
    }
  for (int y = 0; y < Y - 1; ++y)
    for (int x = 0; x < X; ++x) {
      size_t i = AT(s, y, x);
      if (prop(s, s->solid, x, y, EO_V)) vout[i] = 0.f;
      else if (prop(s, s->count, x, y, EO_V)) vout[i] = v[i] + (neg_inv * (float)(p[i + X] - p[i])) * dt;
      else vout[i] = 0.f;
    }
  s->last_pcg_iterations = iters;
  s->total_pcg_iterations += (uint64_t)iters;
  return iters;
}

/* ---------------------------------------------------------------- EXTENSION: velocity diffusion
 * Not in the reference (it is inviscid); named by the north star only.  Defined by this
 * build, restated here so that the GPU stage has something to be checked against: explicit Jacobi step
 * from the old values, float arithmetic, neighbours visited left, right, down, up. */
void eo_diffuse(const eo_sim* s, const float* q, int type, float dt, float* out) {
  const int ex = ext_x(s, type), ey = ext_y(s, type);
  const float c = s->viscosity * dt / (H_CELL * H_CELL);
  for (int y = 0; y < ey; ++y)
    for (int x = 0; x < ex; ++x) {
      const size_t i = AT(s, y, x);
      float r = q[i];
      if (prop(s, s->count, x, y, type) && !prop(s, s->solid, x, y, type)) {
        float acc = 0.f;
        const int nx[4] = {x - 1, x + 1, x, x}, ny[4] = {y, y, y - 1, y + 1};
        for (int k = 0; k < 4; ++k) {
          if (nx[k] < 0 || nx[k] >= ex || ny[k] < 0 || ny[k] >= ey) continue;
          if (!prop(s, s->count, nx[k], ny[k], type) || prop(s, s->solid, nx[k], ny[k], type)) continue;
          acc += q[AT(s, ny[k], nx[k])] - q[i];
        }
        r = q[i] + c * acc;
      }
      out[i] = r;
    }
}

/* ---------------------------------------------------------------- step driver (main.c:843-900) */

int eo_substep(eo_sim* s, float dt) {
  eo_advect_markers(s, dt);
  eo_refresh_marker_counts(s);
  if (s->rainbow) {   /* main.c:859-863 */
    eo_extrapolate(s, s->cr, EO_P);
    eo_extrapolate(s, s->cg, EO_P);
    eo_extrapolate(s, s->cb, EO_P);
  }
  eo_update_fluid_sources(s);
  eo_extrapolate(s, s->u, EO_U);
  eo_extrapolate(s, s->v, EO_V);
  eo_zero_bounds(s, s->u, EO_U);
  eo_zero_bounds(s, s->v, EO_V);
  eo_advect_u(s, s->u, s->v, dt, s->utmp);
  eo_advect_v(s, s->u, s->v, dt, s->vtmp);
  if (s->rainbow) {   /* main.c:873-882: the memcpy moves the WHOLE tmp array, stale non-fluid entries included */
    const size_t bytes = (size_t)s->X * s->Y * sizeof(float);
    eo_advect_p(s, s->cr, s->u, s->v, dt, s->crtmp); memcpy(s->cr, s->crtmp, bytes);
    eo_advect_p(s, s->cg, s->u, s->v, dt, s->cgtmp); memcpy(s->cg, s->cgtmp, bytes);
    eo_advect_p(s, s->cb, s->u, s->v, dt, s->cbtmp); memcpy(s->cb, s->cbtmp, bytes);
  }
  eo_apply_body_forces(s, s->vtmp, dt);
  eo_zero_bounds(s, s->utmp, EO_U);
  eo_zero_bounds(s, s->vtmp, EO_V);
  if (s->viscosity > 0.f) {   /* extension; u, v are dead between advection and projection: scratch */
    const size_t C = (size_t)s->X * s->Y;
    eo_diffuse(s, s->utmp, EO_U, dt, s->u);
    eo_diffuse(s, s->vtmp, EO_V, dt, s->v);
    for (int y = 0; y < s->Y; ++y)
      for (int x = 0; x < s->X; ++x) {
        if (x < s->X - 1) s->utmp[AT(s, y, x)] = s->u[AT(s, y, x)];
        if (y < s->Y - 1) s->vtmp[AT(s, y, x)] = s->v[AT(s, y, x)];
      }
    (void)C;
  }
  int it = eo_project(s, dt, s->utmp, s->vtmp, s->u, s->v);
  s->total_substeps++;
  s->last_dt = dt;
  return it;
}

void eo_step(eo_sim* s) {
  float frame_time = 0.1f;
  int iters = 0, n = 0;
  for (int step = 0; frame_time > 0.f && step < 8; ++step) {
    float dt = eo_calculate_timestep(s, frame_time);
    frame_time -= dt;
    iters += eo_substep(s, dt);
    n++;
  }
  s->last_substeps = n;
  s->last_pcg_iterations = iters;
  s->frame_count++;
}

/* ---------------------------------------------------------------- render (main.c:914-951) */

int eo_render_rows(const eo_sim* s, int wx, int wy, char* out, int cap) {
  static const char* sym[4] = {" ", "o", "O", "0"};
  const char* BLUE = "\x1B[34m"; const char* RESET = "\x1B[0m"; const char* CLR = "\x1b[K";
  int n = 0;
#define PUT(str) do { const char* _p = (str); while (*_p) { if (n < cap) out[n] = *_p; n++; _p++; } } while (0)
  int X = s->X, Y = s->Y;
  int y_cutoff = (Y - 1 - wy) > 1 ? (Y - 1 - wy) : 1;
  for (int y = Y - 1; y-- > y_cutoff;) {
    int prev_water = 0;
    for (int x = 1; x < X - 1 && x < wx + 1; x++) {
      size_t i = AT(s, y, x);
      if (s->solid[i]) {
        if (prev_water) PUT(RESET);
        PUT("X");
        prev_water = 0;
      } else if (s->sink[i]) {
        if (prev_water) PUT(RESET);
        PUT("="); /* the reference leaves prev_water untouched here (main.c:927-931) */
      } else {
        int k = s->count[i] < 3 ? s->count[i] : 3;
        int has_water = k > 0;
        if (!prev_water && has_water && !s->rainbow) PUT(BLUE);
        else if (has_water && s->rainbow) {   /* buffer_append_color, main.c:902-912; misc/color.h:6-14 */
          char tmp[64];
          const float end = nextafterf(256.f, 0.f);
          const float c3[3] = {s->cr[i], s->cg[i], s->cb[i]};
          int o[3];
          for (int k3 = 0; k3 < 3; ++k3) o[k3] = (int)clampf_(0.f, end * powf(c3[k3], 1 / 2.2f), end);
          snprintf(tmp, sizeof tmp, "\x1B[38;2;%d;%d;%dm", o[0], o[1], o[2]);
          PUT(tmp);
        } else if (prev_water && !has_water) PUT(RESET);
        PUT(sym[k]);
        prev_water = has_water;
      }
    }
    PUT(RESET); PUT(CLR);
    if (y > y_cutoff) PUT("\r\n");
  }
#undef PUT
  return n;
}

uint64_t eo_fnv1a64(const void* data, size_t n) {
  const unsigned char* p = (const unsigned char*)data;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
