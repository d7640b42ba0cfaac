/*
 * euler_host.c — host-only C parts of libeuler_hip.so: scenario text -> cell grids, the
 * xorshift64* stream, initial marker seeding, and the ASCII frame formatter.
 *
 * These are the pieces of the reference's sim_init (main.c:209-274) and draw_rows
 * (main.c:914-951) that never touch the hot path; they run once (init) or over a terminal-sized
 * window (render), so they stay on the host.  Everything here is exported through include/euler.h
 * and usable without a GPU.
 */
#include "euler_host.h"

#include <pthread.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- xorshift64* high half (misc/rng.c:5-20) and randf (main.c:203-207) ------------------- */

uint32_t euler_rng_next_u32(uint64_t* state) {
  uint64_t x = *state;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  *state = x;
  return (uint32_t)((x * 0x2545F4914F6CDD1Dull) >> 32);
}

float euler_rng_next_float(uint64_t* state) {
  /* closed [0,1]: divide by UINT32_MAX in double, then narrow (main.c:206) */
  return (float)(euler_rng_next_u32(state) / (double)UINT32_MAX);
}

/* Jump-ahead: the state update is linear over GF(2), so k steps are the 64 x 64 bit matrix M^k; col[i][b] = M^(2^i) e_b.  A row-slab
 * handle uses it to skip the cells of other ranks' rows while it walks the ONE seeding stream (below). */
#define RNG_JUMPS 48
static uint64_t g_jump[RNG_JUMPS][64];
static pthread_once_t g_jump_once = PTHREAD_ONCE_INIT;   /* handles may load scenarios from several threads at once */
static uint64_t jump_apply(const uint64_t* col, uint64_t x) {
  uint64_t y = 0;
  while (x) { y ^= col[__builtin_ctzll(x)]; x &= x - 1; }
  return y;
}
static void jump_fill(void) {
  for (int b = 0; b < 64; ++b) { uint64_t x = 1ull << b; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; g_jump[0][b] = x; }
  for (int i = 1; i < RNG_JUMPS; ++i)
    for (int b = 0; b < 64; ++b) g_jump[i][b] = jump_apply(g_jump[i - 1], g_jump[i - 1][b]);
}
static void jump_init(void) { pthread_once(&g_jump_once, jump_fill); }
uint64_t euler_rng_jump(uint64_t state, uint64_t steps) {
  jump_init();
  for (int i = 0; steps && i < RNG_JUMPS; ++i, steps >>= 1)
    if (steps & 1) state = jump_apply(g_jump[i], state);
  return state;
}

/* ---- scenario text ---------------------------------------------------------------------- */

static void classify(char c, size_t i, uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid) {
  switch (c) { /* main.c:226-235; anything else is empty */
    case 'X': solid[i] = 1; break;
    case '0': fluid[i] = 1; break;
    case '?': fluid[i] = 1; source[i] = 1; break;
    case '=': sink[i] = 1; break;
    default: break;
  }
}

int euler_parse_scenario(const char* text, int32_t len, int32_t X, int32_t Y, int32_t upscale,
                         uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid) {
  if (!text || len < 0 || X < 4 || Y < 4 || !solid || !source || !sink || !fluid) return EULER_EINVAL;
  const size_t C = (size_t)X * (size_t)Y;
  memset(solid, 0, C); memset(source, 0, C); memset(sink, 0, C); memset(fluid, 0, C);

  if (!upscale) {
    /* The reference consumes the text as a stream (main.c:218-241): the first line fills row
     * Y-2, each line fills x = 1..X-2; a line that reaches the width limit has its remainder
     * (including its newline) discarded. */
    int32_t pos = 0;
    for (int32_t y = Y - 2; y > 0 && pos < len; --y) {
      int32_t x = 1;
      while (x < X - 1 && pos < len) {
        char c = text[pos++];
        if (c == '\n') break;
        classify(c, (size_t)y * X + x, solid, source, sink, fluid);
        ++x;
      }
      if (x == X - 1)
        while (pos < len && text[pos++] != '\n') {}
    }
  } else {
    /* Build extension: nearest-neighbour resample of a Wf x Hf picture onto the interior.
     * ch(x,y) = line[(Y-2-y)*Hf/(Y-2)][(x-1)*Wf/(X-2)], missing characters are blanks. */
    int32_t Hf = 0, Wf = 0;
    for (int32_t i = 0, w = 0; i <= len; ++i) {
      if (i == len || text[i] == '\n') {
        if (i < len || w > 0) { ++Hf; if (w > Wf) Wf = w; }
        w = 0;
      } else ++w;
    }
    if (Hf == 0 || Wf == 0) return EULER_EINVAL;
    int32_t* start = (int32_t*)malloc(sizeof(int32_t) * (size_t)Hf * 2);
    if (!start) return EULER_ENOMEM;
    int32_t* length = start + Hf;
    int32_t k = 0, s0 = 0;
    for (int32_t i = 0; i <= len; ++i)
      if (i == len || text[i] == '\n') {
        if (i < len || i > s0) { start[k] = s0; length[k] = i - s0; ++k; }
        s0 = i + 1;
      }
    for (int32_t y = 1; y <= Y - 2; ++y) {
      int64_t fr = ((int64_t)(Y - 2 - y) * Hf) / (Y - 2);
      for (int32_t x = 1; x <= X - 2; ++x) {
        int64_t fc = ((int64_t)(x - 1) * Wf) / (X - 2);
        char c = fc < length[fr] ? text[start[fr] + fc] : ' ';
        classify(c, (size_t)y * X + x, solid, source, sink, fluid);
      }
    }
    free(start);
  }
  /* sink ring around the domain (main.c:244-252) */
  for (int32_t y = 0; y < Y; ++y) { sink[(size_t)y * X] = 1; sink[(size_t)y * X + X - 1] = 1; }
  for (int32_t x = 0; x < X; ++x) { sink[x] = 1; sink[(size_t)(Y - 1) * X + x] = 1; }
  return EULER_OK;
}

/* `tanks` closed tanks on top of each other (1 = SURVEY 8d config 3): tank k takes the rows [k H, (k+1) H), H = Y / tanks,
 * a solid ring one cell inside its border, fluid in its lower half; the rows where two tanks meet are solid as well, so that
 * every tank is the single tank of an X x H grid (the weak-scaling workload: one tank per row slab). */
int euler_half_tanks_grids(int32_t X, int32_t Y, int32_t tanks, uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid) {
  if (tanks < 1 || Y % tanks) return EULER_EINVAL;
  const int32_t H = Y / tanks;
  if (X < 6 || H < 6) return EULER_EINVAL;
  const size_t C = (size_t)X * (size_t)Y;
  memset(solid, 0, C); memset(source, 0, C); memset(sink, 0, C); memset(fluid, 0, C);
  for (int32_t y = 1; y <= Y - 2; ++y) {
    const int32_t ly = y % H;
    for (int32_t x = 1; x <= X - 2; ++x) {
      size_t i = (size_t)y * X + x;
      if (ly == 0 || ly == H - 1 || ly == 1 || ly == H - 2 || x == 1 || x == X - 2) solid[i] = 1;
      else if (ly < H / 2) fluid[i] = 1;
    }
  }
  for (int32_t y = 0; y < Y; ++y) { sink[(size_t)y * X] = 1; sink[(size_t)y * X + X - 1] = 1; }
  for (int32_t x = 0; x < X; ++x) { sink[x] = 1; sink[(size_t)(Y - 1) * X + x] = 1; }
  return EULER_OK;
}
int euler_half_tank_grids(int32_t X, int32_t Y, uint8_t* solid, uint8_t* source, uint8_t* sink, uint8_t* fluid) {
  return euler_half_tanks_grids(X, Y, 1, solid, source, sink, fluid);
}

/* Four jittered markers per fluid cell (main.c:255-266): columns outer, rows inner, quadrant
 * k = 0..3, the x jitter drawn before the y jitter, all arithmetic in float. */
int euler_seed_markers(const uint8_t* fluid, int32_t X, int32_t Y, uint64_t* rng_state,
                       float* markers_xy, uint64_t* n_markers) {
  if (!fluid || !rng_state || !markers_xy || !n_markers) return EULER_EINVAL;
  uint64_t n = 0;
  for (int32_t cx = 0; cx < X; ++cx)
    for (int32_t cy = 0; cy < Y; ++cy) {
      if (!fluid[(size_t)cy * X + cx]) continue;
      for (int k = 0; k < 4; ++k) {
        float jx = euler_rng_next_float(rng_state) / 2;
        float mx = cx + (k < 2 ? 0 : 0.5f) + jx;
        float jy = euler_rng_next_float(rng_state) / 2;
        float my = cy + (k % 2 ? 0 : 0.5f) + jy;
        markers_xy[2 * n] = 1.f * mx;     /* k_side_length = 1 (main.c:58,262) */
        markers_xy[2 * n + 1] = 1.f * my;
        ++n;
      }
    }
  *n_markers = n;
  return EULER_OK;
}

/* The same stream, keeping only the markers whose row floor(y) lies in [row_lo, row_hi) together with their index in the whole
 * array (a row-slab handle: every rank walks the one sequential RNG stream, none stores the whole array).  Two passes by the
 * caller: markers_xy == NULL counts (n_total, n_kept), the second call fills (the RNG state is only advanced by the caller's
 * copy).  cap = room in markers_xy / keys. */
int euler_seed_markers_rows(const uint8_t* fluid, int32_t X, int32_t Y, int32_t row_lo, int32_t row_hi, uint64_t* rng_state,
                            float* markers_xy, uint32_t* keys, uint64_t cap, uint64_t* n_total, uint64_t* n_kept) {
  if (!fluid || !rng_state || !n_total || !n_kept) return EULER_EINVAL;
  /* A cell's markers land in its own row or (a jitter of exactly 0.5) the next one: only the cells of rows [lo2, row_hi) can give
   * this rank a marker.  The cells below and above them in a column are skipped - 4 keys and 8 draws each - with a jump. */
  const int32_t lo2 = row_lo > 0 ? row_lo - 1 : 0, hi2 = row_hi < Y ? row_hi : Y;
  uint32_t* below = (uint32_t*)calloc((size_t)X, sizeof(uint32_t));
  uint32_t* above = (uint32_t*)calloc((size_t)X, sizeof(uint32_t));
  if (!below || !above) { free(below); free(above); return EULER_ENOMEM; }
  for (int32_t cy = 0; cy < lo2; ++cy)
    for (int32_t cx = 0; cx < X; ++cx) below[cx] += fluid[(size_t)cy * X + cx] != 0;
  for (int32_t cy = hi2; cy < Y; ++cy)
    for (int32_t cx = 0; cx < X; ++cx) above[cx] += fluid[(size_t)cy * X + cx] != 0;
  uint64_t n = 0, kept = 0;
  int rc = EULER_OK;
  for (int32_t cx = 0; cx < X && rc == EULER_OK; ++cx) {
    n += 4ull * below[cx];
    *rng_state = euler_rng_jump(*rng_state, 8ull * below[cx]);
    for (int32_t cy = lo2; cy < hi2; ++cy) {
      if (!fluid[(size_t)cy * X + cx]) continue;
      for (int k = 0; k < 4; ++k) {
        float jx = euler_rng_next_float(rng_state) / 2;
        float mx = cx + (k < 2 ? 0 : 0.5f) + jx;
        float jy = euler_rng_next_float(rng_state) / 2;
        float my = cy + (k % 2 ? 0 : 0.5f) + jy;
        const int32_t row = (int32_t)floorf(1.f * my);
        if (row >= row_lo && row < row_hi) {
          if (markers_xy) {
            if (kept >= cap) { rc = EULER_ENOMEM; break; }
            markers_xy[2 * kept] = 1.f * mx; markers_xy[2 * kept + 1] = 1.f * my; keys[kept] = (uint32_t)n;
          }
          ++kept;
        }
        ++n;
      }
      if (rc != EULER_OK) break;
    }
    n += 4ull * above[cx];
    *rng_state = euler_rng_jump(*rng_state, 8ull * above[cx]);
  }
  free(below); free(above);
  if (rc != EULER_OK) return rc;
  *n_total = n; *n_kept = kept;
  return EULER_OK;
}

/* ---- frame formatter (draw_rows, main.c:914-951; escape codes misc/terminal.h:36,53,60) ---- */

static int render_rows(const uint8_t* solid, const uint8_t* sink, const uint8_t* count,
                       const float* cr, const float* cg, const float* cb,
                       int32_t X, int32_t Y, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len) {
  if (!solid || !sink || !count || !len) return EULER_EINVAL;
  static const char glyph[4] = {' ', 'o', 'O', '0'};
  static const char blue[] = "\x1B[34m", reset[] = "\x1B[0m", clear_line[] = "\x1b[K", crlf[] = "\r\n";
  const int dye = cr && cg && cb;
  int64_t n = 0;
#define EMIT(s, k) do { for (int _i = 0; _i < (int)(k); ++_i) { if (out && n < cap) out[n] = (s)[_i]; ++n; } } while (0)
  int32_t cutoff = Y - 1 - wy;
  if (cutoff < 1) cutoff = 1;
  /* the reference's `for (y = Y-1; y-- > cutoff;)` visits y = Y-2 ... cutoff inclusive */
  for (int32_t y = Y - 2; y >= cutoff; --y) {
    int water_run = 0;
    for (int32_t x = 1; x < X - 1 && x < wx + 1; ++x) {
      size_t i = (size_t)y * X + x;
      if (solid[i]) {
        if (water_run) EMIT(reset, 4);
        EMIT("X", 1);
        water_run = 0;
      } else if (sink[i]) {
        if (water_run) EMIT(reset, 4);
        EMIT("=", 1);           /* the run flag is deliberately left as is (main.c:927-931) */
      } else {
        int k = count[i] < 3 ? count[i] : 3;
        if (!water_run && k && !dye) EMIT(blue, 5);
        else if (k && dye) {    /* buffer_append_color (main.c:902-912): sRGB ~ x^(1/2.2), byte = (int)clamp(0, end*x, end) */
          char esc[32];
          const float end = nextafterf(256.f, 0.f);
          const float lin[3] = {cr[i], cg[i], cb[i]};
          int byte[3];
          for (int c = 0; c < 3; ++c) {
            float v = end * powf(lin[c], 1 / 2.2f);
            byte[c] = (int)(v < 0.f ? 0.f : (v > end ? end : v));
          }
          int m = snprintf(esc, sizeof esc, "\x1B[38;2;%d;%d;%dm", byte[0], byte[1], byte[2]);
          EMIT(esc, m);
        } else if (water_run && !k) EMIT(reset, 4);
        EMIT(&glyph[k], 1);
        water_run = k != 0;
      }
    }
    EMIT(reset, 4);
    EMIT(clear_line, 3);
    if (y > cutoff) EMIT(crlf, 2);
  }
#undef EMIT
  *len = (int32_t)n;
  return EULER_OK;
}

int euler_render_grids(const uint8_t* solid, const uint8_t* sink, const uint8_t* count,
                       int32_t X, int32_t Y, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len) {
  return render_rows(solid, sink, count, NULL, NULL, NULL, X, Y, wx, wy, out, cap, len);
}

int euler_render_grids_rgb(const uint8_t* solid, const uint8_t* sink, const uint8_t* count,
                           const float* r, const float* g, const float* b,
                           int32_t X, int32_t Y, int32_t wx, int32_t wy, char* out, int32_t cap, int32_t* len) {
  if (!r || !g || !b) return EULER_EINVAL;
  return render_rows(solid, sink, count, r, g, b, X, Y, wx, wy, out, cap, len);
}
