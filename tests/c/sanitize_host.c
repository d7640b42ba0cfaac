/*
 * sanitize_host.c — the product's host C code (euler_amd/csrc/euler_host.c: scenario parser, upscaler, marker
 * seeding, frame formatter) and the test oracle under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU
 * (GPU sanitizers are not available on this pool).  The analogue of the reference's SHERLOCK build
 * (CMakeLists.txt:4,14-16).  Awkward inputs on purpose; every host result is cross-checked against the oracle's.
 * Built and run by tests/test_sanitizers.py.  Exit code 0 = clean.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "euler.h"
#include "euler_oracle.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); return 1; } } while (0)

static int one_case(const char* text, int len, int X, int Y, int upscale) {
  size_t C = (size_t)X * Y;
  uint8_t* g = calloc(4, C);
  CHECK(g);
  const int rc_host = euler_parse_scenario(text, len, X, Y, upscale, g, g + C, g + 2 * C, g + 3 * C);
  eo_sim* o = eo_create(X, Y);
  CHECK(o);
  const int rc_oracle = eo_load_scenario_mem(o, text, len, upscale);
  CHECK((rc_host == EULER_OK) == (rc_oracle == 0));     /* an empty picture cannot be resampled: both refuse it */
  if (rc_host != EULER_OK) { CHECK(rc_host == EULER_EINVAL && upscale); free(g); eo_destroy(o); return 0; }
  CHECK(memcmp(g, o->solid, C) == 0 && memcmp(g + C, o->source, C) == 0 && memcmp(g + 2 * C, o->sink, C) == 0);
  /* marker seeding: same stream, same order */
  float* mk = malloc(sizeof(float) * 2 * 4 * C);
  uint64_t rng = 0x9bd185c449534b91ull, n = 0;
  CHECK(mk && euler_seed_markers(g + 3 * C, X, Y, &rng, mk, &n) == EULER_OK);
  CHECK(n == o->n_markers && rng == o->rng_state);
  CHECK(n == 0 || memcmp(mk, o->markers, n * 8) == 0);
  /* a few frames of the oracle, then both formatters on its state, plain and coloured, several windows */
  o->rainbow = 1;
  eo_colorize(o);
  for (int f = 0; f < 3; ++f) eo_step(o);
  const int win[4][2] = {{X - 2, Y - 2}, {7, 3}, {1, 1}, {4 * X, 4 * Y}};
  for (int w = 0; w < 4; ++w) {
    for (int dye = 0; dye < 2; ++dye) {
      o->rainbow = dye;
      int32_t len1 = 0;
      int rc = dye ? euler_render_grids_rgb(o->solid, o->sink, o->count, o->cr, o->cg, o->cb, X, Y, win[w][0], win[w][1], NULL, 0, &len1)
                   : euler_render_grids(o->solid, o->sink, o->count, X, Y, win[w][0], win[w][1], NULL, 0, &len1);
      CHECK(rc == EULER_OK && len1 >= 0);
      char* a = malloc((size_t)len1 + 1);
      char* b = malloc((size_t)len1 + 1);
      CHECK(a && b);
      int32_t len2 = 0;
      rc = dye ? euler_render_grids_rgb(o->solid, o->sink, o->count, o->cr, o->cg, o->cb, X, Y, win[w][0], win[w][1], a, len1, &len2)
               : euler_render_grids(o->solid, o->sink, o->count, X, Y, win[w][0], win[w][1], a, len1, &len2);
      CHECK(rc == EULER_OK && len2 == len1);
      CHECK(eo_render_rows(o, win[w][0], win[w][1], b, len1) == len1);
      CHECK(memcmp(a, b, (size_t)len1) == 0);
      /* a buffer that is too short must be respected and the full length still reported */
      int32_t len3 = 0;
      rc = dye ? euler_render_grids_rgb(o->solid, o->sink, o->count, o->cr, o->cg, o->cb, X, Y, win[w][0], win[w][1], a, len1 / 2, &len3)
               : euler_render_grids(o->solid, o->sink, o->count, X, Y, win[w][0], win[w][1], a, len1 / 2, &len3);
      CHECK(rc == EULER_OK && len3 == len1);
      free(a); free(b);
    }
  }
  free(mk); free(g);
  eo_destroy(o);
  return 0;
}

int main(void) {
  /* lines longer than the grid, no trailing newline, empty lines, every cell kind, stray characters */
  const char* t1 = "XXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXX\nX  000   ??  =  X\n\nX 0?0=Xabc\nXXXX";
  const char* t2 = "";
  const char* t3 = "\n\n\n";
  const char* t4 = "0";
  char t5[4000];
  for (int i = 0; i < 3999; ++i) t5[i] = "X0?= \n"[(i * 7 + i / 13) % 6];
  t5[3999] = 0;
  const char* texts[] = {t1, t2, t3, t4, t5};
  const int sizes[][2] = {{8, 8}, {13, 9}, {100, 40}, {67, 130}};
  for (unsigned t = 0; t < sizeof texts / sizeof *texts; ++t)
    for (unsigned s = 0; s < sizeof sizes / sizeof *sizes; ++s)
      for (int up = 0; up < 2; ++up)
        if (one_case(texts[t], (int)strlen(texts[t]), sizes[s][0], sizes[s][1], up)) {
          fprintf(stderr, "case text %u size %dx%d upscale %d\n", t, sizes[s][0], sizes[s][1], up);
          return 1;
        }
  /* the half tank */
  {
    const int X = 40, Y = 24;
    size_t C = (size_t)X * Y;
    uint8_t* g = calloc(4, C);
    extern int euler_half_tank_grids(int32_t, int32_t, uint8_t*, uint8_t*, uint8_t*, uint8_t*);
    CHECK(g && euler_half_tank_grids(X, Y, g, g + C, g + 2 * C, g + 3 * C) == EULER_OK);
    eo_sim* o = eo_create(X, Y);
    CHECK(o && eo_load_half_tank(o) == 0);
    CHECK(memcmp(g, o->solid, C) == 0 && memcmp(g + 2 * C, o->sink, C) == 0);
    for (int f = 0; f < 4; ++f) eo_step(o);
    eo_destroy(o);
    free(g);
  }
  printf("sanitize_host: clean\n");
  return 0;
}
