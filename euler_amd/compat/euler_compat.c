/*
 * euler_compat.c — the reference's own three entry points over libeuler_hip.so ("a compatibility shim
 * exporting the reference's three names over one default handle").
 *
 * A maintainer of cgmb/euler adds THIS FILE to the build and links -leuler_hip; nothing else in the reference
 * changes.  It defines, with the reference's exact signatures and types,
 *     void sim_init(args_t in);          main.c:209    (args_t: main.c:52-55)
 *     void sim_step(void);               main.c:843
 *     void draw_rows(buffer_t* buf);     main.c:914    (buffer_t: misc/terminal.h:3-6)
 *     void colorize(void);               main.c:187    (the 'r' key, main.c:970-973)
 * and uses what stays in the reference: its globals g_wx, g_wy (main.c:28-29), g_rainbow_enabled (main.c:76),
 * g_pause, g_temp_unpause_counter, g_frame_count (main.c:87-89) and buffer_append / die from misc/terminal.c.
 * main(), process_keypress(), draw(), the pacing and the terminal code keep running as they are and now drive the GPU.
 * The reference's own definitions of the four functions are dropped (deleted, or made weak: oracle/Makefile
 * `ref_on_hip` does the latter with objcopy on the unmodified main.c, which is how the tests prove the drop-in).
 *
 * Grid size: the reference's compile-time 100 x 40 by default; EULER_COMPAT_SIZE=XxY (environment) selects another,
 * EULER_COMPAT_UPSCALE=1 resamples the scenario text onto it.
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "euler.h"

typedef struct args_t {        /* main.c:52-55 */
  const char* scenario_file;
  bool rainbow;
} args_t;

typedef struct buffer_t {      /* misc/terminal.h:3-6 */
  char* data;
  int len;
} buffer_t;

/* what stays in the reference */
extern int g_wx, g_wy;
extern bool g_rainbow_enabled, g_pause;
extern uint32_t g_temp_unpause_counter;
extern uint16_t g_frame_count;
void buffer_append(buffer_t* buf, const char* s, int len);
void die(const char* msg);

static euler_sim* g_sim;

euler_sim* euler_compat_handle(void) { return g_sim; }   /* tools and tests: the handle behind the three names */

void sim_init(args_t in) {
  euler_config cfg;
  euler_config_default(&cfg);                      /* X = 100, Y = 40, 100 iterations, tol 1e-6f, 8 substeps of 0.1 s */
  int upscale = 0;
  const char* size = getenv("EULER_COMPAT_SIZE");
  if (size && sscanf(size, "%dx%d", &cfg.X, &cfg.Y) != 2) {
    fprintf(stderr, "EULER_COMPAT_SIZE=%s: expected XxY\n", size);
    exit(1);
  }
  const char* up = getenv("EULER_COMPAT_UPSCALE");
  if (up && up[0] == '1') upscale = 1;
  /* the preconditioner of the pressure solve (include/euler.h EULER_PRECOND_*): "reference" (default) = main.c:577-627 with bit-identical iterates; "multilevel" with
   * EULER_COMPAT_MAX_ITERATIONS above the reference's 100 lets a large grid's solves converge (30-60 iterations whatever the size) */
  const char* solver = getenv("EULER_COMPAT_SOLVER");
  if (solver) {
    if (!strcmp(solver, "reference")) cfg.precond = EULER_PRECOND_IC0;
    else if (!strcmp(solver, "tile")) cfg.precond = EULER_PRECOND_IC0_TILE;
    else if (!strcmp(solver, "tile-fp32")) { cfg.precond = EULER_PRECOND_IC0_TILE; cfg.pcg_precision = EULER_PCG_F32; cfg.dot_mode = EULER_DOT_TREE; }   /* solver vectors in float (small grids: the resident solver) */
    else if (!strcmp(solver, "two-level")) cfg.precond = EULER_PRECOND_IC0_TILE2;
    else if (!strcmp(solver, "multilevel")) cfg.precond = EULER_PRECOND_IC0_TILE_MG;
    else { fprintf(stderr, "EULER_COMPAT_SOLVER=%s: expected reference, tile, tile-fp32, two-level or multilevel\n", solver); exit(1); }
  }
  const char* maxit = getenv("EULER_COMPAT_MAX_ITERATIONS");
  if (maxit && (cfg.max_iterations = atoi(maxit)) < 1) { fprintf(stderr, "EULER_COMPAT_MAX_ITERATIONS=%s: expected a positive count\n", maxit); exit(1); }
  cfg.rainbow = g_rainbow_enabled;                 /* main() sets the global before sim_init (main.c:1020) */
  if (g_sim) { euler_destroy(g_sim); g_sim = NULL; }
  if (euler_create(&cfg, &g_sim) != EULER_OK || euler_load_scenario_file(g_sim, in.scenario_file, upscale) != EULER_OK) {
    fprintf(stderr, "%s\n", euler_last_error());   /* "Could not load <file>!" like main.c:213 */
    exit(1);                                       /* the reference's own error convention (main.c:214) */
  }
}

void sim_step(void) {
  if (g_pause && g_temp_unpause_counter == 0) return;          /* main.c:844-846 */
  if (euler_step(g_sim) != EULER_OK) die(euler_last_error());
  if (g_temp_unpause_counter) g_temp_unpause_counter--;        /* main.c:896-898 */
  g_frame_count++;                                             /* main.c:899 */
}

void draw_rows(buffer_t* buf) {
  int32_t len = 0;
  if (euler_render(g_sim, g_wx, g_wy, NULL, 0, &len) != EULER_OK) die(euler_last_error());
  char* tmp = (char*)malloc((size_t)len + 1);
  if (!tmp) die("failed to allocate the frame");
  if (euler_render(g_sim, g_wx, g_wy, tmp, len, &len) != EULER_OK) die(euler_last_error());
  buffer_append(buf, tmp, len);                                /* the bytes draw_rows() appends (main.c:914-951) */
  free(tmp);
}

void colorize(void) {
  if (g_sim && g_rainbow_enabled && euler_colorize(g_sim) != EULER_OK) die(euler_last_error());
}
