/*
 * euler — command-line front end over the C ABI of libeuler_hip.so.
 *
 * Mirrors the reference's program shape (main.c:982-1042): `euler [options] <scenario>` loads a
 * scenario text file, then loops  step -> render  at 10 frames per second.  The terminal handling
 * of the reference (raw mode, key presses, SIGWINCH, misc/terminal.c) is out of scope; this front
 * end writes frames with plain ANSI codes, or dumps them for tests with --dump.
 *
 *   euler [--size XxY] [--upscale] [--frames N] [--window WxH] [--dump] [--no-pace]
 *         [--resume FILE] [--checkpoint FILE] <scenario>
 * --resume continues from a state snapshot (include/euler.h) instead of the scenario's initial state
 * (the scenario argument may then be omitted); --checkpoint writes one after the last frame.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "euler.h"

static void usage(const char* argv0) {
  fprintf(stderr, "usage: %s [--size XxY] [--upscale] [--frames N] [--window WxH] [--dump] [--no-pace] [--resume FILE] [--checkpoint FILE] <scenario>\n", argv0);
}

int main(int argc, char** argv) {
  euler_config cfg;
  euler_config_default(&cfg);
  int upscale = 0, frames = -1, wx = 98, wy = 38, dump = 0, pace = 1;
  const char* scenario = NULL;
  const char* resume = NULL;
  const char* checkpoint = NULL;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--size") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &cfg.X, &cfg.Y) != 2) { usage(argv[0]); return 1; } }
    else if (!strcmp(argv[i], "--window") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &wx, &wy) != 2) { usage(argv[0]); return 1; } }
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--upscale")) upscale = 1;
    else if (!strcmp(argv[i], "--dump")) dump = 1;
    else if (!strcmp(argv[i], "--no-pace")) pace = 0;
    else if (!strcmp(argv[i], "--resume") && i + 1 < argc) resume = argv[++i];
    else if (!strcmp(argv[i], "--checkpoint") && i + 1 < argc) checkpoint = argv[++i];
    else if (argv[i][0] == '-') { fprintf(stderr, "Unrecognized input: %s\n", argv[i]); return 1; }   /* main.c:995 */
    else scenario = argv[i];
  }
  if (!scenario && !resume) { usage(argv[0]); return 1; }                                                       /* main.c:986-989 */

  euler_sim* sim = NULL;
  if (euler_create(&cfg, &sim) != EULER_OK ||
      (resume ? euler_load_state(sim, resume) : euler_load_scenario_file(sim, scenario, upscale)) != EULER_OK) {
    fprintf(stderr, "%s\n", euler_last_error());
    return 1;
  }
  int32_t cap = 0;
  euler_render(sim, wx, wy, NULL, 0, &cap);
  cap = cap * 2 + 4096;
  char* buf = (char*)malloc((size_t)cap);
  if (!buf) return 1;
  struct timespec next;
  clock_gettime(CLOCK_MONOTONIC, &next);
  for (int f = 0; frames < 0 || f <= frames; ++f) {
    int32_t len = 0;
    if (f > 0 && euler_step(sim) != EULER_OK) { fprintf(stderr, "%s\n", euler_last_error()); return 1; }
    if (euler_render(sim, wx, wy, buf, cap, &len) != EULER_OK || len > cap) { fprintf(stderr, "%s\n", euler_last_error()); return 1; }
    if (dump) {
      printf("--- frame %d (%d bytes)\n", f, (int)len);
      fwrite(buf, 1, (size_t)len, stdout);
      printf("\n");
    } else {
      fputs("\x1b[H", stdout);           /* reposition cursor (misc/terminal.h T_REPOSITION_CURSOR) */
      fwrite(buf, 1, (size_t)len, stdout);
      fputs("\x1b[?25l", stdout);        /* hide cursor */
      fflush(stdout);
    }
    if (pace && !dump) {                  /* 10 frames per second like main.c:1036 */
      next.tv_nsec += 100000000L;
      if (next.tv_nsec >= 1000000000L) { next.tv_nsec -= 1000000000L; next.tv_sec += 1; }
      clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, NULL);
    }
  }
  if (checkpoint && euler_save_state(sim, checkpoint) != EULER_OK) { fprintf(stderr, "%s\n", euler_last_error()); return 1; }
  euler_stats st;
  if (euler_get_stats(sim, &st) == EULER_OK)
    fprintf(stderr, "frames %llu substeps %llu pcg_iterations %llu markers %llu\n", (unsigned long long)st.frames,
            (unsigned long long)st.total_substeps, (unsigned long long)st.total_pcg_iterations, (unsigned long long)st.n_markers);
  free(buf);
  euler_destroy(sim);
  return 0;
}
