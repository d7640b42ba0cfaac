/*
 * euler — command-line front end over the C ABI of libeuler_hip.so.
 *
 * Mirrors the reference's program (main.c:961-1042): `euler [--rainbow] <scenario>` loads a scenario
 * text file, then loops  key -> step -> wait -> draw  at 10 frames per second.  On a terminal it behaves
 * like the reference: raw mode (misc/terminal.c:62-83), the window size from TIOCGWINSZ and SIGWINCH
 * (main.c:1004-1014), and the keys  p pause / f advance one frame while paused / r recolour the dye /
 * q quit  (main.c:961-980) with the pause gate of sim_step (main.c:844-846, 896-898).  Without a
 * terminal, or with --dump, it writes the frames as plain bytes (tests); --keys feeds the same key
 * handler one character per frame ('.' = no key) so that the gate is testable without a tty.
 *
 *   euler [--rainbow] [--size XxY] [--upscale] [--frames N] [--window WxH] [--dump] [--no-pace]
 *         [--keys STRING] [--resume FILE] [--checkpoint FILE] <scenario>
 * --resume continues from a state snapshot (include/euler.h) instead of the scenario's initial state
 * (the scenario argument may then be omitted); --checkpoint writes one after the last frame.
 */
#include <errno.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ioctl.h>
#include <termios.h>
#include <time.h>
#include <unistd.h>

#include "euler.h"

static void usage(const char* argv0) {
  fprintf(stderr, "usage: %s [--rainbow] [--size XxY] [--upscale] [--frames N] [--window WxH] [--dump] [--no-pace] [--keys STRING] "
                  "[--resume FILE] [--checkpoint FILE] [--solver reference|tile|tile-fp32|two-level|multilevel] [--max-iterations N] <scenario>\n", argv0);
}

/* ---- terminal (misc/terminal.c) ------------------------------------------------------------ */
static struct termios g_orig_termios;
static int g_raw = 0;
static volatile sig_atomic_t g_resized = 0;

static void write_all(const char* s, size_t n) {
  while (n) {
    ssize_t k = write(STDOUT_FILENO, s, n);
    if (k < 0) { if (errno == EINTR) continue; return; }
    s += k; n -= (size_t)k;
  }
}
static void restore_terminal(void) {
  if (!g_raw) return;
  tcsetattr(STDIN_FILENO, TCSAFLUSH, &g_orig_termios);
  write_all("\x1b[?25h", 6);   /* show cursor */
  g_raw = 0;
}
static int enable_raw_mode(void) {
  if (tcgetattr(STDIN_FILENO, &g_orig_termios) == -1) return -1;
  struct termios raw = g_orig_termios;
  raw.c_iflag &= ~(tcflag_t)(BRKINT | ICRNL | INPCK | ISTRIP | IXON);
  raw.c_oflag &= ~(tcflag_t)(OPOST);
  raw.c_cflag |= (tcflag_t)(CS8);
  raw.c_lflag &= ~(tcflag_t)(ECHO | ICANON | IEXTEN | ISIG);
  raw.c_cc[VMIN] = 0;
  raw.c_cc[VTIME] = 0;
  if (tcsetattr(STDIN_FILENO, TCSAFLUSH, &raw) == -1) return -1;
  g_raw = 1;
  atexit(restore_terminal);
  return 0;
}
static void on_winch(int sig) { (void)sig; g_resized = 1; }
static int window_size(int* wx, int* wy) {
  struct winsize ws;
  if (ioctl(STDOUT_FILENO, TIOCGWINSZ, &ws) == -1 || ws.ws_col == 0) return -1;
  *wx = ws.ws_col; *wy = ws.ws_row;
  return 0;
}

/* ---- the reference's loop state (main.c:85-88) ----------------------------------------------- */
typedef struct app {
  euler_sim* sim;
  int pause;                    /* g_pause */
  unsigned temp_unpause;        /* g_temp_unpause_counter */
  int rainbow;
} app_t;

/* process_keypress (main.c:961-980); returns 0 on 'q' */
static int handle_key(app_t* a, char c) {
  if (c == 'p') a->pause = !a->pause;
  else if (c == 'f') a->temp_unpause++;
  else if (c == 'r') { if (a->rainbow && euler_colorize(a->sim) != EULER_OK) fprintf(stderr, "%s\n", euler_last_error()); }
  else if (c == 'q') return 0;
  return 1;
}

/* sim_step's gate (main.c:844-846, 896-898) around euler_step */
static int gated_step(app_t* a) {
  if (a->pause && a->temp_unpause == 0) return EULER_OK;
  int rc = euler_step(a->sim);
  if (a->temp_unpause) a->temp_unpause--;
  return rc;
}

int main(int argc, char** argv) {
  euler_config cfg;
  euler_config_default(&cfg);
  int upscale = 0, frames = -1, wx = 98, wy = 38, dump = 0, pace = 1, window_given = 0;
  const char* scenario = NULL;
  const char* resume = NULL;
  const char* checkpoint = NULL;
  const char* keys = NULL;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--size") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &cfg.X, &cfg.Y) != 2) { usage(argv[0]); return 1; } }
    else if (!strcmp(argv[i], "--window") && i + 1 < argc) { if (sscanf(argv[++i], "%dx%d", &wx, &wy) != 2) { usage(argv[0]); return 1; } window_given = 1; }
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rainbow")) cfg.rainbow = 1;                                                     /* main.c:992 */
    else if (!strcmp(argv[i], "--upscale")) upscale = 1;
    else if (!strcmp(argv[i], "--dump")) dump = 1;
    else if (!strcmp(argv[i], "--no-pace")) pace = 0;
    else if (!strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
    else if (!strcmp(argv[i], "--resume") && i + 1 < argc) resume = argv[++i];
    else if (!strcmp(argv[i], "--checkpoint") && i + 1 < argc) checkpoint = argv[++i];
    /* the pressure solver's preconditioner (include/euler.h EULER_PRECOND_*): `reference` (default) = main.c:577-627, bit-identical iterates;
     * the others reach the same pressure where the solve converges - `multilevel` in 30-60 iterations whatever the grid size, so with
     * --max-iterations lifted above the reference's 100 (main.c:735) a large grid is actually SOLVED each substep */
    else if (!strcmp(argv[i], "--solver") && i + 1 < argc) {
      const char* v = argv[++i];
      if (!strcmp(v, "reference")) cfg.precond = EULER_PRECOND_IC0;
      else if (!strcmp(v, "tile")) cfg.precond = EULER_PRECOND_IC0_TILE;
      else if (!strcmp(v, "tile-fp32")) { cfg.precond = EULER_PRECOND_IC0_TILE; cfg.pcg_precision = EULER_PCG_F32; cfg.dot_mode = EULER_DOT_TREE; }   /* solver vectors in float (resident solver: small grids) */
      else if (!strcmp(v, "two-level")) cfg.precond = EULER_PRECOND_IC0_TILE2;
      else if (!strcmp(v, "multilevel")) cfg.precond = EULER_PRECOND_IC0_TILE_MG;
      else { usage(argv[0]); return 1; }
    }
    else if (!strcmp(argv[i], "--max-iterations") && i + 1 < argc) { cfg.max_iterations = atoi(argv[++i]); if (cfg.max_iterations < 1) { usage(argv[0]); return 1; } }
    else if (argv[i][0] == '-') { fprintf(stderr, "Unrecognized input: %s\n", argv[i]); return 1; }   /* main.c:995 */
    else scenario = argv[i];
  }
  if (!scenario && !resume) { usage(argv[0]); return 1; }                                                       /* main.c:986-989 */

  const int interactive = !dump && isatty(STDIN_FILENO) && isatty(STDOUT_FILENO);
  if (interactive && !window_given) {
    if (window_size(&wx, &wy) == -1) { perror("get_window_size"); return 1; }                                    /* main.c:1004-1008 */
    struct sigaction sa;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = 0;
    sa.sa_handler = on_winch;
    sigaction(SIGWINCH, &sa, 0);
  }

  app_t app;
  memset(&app, 0, sizeof app);
  app.rainbow = cfg.rainbow;
  if (euler_create(&cfg, &app.sim) != EULER_OK ||
      (resume ? euler_load_state(app.sim, resume) : euler_load_scenario_file(app.sim, scenario, upscale)) != EULER_OK) {
    fprintf(stderr, "%s\n", euler_last_error());
    return 1;
  }
  if (interactive) {
    if (enable_raw_mode() == -1) { perror("failed to enable raw mode"); return 1; }
    write_all("\x1b[2J\x1b[H", 7);      /* clear_screen_now */
  }
  int32_t cap = 0;
  char* buf = NULL;
  struct timespec next;
  clock_gettime(CLOCK_MONOTONIC, &next);
  size_t key_pos = 0;
  int rc_exit = 0;
  for (int f = 0; frames < 0 || f <= frames; ++f) {
    if (f > 0) {
      /* one key per frame: a scripted one (--keys) or whatever the terminal has (non-blocking read) */
      char c = '\0';
      if (keys) { if (keys[key_pos]) c = keys[key_pos++]; }
      else if (interactive) { if (read(STDIN_FILENO, &c, 1) == -1 && errno != EAGAIN && errno != EINTR) { perror("read"); rc_exit = 1; break; } }
      if (!handle_key(&app, c)) break;
      if (gated_step(&app) != EULER_OK) { fprintf(stderr, "%s\n", euler_last_error()); rc_exit = 1; break; }
      if (pace && !dump) {                  /* 10 frames per second (main.c:1036, misc/time.c:17-32) */
        next.tv_nsec += 100000000L;
        if (next.tv_nsec >= 1000000000L) { next.tv_nsec -= 1000000000L; next.tv_sec += 1; }
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if (now.tv_sec > next.tv_sec || (now.tv_sec == next.tv_sec && now.tv_nsec > next.tv_nsec)) next = now;   /* running late: no catch-up burst */
        else clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, NULL);
      }
    }
    if (g_resized) {                        /* handle_window_size_changed (main.c:1010-1014) */
      g_resized = 0;
      if (window_size(&wx, &wy) == 0) write_all("\x1b[2J\x1b[H", 7);
    }
    int32_t len = 0;
    if (euler_render(app.sim, wx, wy, NULL, 0, &len) != EULER_OK) { fprintf(stderr, "%s\n", euler_last_error()); rc_exit = 1; break; }
    if (len > cap) {
      cap = len + 4096;
      char* nb = (char*)realloc(buf, (size_t)cap);
      if (!nb) { rc_exit = 1; break; }
      buf = nb;
    }
    if (euler_render(app.sim, wx, wy, buf, cap, &len) != EULER_OK || len > cap) { fprintf(stderr, "%s\n", euler_last_error()); rc_exit = 1; break; }
    if (dump) {
      printf("--- frame %d (%d bytes)\n", f, (int)len);
      fwrite(buf, 1, (size_t)len, stdout);
      printf("\n");
    } else {                               /* draw (main.c:953-959) */
      fflush(stdout);
      write_all("\x1b[H", 3);              /* reposition cursor */
      write_all(buf, (size_t)len);
      write_all("\x1b[?25l", 6);           /* hide cursor */
    }
  }
  if (interactive) { write_all("\x1b[2J\x1b[H", 7); restore_terminal(); }
  if (!rc_exit && checkpoint && euler_save_state(app.sim, checkpoint) != EULER_OK) { fprintf(stderr, "%s\n", euler_last_error()); rc_exit = 1; }
  euler_stats st;
  if (euler_get_stats(app.sim, &st) == EULER_OK)
    fprintf(stderr, "frames %llu substeps %llu pcg_iterations %llu markers %llu\n", (unsigned long long)st.frames,
            (unsigned long long)st.total_substeps, (unsigned long long)st.total_pcg_iterations, (unsigned long long)st.n_markers);
  free(buf);
  euler_destroy(app.sim);
  return rc_exit;
}
