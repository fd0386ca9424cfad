/*
 * slb_oracle_main.c -- command-line front end of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Takes the reference's name=value arguments
 * (reference: src/boltzmann_cli.c:105-122) and writes what the reference CPU
 * solver writes for display=3 / 4 / 77, and the frame files of display=7 / 8 / 9
 * as the reference's CUDA host writes them, built from the oracle fields.  Extra argument: threads=K (OpenMP threads).
 */
#include "slb_oracle.h"

#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
  orc_params p;
  memset(&p, 0, sizeof(p));
  int display = -999, threads = 1, from_stdin = 0;
  long max_steps = -1;
  const char *o = "-";
  p.N = -999; p.M = 3069; p.dt = 0.001;
  p.E_dc = p.E_omega = p.omega = p.mu = p.alpha = p.B = p.PhiYmin = p.PhiYmax = p.t_start = -999;

  for (int i = 1; i < argc; i++) {
    char *eq = strchr(argv[i], '=');
    if (!eq || eq == argv[i] || !eq[1]) break;
    *eq = 0;
    const char *k = argv[i], *v = eq + 1;
    if      (!strcmp(k, "display"))     display   = atoi(v);
    else if (!strcmp(k, "E_dc"))        p.E_dc    = strtod(v, NULL);
    else if (!strcmp(k, "E_omega"))     p.E_omega = strtod(v, NULL);
    else if (!strcmp(k, "omega"))       p.omega   = strtod(v, NULL);
    else if (!strcmp(k, "mu"))          p.mu      = strtod(v, NULL);
    else if (!strcmp(k, "alpha"))       p.alpha   = strtod(v, NULL);
    else if (!strcmp(k, "n-harmonics")) p.N       = strtod(v, NULL);
    else if (!strcmp(k, "PhiYmin"))     p.PhiYmin = strtod(v, NULL);
    else if (!strcmp(k, "PhiYmax"))     p.PhiYmax = strtod(v, NULL);
    else if (!strcmp(k, "B"))           p.B       = strtod(v, NULL);
    else if (!strcmp(k, "t-max"))       p.t_start = strtod(v, NULL);
    else if (!strcmp(k, "dt"))          p.dt      = strtod(v, NULL);
    else if (!strcmp(k, "g-grid"))      p.M       = atoi(v);
    else if (!strcmp(k, "o"))           o         = v;
    else if (!strcmp(k, "threads"))     threads   = atoi(v);
    else if (!strcmp(k, "max-steps"))   max_steps = atol(v);
    else if (!strcmp(k, "frame-start")) orc_set_frame_start(strtod(v, NULL));
    else if (!strcmp(k, "read-from"))   from_stdin = !strcmp(v, "stdin");
  }
  if (display < -900 || p.N < -900 || p.E_dc < -900 || p.E_omega < -900 || p.omega < -900 ||
      p.mu < -900 || p.alpha < -900 || p.PhiYmin < -900 || p.PhiYmax < -900 || p.B < -900 ||
      p.t_start < -900) {
    fprintf(stderr, "ERROR: all of display E_dc E_omega omega mu alpha n-harmonics PhiYmin PhiYmax B t-max must be set.\n");
    return EXIT_FAILURE;
  }
  FILE *out;
  if      (!strcmp(o, "stdout")) out = stdout;
  else if (!strcmp(o, "stderr")) out = stderr;
  else if (o[0] == '+')          out = fopen(o + 1, "a");
  else                           out = fopen(o, "w");
  if (!out) { perror("ERROR"); return EXIT_FAILURE; }

  orc_set_threads(threads);
  orc_state s;
  if (orc_state_init(&s, &p) != 0) { fprintf(stderr, "ERROR: out of memory\n"); return EXIT_FAILURE; }
  int rc = orc_solver_run(&s, display, out, stdout, max_steps);
  while (rc == 0 && from_stdin && display != 3 && display != 8) {
    /* "name value timeout" records; "exit" or end of input ends the run */
    char name[80];
    float value, timeout;
    const int got = fscanf(stdin, "%79s %f %f", name, &value, &timeout);
    if (got == EOF || (got == 1 && !strcmp(name, "exit"))) break;
    if (got != 3) continue;
    orc_solver_continue(&s, name, value, timeout);
    rc = orc_solver_run(&s, display, out, stdout, max_steps);
  }
  if (out != stdout && out != stderr) fclose(out);
  orc_state_free(&s);
  return rc == 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
