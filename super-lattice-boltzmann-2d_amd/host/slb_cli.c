/*
 * slb_cli.c -- name=value command line of boltzmann_solver and the stdin
 * re-parameterisation reader.  Same parameter names, defaults, required set,
 * validation messages and o= conventions as the reference parser
 * (reference: src/boltzmann_cli.c:70-189); table-driven instead of an if-chain.
 */
#include <stdlib.h>
#include <string.h>
#include "slb_host.h"

enum kind { K_INT, K_FLOAT, K_FLAG, K_STR };

struct option {
  const char *name;
  enum kind   kind;
  void       *target;
  int         required;
};

static char *rf_name = NULL;

static const struct option options[] = {
  {"display",     K_INT,   &display,      1},
  {"E_dc",        K_FLOAT, &host_E_dc,    1},
  {"E_omega",     K_FLOAT, &host_E_omega, 1},
  {"omega",       K_FLOAT, &host_omega,   1},
  {"mu",          K_FLOAT, &host_mu,      1},
  {"alpha",       K_FLOAT, &host_alpha,   1},
  {"n-harmonics", K_INT,   &host_N,       1},
  {"PhiYmin",     K_FLOAT, &PhiYmin,      1},
  {"PhiYmax",     K_FLOAT, &PhiYmax,      1},
  {"B",           K_FLOAT, &host_B,       1},
  {"t-max",       K_FLOAT, &t_start,      1},
  {"frame-start", K_FLOAT, &frame_start,  0},
  {"dt",          K_FLOAT, &host_dt,      0},
  {"g-grid",      K_INT,   &host_M,       0},
  {"read-from",   K_STR,   &rf_name,      0},
  {"quiet",       K_FLAG,  &quiet,        0},
  {"device",      K_INT,   &device,       0},
  {"ranks",       K_INT,   &ranks,        0},   /* not in the reference: phi_y slabs, one process per GPU */
  {"o",           K_STR,   &out_file,     0},
};
#define N_OPTIONS ((int)(sizeof(options) / sizeof(options[0])))

static void die(const char *msg)
{
  fputs(msg, stderr);
  exit(EXIT_FAILURE);
}

static void assign(const struct option *o, const char *value)
{
  switch (o->kind) {
    case K_INT:
      /* n-harmonics goes through strtod in the reference (accepts "32.0") */
      *(int *)o->target = strcmp(o->name, "n-harmonics") == 0 ? (int)strtod(value, NULL) : atoi(value);
      break;
    case K_FLOAT: *(float *)o->target = strtod(value, NULL); break;
    case K_FLAG:  *(int *)o->target = 1; break;
    case K_STR:   *(char **)o->target = strdup(value); break;
  }
}

void parse_cmd(int argc, char **argv)
{
  for (int i = 1; i < argc; i++) {
    /* a token is name=value; parsing stops at the first token that is not */
    char *name = argv[i];
    while (*name == '=') name++;
    char *eq = strchr(name, '=');
    if (*name == 0 || eq == NULL) break;
    *eq = 0;
    char *value = eq + 1;
    while (*value == '=') value++;
    if (*value == 0) break;
    char *end = strchr(value, '=');
    if (end) *end = 0;

    for (int k = 0; k < N_OPTIONS; k++) {
      if (strcmp(options[k].name, name) == 0) { assign(&options[k], value); break; }
    }
  }

  for (int k = 0; k < N_OPTIONS; k++) {
    if (!options[k].required) continue;
    const double v = options[k].kind == K_INT ? (double)*(int *)options[k].target
                                              : (double)*(float *)options[k].target;
    if (v < -900) {
      fprintf(stderr, "ERROR: Parameter \"%s\" must be set.\n", options[k].name);
      exit(EXIT_FAILURE);
    }
  }

  if (display != 3 && display != 4 && display != 7 && display != 8 && display != 9 && display != 77)
    die("ERROR: Invalid value of display= parameter. Possible values are 3, 4, 8 or 77.\n");
  if (t_start <= 0)
    die("ERROR: Invalid value of t-max= parameter. it must be greater than 0.\n");

  if (rf_name != NULL) {
    if (strcmp(rf_name, "stdin") != 0) die("ERROR: Invalid value of read-from=\n");
    read_from = stdin;
  }

  if (strcmp(out_file, "stdout") == 0) {
    out = stdout;
  } else if (strcmp(out_file, "stderr") == 0) {
    out = stderr;
  } else {
    out = out_file[0] == '+' ? fopen(out_file + 1, "a") : fopen(out_file, "w");
    if (out == NULL) {
      fprintf(stderr, "ERROR: Faled to open file \"%s\" for writing.\n", out_file);
      perror("ERROR");
      exit(EXIT_FAILURE);
    }
  }
}

/* Reads "name value timeout" records from read_from until one parses; "exit"
 * ends the run (returns -999).  One of E_dc, E_omega, omega, mu, alpha, B is
 * replaced per record; other names are read and ignored. */
float scan_for_new_parameters(void)
{
  static const char *settable[] = {"E_dc", "E_omega", "omega", "mu", "alpha", "B"};
  char  name[80];
  float value, timeout;
  for (;;) {
    const int got = fscanf(read_from, "%79s %f %f", name, &value, &timeout);
    if (got == EOF) return -999;
    if (got == 1 && strcmp(name, "exit") == 0) return -999;
    if (got != 3) continue;
    for (int k = 0; k < N_OPTIONS; k++) {
      if (strcmp(options[k].name, name) != 0) continue;
      for (unsigned q = 0; q < sizeof(settable) / sizeof(settable[0]); q++)
        if (strcmp(settable[q], name) == 0) *(float *)options[k].target = value;
    }
    return timeout;
  }
}
