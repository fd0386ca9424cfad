/*
 * slb_globals.c -- the host-owned parameters that cross the boundary.
 *
 * The reference keeps these as plain globals, set by the command-line parser
 * and by main(), and the device side imports them as externs (reference:
 * src/boltzmann_cli.c:20-64, src/boltzmann_solver.c:42-65,
 * src/boltzmann_gpu.cu:40-44).  They live in the library so that a
 * foreign-function caller can set them before load_data().
 */
#include <stdio.h>
#include "slb_host.h"

int   quiet   = 0;
int   display = -999;
int   device  = 0;

float host_E_dc = -999, host_E_omega = -999, host_omega = -999;
float host_mu = -999, host_alpha = -999;
int   host_N = -999;
float PhiYmin = -999, PhiYmax = -999;
float host_B = -999;
float t_start = -999;       /* the t-max= argument: averaging starts here      */
float frame_start = 0;
char *out_file = (char *)"-";
FILE *out = NULL;
FILE *read_from = NULL;

float t_max = -999;         /* t_start + one period of the a/c field            */
float host_dPhi = 0;
int   host_M = 3069;
int   MSIZE, PADDED_MSIZE, MP1, host_TMSIZE, NSIZE;
float host_dt = 0.001;
float host_bdt, host_nu_tilde, host_nu2, host_nu;
int   ranks = 1;            /* ranks=: processes (GPUs) of the run; not in the reference */
