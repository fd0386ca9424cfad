/*
 * slb_host.h -- host side of the solver: command line, derived constants,
 * initial state, time loop and the display=4/8/77 reductions.  Mirrors the
 * reference's host program (reference: src/boltzmann_solver.c,
 * src/boltzmann_cli.c) on top of the boundary in include/slb_abi.h.
 */
#ifndef SLB_HOST_H
#define SLB_HOST_H

#include <stddef.h>
#include <stdio.h>
#include "../../include/slb_abi.h"
#include "../../include/slb_slabs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the reference's PI literal (src/constants.h:11): the double nearest pi */
#define SLB_PI 3.141592653589793115998

/* command-line state (reference: src/boltzmann_cli.c:20-64) */
extern int   quiet, display, device;
extern float frame_start, t_max;
extern char *out_file;
extern FILE *out, *read_from;

/* reference: src/boltzmann_cli.c:93-189 and :70-91 */
void  parse_cmd(int argc, char **argv);
float scan_for_new_parameters(void);

/* Modified Bessel function I_n(x), integer n (the reference calls GSL:
 * src/boltzmann_solver.c:122, 359-360, 426-427). */
double slb_bessel_In(int n, double x);

/* host_dPhi, NSIZE, MSIZE, PADDED_MSIZE, MP1, host_TMSIZE, host_nu, host_nu2,
 * host_nu_tilde, host_bdt from the command-line values, then load_data()
 * (reference: src/boltzmann_solver.c:97-117). */
void slb_derive_and_load(void);

/* ---- solver object: device arrays + ping-pong indices + loop time ------- */
struct slb_solver {
  int   N, M;            /* global n-harmonics, g-grid                           */
  int   rank, nranks;    /* phi_y slab index / count                             */
  int   W;               /* columns this slab owns on the half-step lattice      */
  int   pitch;           /* floats per row of every device array                 */
  int   m_offset;        /* global column of local column 0                      */
  slb_slab main_slab, half_slab, av_slab;
  float *a0, *a[4], *b[4];   /* device */
  float *av_data;            /* device, 6 floats */
  double *sums;              /* device, 4 doubles */
  float *halo_buf;           /* device, 4 x 2*(N+1) floats: send L, send R, recv L, recv R */
  int   current, next, current_hs, next_hs;
  float t, frame_time;
  long  steps;
  int   fused;           /* run_timed / the program use the one-launch step       */
  int   G;               /* ghost columns per side; owned columns are G..G+W-1    */
  /* spare arrays for the double step (outputs must not alias inputs): their
   * never-written cells mirror a/b[spare_of] and a/b[spare_of_hs] */
  float *spare_a, *spare_b, *spare_ah, *spare_bh;
  int   spare_of, spare_of_hs;
  /* launch stream of this object's calls (slb_solver_set_stream); without one they go where
   * the calling thread's launches go (slb_push_stream / slb_set_stream) */
  void *stream;
  int   has_stream;
  /* a0 in separable form, for the double step (slb_step_fused2_factored): device doubles,
   * N+1 row factors and `pitch` column factors, both NULL unless (float)(row * column) was
   * checked to give the stored a0 in every cell.  a0_rows_adjusted: rows whose factor had to
   * be moved to a neighbouring double for that; -1: the check failed, a0 is loaded. */
  double *a0_row, *a0_col;
  int   a0_rows_adjusted;
};

/* Allocate the slab of rank `rank` of `nranks`, build a0 for its columns
 * (ghost columns included), a[0] <- a0, everything else zero
 * (reference: src/boltzmann_solver.c:119-153).  Uses the host_* globals;
 * call slb_derive_and_load() first.  Returns NULL after printing the reason. */
slb_solver *slb_solver_create(int rank, int nranks);
void        slb_solver_destroy(slb_solver *s);
/* every later launch of this solver's calls goes to `hip_stream` (a hipStream_t) */
void        slb_solver_set_stream(slb_solver *s, void *hip_stream);

/* the tiptoe half step (reference: src/boltzmann_solver.c:158-165) */
void slb_solver_tiptoe(slb_solver *s);

/* the two sweeps at the solver's own ping-pong position, and the swap */
void slb_solver_sweep_main(slb_solver *s, float cos_omega_t, float cos_omega_t_plus_dt);
void slb_solver_sweep_half(slb_solver *s, float cos_omega_t, float cos_omega_t_plus_dt);
void slb_solver_swap(slb_solver *s);
/* lattice 0 = main, 1 = half-step; local columns j_first..j_last clipped to the
 * columns that sweep owns in this slab */
void slb_solver_sweep_range(slb_solver *s, int lattice, int j_first, int j_last,
                            float cos_omega_t, float cos_omega_t_plus_dt);
/* t += dt, frame_time += dt (float accumulation), steps++ */
void slb_solver_advance(slb_solver *s);

/* cosines of one loop iteration at time t, formed as the reference forms them
 * (reference: src/boltzmann_solver.c:204-214): c[0..1] main sweep, c[2..3] half */
void slb_step_cosines(float t, float c[4]);

/* one loop iteration without display work, single slab (no halo exchange) */
void slb_solver_step(slb_solver *s);

/* the same iteration as one fused launch (slb_step_fused); single slab */
void slb_solver_step_fused(slb_solver *s);
/* the fused step for half-step-lattice columns j2_first..j2_last only (no swap,
 * no time advance); c = slb_step_cosines() of the current time */
void slb_solver_step_fused_range(slb_solver *s, int j2_first, int j2_last, const float c[4]);

/* TWO loop iterations in one launch (slb_step_fused2); single slab.  The first
 * call allocates four spare arrays.  Same state afterwards as two
 * slb_solver_step() calls, bit for bit (ping-pong indices included). */
void slb_solver_double_step(slb_solver *s);
/* refresh the spare arrays now (current stream) if the next double step would need to */
void slb_solver_prepare_double_step(slb_solver *s);
/* the same, also writing the rows av() reads of both new main lattices into two
 * [3][pitch] device buffers (a row 0, a row 1, b row 1) */
void slb_solver_double_step_rows(slb_solver *s, float *rows_first, float *rows_second);
void slb_solver_double_step_range_rows(slb_solver *s, int j_first, int j_last, const float c[8],
                                       float *rows_first, float *rows_second);
/* the same in pieces, for drivers that overlap a halo exchange: the launch for
 * the last sweep's columns j_first..j_last (results go to the spare arrays),
 * then the hand-over of the spares and the time advance */
void slb_solver_double_step_range(slb_solver *s, int j_first, int j_last, const float c[8]);
void slb_solver_double_step_finish(slb_solver *s);
/* S loop iterations per launch (1 <= S <= 4; slb_step_multi), whole slab */
void slb_solver_multi_step(slb_solver *s, int S);
void slb_solver_multi_step_range(slb_solver *s, int S, int j_first, int j_last, const float *c);
void slb_solver_multi_step_finish(slb_solver *s, int S);
/* the same (S <= 3), also writing the rows av() reads of each of the S new main lattices into
 * rows[0..S-1] ([3][pitch] device buffers: a row 0, a row 1, b row 1) */
void slb_solver_multi_step_rows(slb_solver *s, int S, float *const *rows);
void slb_solver_multi_step_range_rows(slb_solver *s, int S, int j_first, int j_last, const float *c, float *const *rows);
/* iterations per launch the time loop should use on this slab (3 on wide lattices, else 2;
 * SLB_STEPS_PER_LAUNCH overrides) */
int  slb_solver_steps_per_launch(const slb_solver *s);
/* cosines of S consecutive iterations starting at t, 4 per iteration */
void slb_steps_cosines(float t, int S, float *c);
/* cosines of two consecutive iterations starting at t (see slb_step_cosines) */
void slb_step2_cosines(float t, float c[8]);

/* `steps` iterations bracketed by HIP events on the launch stream; *ms gets the
 * elapsed device time.  Single slab. */
int slb_solver_run_timed(slb_solver *s, int steps, float *ms);

/* halo columns of a lattice that a sweep reads as stencil.
 * which: 0 = a/b[current_hs] (stencil of the main sweep),
 *        1 = a/b[next]       (output of the main sweep, stencil of the half sweep),
 *        2 = a/b[next_hs]    (output of the half sweep, next step's current_hs),
 *        3 = a/b[current], 4 / 5 = the double step's spare main / half-step lattice.
 * side:  0 = left neighbour (send my first ncols owned columns, receive into the
 *            ncols ghost columns left of them), 1 = right neighbour likewise.
 * ncols: 1..G.
 * buf holds 2*ncols*(N+1) floats (device). */
int slb_solver_halo_pack  (slb_solver *s, int which, int side, int ncols, float *buf);
int slb_solver_halo_unpack(slb_solver *s, int which, int side, int ncols, const float *buf);

/* av() over slabs: each slab's partial sums (3 doubles, device) of its own columns of
 * the current main lattice; add them across slabs, then slb_av_update(sums, av_data, t) */
int slb_solver_av_partial(slb_solver *s, double *sums);
/* the three rows av() reads of the current main lattice -> a [3][pitch] device buffer
 * (a row 0, a row 1, b row 1), on the launch stream */
int slb_solver_copy_av_rows(slb_solver *s, float *rows);

/* halo traffic of a double step in one launch per direction: per neighbour a buffer of
 * 2*7*(N+1) floats (3 main-lattice columns, then 4 half-step columns, taken from / put
 * into the spare arrays); NULL for a side without a neighbour */
int slb_solver_halo_pack_double  (slb_solver *s, float *buf_left, float *buf_right);
int slb_solver_halo_unpack_double(slb_solver *s, float *buf_left, float *buf_right);
/* the same for a launch of S = 1..3 steps, between its edge ranges and
 * slb_solver_multi_step_finish(), to the full ghost depth (what any launch that follows may
 * read): G-1 main-lattice columns, then G half-step columns, per neighbour 2*(2G-1)*(N+1)
 * floats; the new lattices are in the spare arrays for S = 2, in a/b[next] and a/b[next_hs]
 * for odd S.  The edge ranges must have formed those columns: the G outermost owned ones. */
int slb_solver_halo_pack_multi  (slb_solver *s, int S, float *buf_left, float *buf_right);
int slb_solver_halo_unpack_multi(slb_solver *s, int S, float *buf_left, float *buf_right);

/* frame values (device sums, host libm factors) of the slab's main-lattice columns for
 * nphi values of phi_x; out = nphi x (main_slab.j_last-j_first+1) floats on the host.
 * source 0: current lattice; 1: equilibrium a0 (cosine terms only) */
int slb_solver_frame_values(slb_solver *s, int source, const float *phi_x, int nphi, float *out);
/* the same with the values left on the device (d_out: nphi x columns floats) */
int slb_solver_frame_values_device(slb_solver *s, int source, const float *phi_x, int nphi, float *d_out);

/* copy rows [row0, row0+nrows) of device array `arr` (0 = a0, 1..4 = a[0..3],
 * 5..8 = b[0..3]) into dst as nrows x pitch floats */
int slb_solver_get_rows(slb_solver *s, int arr, int row0, int nrows, float *dst);

/* copy columns [col0, col0+ncols) of all N+1 rows of device array `arr` into dst
 * as (N+1) x ncols floats */
int slb_solver_get_window(slb_solver *s, int arr, int col0, int ncols, float *dst);

/* ---- display reductions on host rows (single slab) ----------------------
 * rows are passed with the DEVICE pitch; the sums index them the way the CPU
 * solver indexes its unpadded arrays, including its flat-index overrun. */
float slb_host_norm(const float *a_row0, int M, float dPhi, float alpha);

/* display=4 sums over m=1..M-1 (reference: src/boltzmann_solver.c:350-364) */
void slb_host_display4_sums(const float *a_rows01, const float *b_row1, int pitch,
                            float out3[3]);
/* display=77 sums over the CPU solver's flat index 1..2M+1
 * (reference: src/boltzmann_c_solver.c:303-317); a_rows: rows 0..2 of a,
 * b_rows12: rows 1..2 of b */
void slb_host_display77_sums(const float *a_rows012, const float *b_rows12, int pitch,
                             float out3[3]);
void slb_host_multipliers(float k3[3]);

/* the display=4 record written to `out` (reference: src/boltzmann_solver.c:348-380):
 * a01 = rows 0 and 1 of a, b1 = row 1 of b (row pitch `pitch`), av6 = averaging data
 * (rescaled in place), T = period of the a/c field; and the closing of `out` */
void slb_write_display4(float norm, const float *a01, const float *b1, int pitch, float *av6, float T);
/* one display=77 record (reference: src/boltzmann_c_solver.c:298-331): a012 = rows 0..2 of a,
 * b12 = rows 1..2 of b of the lattice BEFORE the step just taken, av6 rescaled in place */
void slb_write_display77(float norm, const float *a012, const float *b12, int pitch, float *av6, float t);
void slb_close_output(void);

/* the display=8 frame file written in pieces (reference: src/boltzmann_solver.c:487-507), for a
 * caller that forms the frame values itself: the phi_x values of the file's blocks (returns
 * their number), the head, one block from the values of columns 1..M+1, the closing norm */
int   slb_frame_phi_x(float *phi, int max);
void *slb_frame_begin(const char *name, float t);
void  slb_frame_block(void *file, float phi_x, const float *vals);
void  slb_frame_end(void *file, const float *a01, int pitch);
/* the same for the display=9 strobe files (reference: src/boltzmann_solver.c:459-484; the
 * running image holds M+1 floats per phi_x block) and the display=3 dump
 * (reference: src/boltzmann_c_solver.c:219-234) */
float *slb_strobe_image(void);
void  slb_strobe_block(void *file, float *image, int p, float phi_x, const float *vals);
void  slb_strobe_end(void *file, const float *a_row0, float t);
void  slb_display3_block(void *file, float phi_x, const float *vals, const float *vals0);
/* read-from=: one round of new parameters, read by the process that owns the input (0 = the
 * run is over) and applied by every process of the run (reference: src/boltzmann_solver.c:384-393) */
int   slb_read_new_parameters(float msg[7]);
void  slb_apply_new_parameters(const float msg[7], float t, float *T, float *t0);

/* ---- several GPUs: one process per phi_y slab (no counterpart in the reference) --------- */
extern int ranks;                      /* ranks= of the command line: processes (GPUs) of the run */

/* links between the processes and the slab time loop: include/slb_slabs.h */
/* the program (every display mode, read-from=) over `nranks` processes, forked here (nothing may have
 * touched a GPU yet); parse_cmd() has run */
int         slb_slab_main(int nranks);

/* the whole program behind main() */
int slb_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
