/*
 * slb_main.c -- the boltzmann_solver program: same command line and the same
 * display=3 / 4 / 7 / 8 / 9 / 77 outputs as the reference, driving the gfx950
 * kernels through the boundary in include/slb_abi.h.
 *
 * Control flow follows the reference host (reference: src/boltzmann_solver.c:74-401);
 * where the reference's CUDA host and its CPU solver disagree (column ranges,
 * which reductions are compounded in place for display=77, the flat-index
 * sums), the CPU solver is followed, because it is the parity target
 * (reference: src/boltzmann_c_solver.c:157-331).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "slb_host.h"

#define HIP_OK(call) HandleError((int)(call), __FILE__, __LINE__)
/* rows of a device array for the host-side sums; a refused request ends the run */
#define GET_ROWS(s, arr, row0, nrows, dst) \
  do { if (slb_solver_get_rows((s), (arr), (row0), (nrows), (dst)) != 0) { \
         fprintf(stderr, "ERROR: rows %d..%d of array %d do not exist (n-harmonics=%d)\n", \
                 (row0), (row0) + (nrows) - 1, (arr), host_N); \
         exit(EXIT_FAILURE); } } while (0)

static const char *HDR4 =
  "#E_{dc}                \\tilde{E}_{\\omega}     \\tilde{\\omega}         mu                     "
  "v_{dr}/v_{p}         A(\\omega)              NORM     v_{y}/v_{p}    m/m_{x,k}   "
  "<v_{dr}/v_{p}>   <v_{y}/v_{p}>    <m/m_{x,k}>    Asin\n";
static const char *HDR77 =
  "#E_{dc}                \\tilde{E}_{\\omega}     \\tilde{\\omega}         mu                     "
  "v_{dr}/v_{p}         A(\\omega)              NORM     v_{y}/v_{p}    m/m_{x,k}   "
  "<v_{dr}/v_{p}>   <v_{y}/v_{p}>    <m/m_{x,k}>  A_{inst}  t    Asin\n";

/* the in-place rescaling both outputs apply to the averaging data
 * (reference: src/boltzmann_c_solver.c:254-260, 319-325) */
static void scale_av(float *av6, const float k3[3], float period)
{
  av6[1] *= k3[0];
  av6[2] *= k3[1];
  av6[3] *= k3[2];
  av6[4] *= k3[0]; av6[4] /= period;
  av6[5] *= k3[0]; av6[5] /= period;
}

/* one display=77 record (reference: src/boltzmann_c_solver.c:298-331) */
void slb_write_display77(float norm, const float *a012, const float *b12, int pitch,
                          float *av6, float t)
{
  printf("\n# t=%0.20f norm=%0.20f\n", t, norm);
  float s3[3], k3[3];
  slb_host_display77_sums(a012, b12, pitch, s3);
  slb_host_multipliers(k3);
  s3[0] *= k3[0]; s3[1] *= k3[1]; s3[2] *= k3[2];
  scale_av(av6, k3, t);
  fputs(HDR77, out);
  fprintf(out, "%0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f\n",
          host_E_dc, host_E_omega, host_omega, host_mu, s3[0], av6[4], norm, s3[1],
          s3[2], av6[1], av6[2], av6[3], cos(host_omega * t) * s3[0], t, av6[4]);
}

/* the display=4 record (reference: src/boltzmann_solver.c:348-380); a01 = rows 0 and 1 of
 * a, b1 = row 1 of b, both with row pitch `pitch`; av6 = the averaging data, rescaled in place */
void slb_write_display4(float norm, const float *a01, const float *b1, int pitch,
                        float *av6, float T)
{
  if (quiet == 0) printf("\n# norm=%0.20f\n", norm);
  float s3[3], k3[3];
  slb_host_display4_sums(a01, b1, pitch, s3);
  slb_host_multipliers(k3);
  s3[0] *= k3[0]; s3[1] *= k3[1]; s3[2] *= k3[2];
  scale_av(av6, k3, T);
  fprintf(out, "# display=%d E_dc=%0.20f E_omega=%0.20f omega=%0.20f mu=%0.20f alpha=%0.20f n-harmonics=%d PhiYmin=%0.20f PhiYmax=%0.20f B=%0.20f t-max=%0.20f dt=%0.20f g-grid=%d\n",
          display, host_E_dc, host_E_omega, host_omega, host_mu, host_alpha, host_N,
          PhiYmin, PhiYmax, host_B, t_start, host_dt, host_M);
  fputs(HDR4, out);
  fprintf(out, "%0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f\n",
          host_E_dc, host_E_omega, host_omega, host_mu, s3[0], av6[4], norm, s3[1],
          s3[2], av6[1], av6[2], av6[3], av6[5]);
}

/* f(phi_x, phi_y) = max(0, sum_n a_n cos(n phi_x) + b_n sin(n phi_x)) on the reference's
 * 629 phi_x values (a float advanced by 0.01 in double, src/boltzmann_solver.c:495) x
 * grid columns: the sums run on the device a batch of phi_x values at a time
 * (slb_solver_frame_values); `line` gets every batch to print.  with_a0 also
 * forms the equilibrium distribution's values (display=3). */
typedef void (*frame_line_fn)(FILE *f, void *ctx, float phi_x, int p_global, const float *vals,
                              const float *vals0, int ncols);

static int for_each_frame_row(slb_solver *s, FILE *f, int with_a0, frame_line_fn line, void *ctx)
{
  const int ncols = host_M + 1;
  int batch = (int)((64u << 20) / (unsigned)ncols);        /* at most 64 Mi values (256 MiB) per batch */
  batch = batch < 8 ? 8 : (batch > 640 ? 640 : batch / 8 * 8);
  float *phis  = (float *)malloc(sizeof(float) * batch);
  float *vals  = (float *)malloc(sizeof(float) * (size_t)batch * ncols);
  float *vals0 = with_a0 ? (float *)malloc(sizeof(float) * (size_t)batch * ncols) : NULL;
  if (!phis || !vals || (with_a0 && !vals0)) {
    fprintf(stderr, "ERROR: out of host memory for the frame\n");
    return -1;
  }
  float phi_x = -SLB_PI;
  int done = 0;
  while (phi_x < SLB_PI) {
    int np = 0;
    for (; np < batch && phi_x < SLB_PI; phi_x += 0.01) phis[np++] = phi_x;
    if (slb_solver_frame_values(s, 0, phis, np, vals) != 0) return -1;
    if (with_a0 && slb_solver_frame_values(s, 1, phis, np, vals0) != 0) return -1;
    for (int p = 0; p < np; p++)
      line(f, ctx, phis[p], done + p, vals + (size_t)p * ncols, with_a0 ? vals0 + (size_t)p * ncols : NULL, ncols);
    done += np;
  }
  free(phis); free(vals); free(vals0);
  return 0;
}

static void line_frame(FILE *f, void *ctx, float phi_x, int p, const float *vals, const float *vals0, int ncols)
{
  (void)ctx; (void)p; (void)vals0;
  for (int m = 1; m < host_M + 2; m++)
    fprintf(f, "%0.5f %0.5f %0.20f\n", phi_x, PhiYmin + host_dPhi * (m - 1), vals[m - 1]);
  (void)ncols;
}

static void line_strobe(FILE *f, void *ctx, float phi_x, int p, const float *vals, const float *vals0, int ncols)
{
  float *strobe = (float *)ctx + (size_t)p * ncols;
  (void)vals0;
  for (int m = 1; m < host_M + 2; m++) {
    strobe[m - 1] = strobe[m - 1] + vals[m - 1];
    fprintf(f, "%0.5f %0.5f %0.20f\n", phi_x, PhiYmin + host_dPhi * (m - 1), strobe[m - 1]);
  }
}

static void line_display3(FILE *f, void *ctx, float phi_x, int p, const float *vals, const float *vals0, int ncols)
{
  (void)ctx; (void)p; (void)ncols;
  for (int m = 1; m < host_M; m++)
    fprintf(f, "%0.5f %0.5f %0.20f %0.20f\n", phi_x, PhiYmin + host_dPhi * (m - 1), vals[m - 1], vals0[m - 1]);
}

/* display=7 / display=8 frame file (reference: src/boltzmann_solver.c:487-507).  The
 * closing norm uses the unpadded (CPU) row pitch for its flat-index sum, so it does
 * not depend on how device rows are padded; a01 = rows 0 and 1 of a. */
static int write_frame8(FILE *f, slb_solver *s, const float *a01, int pitch, float t)
{
  const int cpu_pitch = host_M + 3;
  fprintf(f, "# t=%0.20f\n", t);
  float norm = 0;
  for (int m = 1; m < 2 * host_M + 2; m++) {
    const int r = m >= cpu_pitch ? 1 : 0;
    norm += a01[(size_t)r * pitch + (m - r * cpu_pitch)] * host_dPhi;
  }
  norm *= 2 * SLB_PI * sqrt(host_alpha);
  if (for_each_frame_row(s, f, 0, line_frame, NULL) != 0) return -1;
  fprintf(f, "# norm=%0.20f\n", norm);
  printf("# norm=%0.20f\n", norm);
  return 0;
}

/* The same file written in pieces, for a caller that has the frame values itself (phi_y slabs:
 * each slab forms the values of its columns, rank 0 puts the rows together):
 * slb_frame_phi_x() lists the phi_x values of the file's blocks, slb_frame_begin() opens
 * the file and writes its head, slb_frame_block() one phi_x block from the M+1 values of
 * columns 1..M+1, slb_frame_end() the closing norm (a01 = rows 0 and 1 of a). */
int slb_frame_phi_x(float *phi, int max)
{
  int n = 0;
  for (float phi_x = -SLB_PI; phi_x < SLB_PI; phi_x += 0.01) {
    if (n < max) phi[n] = phi_x;
    n++;
  }
  return n;
}

void *slb_frame_begin(const char *name, float t)
{
  FILE *f = fopen(name, "w");
  if (!f) { perror("ERROR"); return NULL; }
  printf("\nWriting frame %s\n", name);
  fprintf(f, "# t=%0.20f\n", t);
  return f;
}

void slb_frame_block(void *file, float phi_x, const float *vals)
{
  line_frame((FILE *)file, NULL, phi_x, 0, vals, NULL, host_M + 1);
}

void slb_frame_end(void *file, const float *a01, int pitch)
{
  FILE *f = (FILE *)file;
  const int cpu_pitch = host_M + 3;
  float norm = 0;
  for (int m = 1; m < 2 * host_M + 2; m++) {
    const int r = m >= cpu_pitch ? 1 : 0;
    norm += a01[(size_t)r * pitch + (m - r * cpu_pitch)] * host_dPhi;
  }
  norm *= 2 * SLB_PI * sqrt(host_alpha);
  fprintf(f, "# norm=%0.20f\n", norm);
  printf("# norm=%0.20f\n", norm);
  fclose(f);
}

/* The display=9 strobe file and the display=3 dump in pieces, for the same kind of caller:
 * block number p of the strobe file (this period's values added to row p of the running
 * image `image`, M+1 floats per row), the strobe file's closing lines (a_row0 = row 0 of a),
 * and one block of the display=3 dump (vals0 = the equilibrium distribution's values). */
void slb_strobe_block(void *file, float *image, int p, float phi_x, const float *vals)
{
  line_strobe((FILE *)file, image, phi_x, p, vals, NULL, host_M + 1);
}

void slb_strobe_end(void *file, const float *a_row0, float t)
{
  FILE *f = (FILE *)file;
  float norm = 0;
  float dphi_over_2 = host_dPhi / 2.0;
  for (int m = 1; m < host_M + 1; m++) norm += (a_row0[m] + a_row0[m]) * dphi_over_2;
  norm *= 2 * SLB_PI * sqrt(host_alpha);
  fprintf(f, "# norm=%0.20f\n", norm);
  fprintf(f, "# t=%0.20f\n", t);
  printf("# norm=%0.20f\n", norm);
}

void slb_display3_block(void *file, float phi_x, const float *vals, const float *vals0)
{
  line_display3((FILE *)file, NULL, phi_x, 0, vals, vals0, host_M + 1);
}

/* display=9 strobe file: this period's frame added to the running image
 * (reference: src/boltzmann_solver.c:459-484) */
static int write_strobe(FILE *f, slb_solver *s, float *strobe, const float *a0row, float t)
{
  if (for_each_frame_row(s, f, 0, line_strobe, strobe) != 0) return -1;
  slb_strobe_end(f, a0row, t);
  return 0;
}

/* the strobe image of display=9: one row of M+1 values per phi_x block, zeroed */
float *slb_strobe_image(void)
{
  size_t size = 0;
  for (float phi_x = -SLB_PI; phi_x < SLB_PI; phi_x += 0.01) size += (size_t)host_M + 1;
  float *image = (float *)calloc(size, sizeof(float));
  if (!image) fprintf(stderr, "ERROR: out of host memory for the strobe image\n");
  return image;
}

/* read-from=: one round of new parameters (reference: src/boltzmann_solver.c:384-393).
 * slb_read_new_parameters() is for the process that owns the input: it returns 0 when the
 * run is over, else msg = the six host_* values to go on with and the timeout.  What a new
 * value does is the reference host's by default -- load_data() re-reads the host_* values,
 * but bdt = B dt / (4 dPhi) is only ever formed at start-up (:113), so a new B acts through
 * the field term alone.  SLB_REPARAM=fixed forms the derived constants again;
 * SLB_REPARAM=cpu does what the reference's CPU solver does
 * (src/boltzmann_c_solver.c:270-280: nothing reads the new value, only the averaging
 * window moves).  slb_apply_new_parameters() is for every process of the run: t is the loop
 * time the last pass ended at. */
int slb_read_new_parameters(float msg[7])
{
  const char *mode = getenv("SLB_REPARAM");
  const float keep[6] = {host_E_dc, host_E_omega, host_omega, host_mu, host_alpha, host_B};
  const float timeout = scan_for_new_parameters();
  if (timeout < -900) return 0;
  if (mode != NULL && strcmp(mode, "cpu") == 0) {
    host_E_dc = keep[0]; host_E_omega = keep[1]; host_omega = keep[2];
    host_mu = keep[3]; host_alpha = keep[4]; host_B = keep[5];
  }
  msg[0] = host_E_dc; msg[1] = host_E_omega; msg[2] = host_omega;
  msg[3] = host_mu; msg[4] = host_alpha; msg[5] = host_B; msg[6] = timeout;
  return 1;
}

void slb_apply_new_parameters(const float msg[7], float t, float *T, float *t0)
{
  const char *mode = getenv("SLB_REPARAM");
  host_E_dc = msg[0]; host_E_omega = msg[1]; host_omega = msg[2];
  host_mu = msg[3]; host_alpha = msg[4]; host_B = msg[5];
  t_start = t + msg[6];
  t_max = t_start + *T;                         /* the period of the field BEFORE the change, as in the reference */
  *t0 = t + host_dt;
  *T = host_omega > 0 ? (2 * SLB_PI / host_omega) : 0;
  if (mode != NULL && strcmp(mode, "fixed") == 0) slb_derive_and_load();
  else load_data();
}

/* av() beside the time loop.  The sums need three rows of the main lattice a step has
 * written (a rows 0 and 1, b row 1).  Those rows go into a slot of a small ring -- written
 * by the double-step kernel itself for both of its iterations, or copied after a single
 * step -- so that the time loop never has to wait for the sums.  The exact-order sums of one
 * lattice are a chain of dependent passes that keeps three workgroups busy for about as long
 * as a double step keeps the whole device busy; the sums of several lattices side by side
 * take no longer, and the sums of different steps do not depend on each other.  So the rows
 * of up to four double steps wait until the stream of the sums is idle (or eight lattices
 * have gathered) and then go in ONE launch, followed on the same stream by the
 * running-average updates in step order.  A slot is reused only after its update is done.
 * (The tree sums are short: they go out at once.) */
#define AV_SLOTS       32
static struct { float *rows, *tot; double *sums; hipEvent_t ev_upd; int used; } av_slot[AV_SLOTS];
static hipStream_t av_sum_stream;                  /* sums, and the updates behind them */
static hipEvent_t  ev_rows;
static long        av_index = 0;
static int         av_tree = 0;
static slb_slab    av_cols;
#define AV_BATCH 8                               /* lattices per launch of the exact sums */
static struct { int slot; float t; } av_pending[AV_BATCH];
static int         av_n_pending = 0, av_batch_out = 0;
static hipEvent_t  ev_batch;                     /* end of the last launch of the sums + updates */

static void av_ring_init(const slb_solver *s)
{
  const size_t row_bytes = sizeof(float) * (size_t)s->pitch;
  for (int k = 0; k < AV_SLOTS; k++) {
    HIP_OK(hipMalloc((void **)&av_slot[k].rows, 3 * row_bytes));
    HIP_OK(hipMemset(av_slot[k].rows, 0, 3 * row_bytes));
    HIP_OK(hipMalloc((void **)&av_slot[k].tot, 4 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&av_slot[k].sums, 4 * sizeof(double)));
    HIP_OK(hipEventCreateWithFlags(&av_slot[k].ev_upd, hipEventDisableTiming));
    av_slot[k].used = 0;
  }
  HIP_OK(hipStreamCreateWithFlags(&av_sum_stream, hipStreamNonBlocking));
  HIP_OK(hipEventCreateWithFlags(&ev_rows, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&ev_batch, hipEventDisableTiming));
  /* columns 1..M, i.e. m < TMSIZE (reference: src/boltzmann_c_solver.c:417) */
  av_cols.pitch = s->pitch; av_cols.m_offset = 0; av_cols.j_first = 1; av_cols.j_last = host_TMSIZE - 1;
}

/* the slot the av() number av_index + ahead will use; the time loop's stream waits until
 * the slot's previous occupant has been summed and applied */
static void av_ring_flush(slb_solver *s);

static float *av_ring_acquire(slb_solver *s, int ahead)
{
  const int k = (int)((av_index + ahead) % AV_SLOTS);
  for (int i = 0; i < av_n_pending; i++)
    if (av_pending[i].slot == k) { av_ring_flush(s); break; }     /* its rows are still waiting to be summed */
  if (av_slot[k].used) HIP_OK(hipStreamWaitEvent(NULL, av_slot[k].ev_upd, 0));
  return av_slot[k].rows;
}

/* exact sums: everything that is waiting goes to the device in one launch, updates behind it */
static void av_ring_flush(slb_solver *s)
{
  if (av_n_pending == 0) return;
  hipStream_t st = av_sum_stream;
  const float *a[AV_BATCH], *b[AV_BATCH];
  float *tot[AV_BATCH];
  for (int i = 0; i < av_n_pending; i++) {
    const int k = av_pending[i].slot;
    a[i] = av_slot[k].rows; b[i] = a[i] + s->pitch; tot[i] = av_slot[k].tot;
  }
  HIP_OK(hipEventRecord(ev_rows, NULL));
  HIP_OK(hipStreamWaitEvent(st, ev_rows, 0));
  slb_push_stream(st);
  int rc = slb_av_sums_exact_n(&av_cols, av_n_pending, a, b, tot);
  if (rc != 0) { fprintf(stderr, "ERROR: av sums failed (%d)\n", rc); exit(EXIT_FAILURE); }
  for (int i = 0; i < av_n_pending && rc == 0; i++) {
    const int k = av_pending[i].slot;
    rc = slb_av_update_exact(av_slot[k].tot, s->av_data, av_pending[i].t);
    HIP_OK(hipEventRecord(av_slot[k].ev_upd, st));
    av_slot[k].used = 1;
  }
  if (rc != 0) { fprintf(stderr, "ERROR: av update failed (%d)\n", rc); exit(EXIT_FAILURE); }
  HIP_OK(hipEventRecord(ev_batch, st));
  av_batch_out = 1;
  slb_pop_stream();
  av_n_pending = 0;
}

/* the next n (1 to 3) slots' rows have been queued on the time loop's stream: queue their
 * sums (one launch) and their updates of av_data for the loop times t[0..n-1] */
static void av_ring_submit(slb_solver *s, int n, const float *t)
{
  if (!av_tree) {
    if (av_n_pending + n > AV_BATCH) av_ring_flush(s);
    for (int i = 0; i < n; i++) {
      av_pending[av_n_pending].slot = (int)((av_index + i) % AV_SLOTS);
      av_pending[av_n_pending].t = t[i];
      av_n_pending++;
    }
    av_index += n;
    /* out at once when nothing is running; otherwise when a full launch has gathered (three
     * more lattices would not fit) */
    if (av_n_pending + 3 > AV_BATCH || !av_batch_out || hipEventQuery(ev_batch) == hipSuccess) av_ring_flush(s);
    return;
  }
  /* tree sums: short launches that share a workspace; sums and update of each step in turn */
  hipStream_t st = av_sum_stream;
  int rc = 0;
  HIP_OK(hipEventRecord(ev_rows, NULL));
  HIP_OK(hipStreamWaitEvent(st, ev_rows, 0));
  slb_push_stream(st);
  for (int i = 0; i < n && rc == 0; i++) {
    const int k = (int)((av_index + i) % AV_SLOTS);
    rc = slb_av_partial(&av_cols, av_slot[k].rows, av_slot[k].rows + s->pitch, av_slot[k].sums);
    if (rc == 0) rc = slb_av_update(av_slot[k].sums, s->av_data, t[i]);
    HIP_OK(hipEventRecord(av_slot[k].ev_upd, st));
    av_slot[k].used = 1;
  }
  if (rc != 0) { fprintf(stderr, "ERROR: av sums failed (%d)\n", rc); exit(EXIT_FAILURE); }
  slb_pop_stream();
  av_index += n;
}

/* after a single step: copy the three rows of the lattice it wrote, then as above */
static void av_ring_copy_and_submit(slb_solver *s, float t)
{
  const size_t row_bytes = sizeof(float) * (size_t)s->pitch;
  float *rows = av_ring_acquire(s, 0);
  HIP_OK(hipMemcpyAsync(rows, s->a[s->next], 2 * row_bytes, hipMemcpyDeviceToDevice, NULL));
  HIP_OK(hipMemcpyAsync(rows + 2 * (size_t)s->pitch, s->b[s->next] + s->pitch, row_bytes,
                        hipMemcpyDeviceToDevice, NULL));
  av_ring_submit(s, 1, &t);
}

static void av_ring_drain(slb_solver *s)
{
  av_ring_flush(s);
  HIP_OK(hipStreamSynchronize(av_sum_stream));      /* one stream, updates in step order */
  for (int k = 0; k < AV_SLOTS; k++) av_slot[k].used = 0;
}

/* What follows the step taken at loop time t with this frame_time: 0 nothing, 1 only
 * av(), 2 a record, a frame or a strobe test.  Two steps of kind 0 or 1 may share a launch. */
static int step_kind(float t, float frame_time)
{
  if (display == 77) return host_E_omega > 0 && frame_time >= 0.01 ? 2 : 0;
  if (display == 7)  return frame_time >= 0.01 && t > frame_start ? 2 : 0;
  if (display == 8)  return 0;
  if (display == 9 && t >= t_start) return 2;
  return host_E_omega > 0 && t >= t_start ? 1 : 0;     /* av() */
}

void slb_close_output(void)
{
  if (out != NULL && out != stdout && out != stderr) fclose(out);
  out = NULL;
}

int slb_main(int argc, char **argv)
{
  parse_cmd(argc, argv);
  /* ranks=N: N processes, one per GPU, each with a phi_y slab of the lattice (forked
   * before anything here has touched a GPU) */
  if (ranks != 1 || getenv("SLB_FORCE_SLABS") != NULL) return slb_slab_main(ranks);   /* (the variable: the slab loop with one rank, for tests) */
  if (display == 77 && host_N < 2) {
    /* its sums read rows 0-2 of a and 1-2 of b (reference: src/boltzmann_c_solver.c:303-310);
     * the reference reads past the end of its arrays here */
    fprintf(stderr, "ERROR: display=77 needs n-harmonics >= 2.\n");
    return EXIT_FAILURE;
  }
  HIP_OK(hipSetDevice(device));

  float T = host_omega > 0 ? (2 * SLB_PI / host_omega) : 0;   /* period of the a/c field */
  float *strobe = NULL;
  if (display == 9) {                                          /* 101 periods, one strobe image each */
    t_max = t_start + 101 * T;
    if (!(strobe = slb_strobe_image())) return EXIT_FAILURE;
  } else {
    t_max = t_start + T;
  }
  if (quiet == 0) printf("# t_max = %0.20f\n", t_max);

  slb_derive_and_load();
  slb_solver *s = slb_solver_create(0, 1);
  if (!s) return EXIT_FAILURE;
  slb_solver_tiptoe(s);
  av_ring_init(s);

  const int pitch = s->pitch;
  float *a_rows = (float *)malloc(sizeof(float) * 3 * (size_t)pitch);
  float *b_rows = (float *)malloc(sizeof(float) * 3 * (size_t)pitch);
  float av6[6] = {0, 0, 0, 0, 0, 0};

  /* SLB_FORM=sweeps selects the reference's two launches per step, SLB_FORM=fused
   * one launch per step, SLB_FORM=double one launch per two steps; default: one launch per
   * three steps on lattices wide enough for that kernel (slb_solver_steps_per_launch), per two
   * otherwise, wherever the loop allows */
  const char *form = getenv("SLB_FORM");
  const int two_launches = form != NULL && strcmp(form, "sweeps") == 0;
  const int one_launch   = form != NULL && strcmp(form, "fused") == 0;
  const int triples = !(form != NULL && strcmp(form, "double") == 0) && slb_solver_steps_per_launch(s) == 3;

  /* SLB_AV=tree: wave64 tree reduction for av() (about 10 us) instead of the default
   * exact-order sums (bit-identical to the CPU solver; about 1 ms at g-grid 2^20);
   * SLB_AV=chain: the exact sums by one lane adding term after term (cross-check) */
  const char *avm = getenv("SLB_AV");
  av_tree = avm != NULL && strcmp(avm, "tree") == 0;
  slb_set_av_mode(av_tree ? SLB_AV_TREE :
                  avm != NULL && strcmp(avm, "chain") == 0 ? SLB_AV_CHAIN : SLB_AV_EXACT);

  int   progress = 0, frame_number = 1;
  float frame_time = 0, last_tT_reminder = 0;
  char  file_name[64];
  float t0 = 0, t = 0;

  /* SLB_TIMING=1: wall time of every pass of the time loop, on stderr */
  const int timing = getenv("SLB_TIMING") != NULL;

  for (;;) {
    struct timespec w0, w1;
    long loop_steps = 0;
    if (timing) { HIP_OK(hipDeviceSynchronize()); clock_gettime(CLOCK_MONOTONIC, &w0); }
    for (t = t0; t < t_max; t += host_dt, loop_steps++) {
      /* Two iterations in one launch whenever neither of them has display work beyond
       * av() and both lie before t_max: same bits, the lattices in between are just never
       * stored (the rows av() reads of them are). */
      if (!two_launches && !one_launch) {
        const float t2 = t + host_dt, ft2 = frame_time + host_dt;
        const int k1 = step_kind(t, frame_time), k2 = step_kind(t2, ft2);
        /* three iterations in one launch likewise */
        const float t3 = t2 + host_dt, ft3 = ft2 + host_dt;
        const int k3 = step_kind(t3, ft3);
        if (triples && k1 < 2 && k2 < 2 && k3 < 2 && t2 < t_max && t3 < t_max) {
          s->t = t;
          if (k1 == 0 && k2 == 0 && k3 == 0) {
            slb_solver_multi_step(s, 3);
          } else {
            /* av() after every iteration from t_start on: the iterations before it leave their rows
             * in slots nobody reads (the ones the next launches will use) */
            const int kk[3] = {k1, k2, k3};
            const float tk[3] = {t, t2, t3};
            const int n_av = k1 + k2 + k3;
            float *rows[3], tt[3];
            int used = 0;
            for (int i = 0; i < 3; i++) if (kk[i]) { tt[used] = tk[i]; rows[i] = av_ring_acquire(s, used++); }
            for (int i = 0, spare = n_av; i < 3; i++) if (!kk[i]) rows[i] = av_ring_acquire(s, spare++);
            slb_solver_multi_step_rows(s, 3, rows);
            av_ring_submit(s, n_av, tt);
          }
          t = t3;
          loop_steps += 2;
          frame_time = ft3 + host_dt;
          if (out != stdout && display != 7 && quiet == 0 && (progress += 3) >= 300) {
            printf("\rt=%0.9f %0.2f%%", t, t / t_max * 100);
            fflush(stdout);
            progress = 0;
          }
          continue;
        }
        if (k1 < 2 && k2 < 2 && t2 < t_max) {
          s->t = t;                      /* the engine forms both iterations' cosines from its own clock */
          if (k1 == 0 && k2 == 0) {
            slb_solver_double_step(s);
          } else {
            /* av() after both iterations, or only after the second when t_start falls
             * between them (the first iteration's rows then land in a slot nobody reads) */
            float *ra = av_ring_acquire(s, 0), *rb = av_ring_acquire(s, 1);
            slb_solver_double_step_rows(s, k1 ? ra : rb, k1 ? rb : ra);
            const float tt[2] = {k1 ? t : t2, t2};
            av_ring_submit(s, k1 + k2, tt);
          }
          t = t2;
          loop_steps++;
          frame_time = ft2 + host_dt;
          if (out != stdout && display != 7 && quiet == 0 && (progress += 2) >= 300) {
            printf("\rt=%0.9f %0.2f%%", t, t / t_max * 100);
            fflush(stdout);
            progress = 0;
          }
          continue;
        }
      }
      float c[4];
      slb_step_cosines(t, c);
      if (two_launches) {
        slb_solver_sweep_main(s, c[0], c[1]);
        slb_solver_sweep_half(s, c[2], c[3]);
      } else {
        /* both sweeps in one launch; same bits */
        slb_solver_step_fused_range(s, s->half_slab.j_first, s->half_slab.j_last, c);
      }

      if (host_E_omega > 0 && display == 77 && frame_time >= 0.01) {
        av_ring_drain(s);
        av(0, s->a[s->next], s->b[s->next], s->av_data, t);
        GET_ROWS(s, 1 + s->current, 0, 3, a_rows);
        GET_ROWS(s, 5 + s->current, 1, 2, b_rows);
        HIP_OK(hipMemcpy(av6, s->av_data, sizeof(av6), hipMemcpyDeviceToHost));
        const float norm = slb_host_norm(a_rows, host_M, host_dPhi, host_alpha);
        slb_write_display77(norm, a_rows, b_rows, pitch, av6, t);
        /* the CPU solver rescales its averaging data in place at every record */
        HIP_OK(hipMemcpy(s->av_data, av6, sizeof(av6), hipMemcpyHostToDevice));
        frame_time = 0;
      }
      if (host_E_omega > 0 && display != 7 && display != 77 && display != 8 && t >= t_start)
        av_ring_copy_and_submit(s, t);

      slb_solver_swap(s);

      if (display == 9 && t >= t_start) {
        /* a strobe image whenever t/T wraps (reference: src/boltzmann_solver.c:260-275) */
        float tT = t / T;
        float tT_reminder = tT - ((int)tT);
        if (tT_reminder < last_tT_reminder) {
          GET_ROWS(s, 1 + s->current, 0, 1, a_rows);
          sprintf(file_name, "strobe%08d.data", frame_number++);
          FILE *f = fopen(file_name, "w");
          if (!f) { perror("ERROR"); return EXIT_FAILURE; }
          printf("\nWriting strobe %s\n", file_name);
          const int rc = write_strobe(f, s, strobe, a_rows, t);
          fclose(f);
          if (rc != 0) return EXIT_FAILURE;
          frame_time = 0;
        }
        last_tT_reminder = tT_reminder;
      }
      if (display == 7 && frame_time >= 0.01 && t > frame_start) {      /* a movie frame */
        GET_ROWS(s, 1 + s->current, 0, 2, a_rows);
        sprintf(file_name, "frame%08d.data", frame_number++);
        FILE *f = fopen(file_name, "w");
        if (!f) { perror("ERROR"); return EXIT_FAILURE; }
        printf("\nWriting frame %s\n", file_name);
        const int rc = write_frame8(f, s, a_rows, pitch, t);
        fclose(f);
        if (rc != 0) return EXIT_FAILURE;
        frame_time = 0;
      }

      if (out != stdout && display != 7 && quiet == 0) {
        if (++progress >= 300) {
          printf("\rt=%0.9f %0.2f%%", t, t / t_max * 100);
          fflush(stdout);
          progress = 0;
        }
      }
      frame_time += host_dt;
    }

    av_ring_drain(s);
    if (timing) {
      HIP_OK(hipDeviceSynchronize());
      clock_gettime(CLOCK_MONOTONIC, &w1);
      const double sec = (double)(w1.tv_sec - w0.tv_sec) + 1e-9 * (double)(w1.tv_nsec - w0.tv_nsec);
      fprintf(stderr, "# time loop: %ld iterations in %.4f s (%.4f ms each)\n", loop_steps, sec,
              loop_steps ? 1e3 * sec / loop_steps : 0.0);
    }
    GET_ROWS(s, 1 + s->current, 0, 2, a_rows);
    GET_ROWS(s, 5 + s->current, 1, 1, b_rows);
    HIP_OK(hipMemcpy(av6, s->av_data, sizeof(av6), hipMemcpyDeviceToHost));
    const float norm = slb_host_norm(a_rows, host_M, host_dPhi, host_alpha);

    if (display == 3) {
      /* the final distribution next to the equilibrium one
       * (reference: src/boltzmann_c_solver.c:219-234) */
      if (for_each_frame_row(s, out, 1, line_display3, NULL) != 0) return EXIT_FAILURE;
      fprintf(out, "# norm=%0.20f\n", norm);
      printf("# norm=%0.20f\n", norm);
      break;
    }
    if (display == 8) {
      FILE *f = fopen("frame.data", "w");
      if (!f) { perror("ERROR"); return EXIT_FAILURE; }
      printf("\nWriting frame %s\n", "frame.data");
      const int rc = write_frame8(f, s, a_rows, pitch, t);
      fclose(f);
      if (rc != 0) return EXIT_FAILURE;
      break;
    }
    if (display == 4) slb_write_display4(norm, a_rows, b_rows, pitch, av6, T);

    if (read_from == NULL) break;

    /* new parameter from stdin, continue from the present state */
    float msg[7];
    if (!slb_read_new_parameters(msg)) break;
    slb_apply_new_parameters(msg, t, &T, &t0);
    HIP_OK(hipMemset(s->av_data, 0, 6 * sizeof(float)));
    if (quiet == 0) printf("# t_max = %0.20f\n", t_max);
  }

  slb_close_output();
  free(a_rows); free(b_rows);
  slb_solver_destroy(s);
  return EXIT_SUCCESS;
}
