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

#include "slb_host.h"

#define HIP_OK(call) HandleError((int)(call), __FILE__, __LINE__)

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
static void write_frame77(float norm, const float *a012, const float *b12, int pitch,
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

/* the display=4 record (reference: src/boltzmann_solver.c:348-380) */
static void write_display4(float norm, const float *a01, const float *b1, int pitch,
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

/* display=9 strobe file: this period's frame added to the running image
 * (reference: src/boltzmann_solver.c:459-484) */
static int write_strobe(FILE *f, slb_solver *s, float *strobe, const float *a0row, float t)
{
  float norm = 0;
  float dphi_over_2 = host_dPhi / 2.0;
  for (int m = 1; m < host_M + 1; m++) norm += (a0row[m] + a0row[m]) * dphi_over_2;
  norm *= 2 * SLB_PI * sqrt(host_alpha);
  if (for_each_frame_row(s, f, 0, line_strobe, strobe) != 0) return -1;
  fprintf(f, "# norm=%0.20f\n", norm);
  fprintf(f, "# t=%0.20f\n", t);
  printf("# norm=%0.20f\n", norm);
  return 0;
}

/* av() beside the time loop.  The sums read the main lattice a step has just written;
 * the following step only reads that lattice too, and the one after it is the first to
 * write those arrays again.  So av() goes to a stream of its own, ordered after its step
 * by an event, and runs while the next step is computed; a step waits for the av() issued
 * two steps before it.  (The exact-order sums keep three compute units busy for about as
 * long as a step keeps all of them busy; sharing the machine with a step slows them, so
 * at g-grid 2^20 a step with av() takes about 1.6 ms instead of 2.0, not yet 1.0.) */
static hipStream_t av_stream = NULL;
static hipEvent_t  ev_step, ev_av[2];
static int         av_issued[2] = {0, 0};
static long        av_index = 0;

static void av_overlap_init(void)
{
  /* highest priority: three workgroups that should not queue behind a step's thousands */
  int prio_low = 0, prio_high = 0;
  HIP_OK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  HIP_OK(hipStreamCreateWithPriority(&av_stream, hipStreamNonBlocking, prio_high));
  HIP_OK(hipEventCreateWithFlags(&ev_step, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&ev_av[0], hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&ev_av[1], hipEventDisableTiming));
}

/* before a step is launched: the av() that read the arrays this step overwrites */
static void av_overlap_before_step(int all)
{
  const int slot = (int)(av_index & 1);
  if (av_issued[slot]) HIP_OK(hipStreamWaitEvent(NULL, ev_av[slot], 0));
  if (all && av_issued[slot ^ 1]) HIP_OK(hipStreamWaitEvent(NULL, ev_av[slot ^ 1], 0));
}

static void av_overlap_issue(slb_solver *s, float t)
{
  const int slot = (int)(av_index & 1);
  HIP_OK(hipEventRecord(ev_step, NULL));
  HIP_OK(hipStreamWaitEvent(av_stream, ev_step, 0));
  slb_set_stream(av_stream);
  av(0, s->a[s->next], s->b[s->next], s->av_data, t);
  slb_set_stream(NULL);
  HIP_OK(hipEventRecord(ev_av[slot], av_stream));
  av_issued[slot] = 1;
  av_index++;
}

static void av_overlap_drain(void)
{
  HIP_OK(hipStreamSynchronize(av_stream));
  av_issued[0] = av_issued[1] = 0;
}

/* is there display work (averaging, a record, a frame) right after the step taken at
 * loop time t with this frame_time?  Steps without any may be taken two at a time. */
static int step_has_work(float t, float frame_time)
{
  if (display == 77) return host_E_omega > 0 && frame_time >= 0.01;
  if (display == 7)  return frame_time >= 0.01 && t > frame_start;
  if (display == 8)  return 0;
  if (display == 9 && t >= t_start) return 1;
  return host_E_omega > 0 && t >= t_start;             /* av() */
}

int slb_main(int argc, char **argv)
{
  parse_cmd(argc, argv);
  HIP_OK(hipSetDevice(device));

  float T = host_omega > 0 ? (2 * SLB_PI / host_omega) : 0;   /* period of the a/c field */
  float *strobe = NULL;
  if (display == 9) {                                          /* 101 periods, one strobe image each */
    t_max = t_start + 101 * T;
    size_t size = 0;
    for (float phi_x = -SLB_PI; phi_x < SLB_PI; phi_x += 0.01) size += (size_t)host_M + 1;
    strobe = (float *)calloc(size, sizeof(float));
    if (!strobe) { fprintf(stderr, "ERROR: out of host memory for the strobe image\n"); return EXIT_FAILURE; }
  } else {
    t_max = t_start + T;
  }
  if (quiet == 0) printf("# t_max = %0.20f\n", t_max);

  slb_derive_and_load();
  slb_solver *s = slb_solver_create(0, 1);
  if (!s) return EXIT_FAILURE;
  slb_solver_tiptoe(s);
  av_overlap_init();

  const int pitch = s->pitch;
  float *a_rows = (float *)malloc(sizeof(float) * 3 * (size_t)pitch);
  float *b_rows = (float *)malloc(sizeof(float) * 3 * (size_t)pitch);
  float av6[6] = {0, 0, 0, 0, 0, 0};

  /* SLB_FORM=sweeps selects the reference's two launches per step, SLB_FORM=fused
   * one launch per step; default: one launch per two steps where the loop allows */
  const char *form = getenv("SLB_FORM");
  const int two_launches = form != NULL && strcmp(form, "sweeps") == 0;
  const int one_launch   = form != NULL && strcmp(form, "fused") == 0;

  /* SLB_AV=tree: wave64 tree reduction for av() (about 10 us) instead of the default
   * exact-order sums (bit-identical to the CPU solver; about 1 ms at g-grid 2^20);
   * SLB_AV=chain: the exact sums by one lane adding term after term (cross-check) */
  const char *avm = getenv("SLB_AV");
  slb_set_av_mode(avm != NULL && strcmp(avm, "tree") == 0 ? SLB_AV_TREE :
                  avm != NULL && strcmp(avm, "chain") == 0 ? SLB_AV_CHAIN : SLB_AV_EXACT);

  int   progress = 0, frame_number = 1;
  float frame_time = 0, last_tT_reminder = 0;
  char  file_name[64];
  float t0 = 0, t = 0;

  for (;;) {
    for (t = t0; t < t_max; t += host_dt) {
      /* Two iterations in one launch whenever neither of them has display work
       * (averaging or a display=77 record) and both lie before t_max: same bits,
       * the lattices in between are just never stored. */
      if (!two_launches && !one_launch) {
        const float t2 = t + host_dt, ft2 = frame_time + host_dt;
        if (!step_has_work(t, frame_time) && !step_has_work(t2, ft2) && t2 < t_max) {
          s->t = t;                      /* the engine forms both iterations' cosines from its own clock */
          av_overlap_before_step(1);
          slb_solver_double_step(s);
          t = t2;
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
      av_overlap_before_step(0);
      if (two_launches) {
        slb_solver_sweep_main(s, c[0], c[1]);
        slb_solver_sweep_half(s, c[2], c[3]);
      } else {
        /* both sweeps in one launch; same bits */
        slb_solver_step_fused_range(s, s->half_slab.j_first, s->half_slab.j_last, c);
      }

      if (host_E_omega > 0 && display == 77 && frame_time >= 0.01) {
        av_overlap_drain();
        av(0, s->a[s->next], s->b[s->next], s->av_data, t);
        slb_solver_get_rows(s, 1 + s->current, 0, 3, a_rows);
        slb_solver_get_rows(s, 5 + s->current, 1, 2, b_rows);
        HIP_OK(hipMemcpy(av6, s->av_data, sizeof(av6), hipMemcpyDeviceToHost));
        const float norm = slb_host_norm(a_rows, host_M, host_dPhi, host_alpha);
        write_frame77(norm, a_rows, b_rows, pitch, av6, t);
        /* the CPU solver rescales its averaging data in place at every record */
        HIP_OK(hipMemcpy(s->av_data, av6, sizeof(av6), hipMemcpyHostToDevice));
        frame_time = 0;
      }
      if (host_E_omega > 0 && display != 7 && display != 77 && display != 8 && t >= t_start)
        av_overlap_issue(s, t);

      slb_solver_swap(s);

      if (display == 9 && t >= t_start) {
        /* a strobe image whenever t/T wraps (reference: src/boltzmann_solver.c:260-275) */
        float tT = t / T;
        float tT_reminder = tT - ((int)tT);
        if (tT_reminder < last_tT_reminder) {
          slb_solver_get_rows(s, 1 + s->current, 0, 1, a_rows);
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
        slb_solver_get_rows(s, 1 + s->current, 0, 2, a_rows);
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

    av_overlap_drain();
    slb_solver_get_rows(s, 1 + s->current, 0, 2, a_rows);
    slb_solver_get_rows(s, 5 + s->current, 1, 1, b_rows);
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
    if (display == 4) write_display4(norm, a_rows, b_rows, pitch, av6, T);

    if (read_from == NULL) break;

    /* new parameter from stdin, continue from the present state
     * (reference: src/boltzmann_solver.c:384-393) */
    const float timeout = scan_for_new_parameters();
    if (timeout < -900) break;
    t_start = t + timeout;
    t_max = t_start + T;
    t0 = t + host_dt;
    T = host_omega > 0 ? (2 * SLB_PI / host_omega) : 0;
    /* B enters the sweeps through bdt, so the derived constants are refreshed
     * as well (the reference re-uploads a stale bdt here) */
    slb_derive_and_load();
    HIP_OK(hipMemset(s->av_data, 0, 6 * sizeof(float)));
    if (quiet == 0) printf("# t_max = %0.20f\n", t_max);
  }

  if (out != NULL && out != stdout && out != stderr) fclose(out);
  free(a_rows); free(b_rows);
  slb_solver_destroy(s);
  return EXIT_SUCCESS;
}
