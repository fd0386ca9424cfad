/*
 * slb_oracle.c -- CPU oracle for the super-lattice Boltzmann stepping loop.
 *
 * TEST INFRASTRUCTURE ONLY (see slb_oracle.h for the rules and for the pinning
 * status: PARITY UNPINNED beyond one recorded display=4 line of the reference).
 *
 * Everything is evaluated in IEEE binary32 with one rounding per operation and
 * the association order the reference source spells out; build without
 * -march=native / -mfma / -ffast-math so no product-sum is contracted.
 */
#include "slb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The reference's PI literal (src/constants.h:11) is the double nearest pi. */
#define ORC_PI 3.141592653589793115998

static int orc_threads = 1;

void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }
int  orc_get_threads(void)  { return orc_threads; }

/* reference: src/boltzmann_c_solver.c:100-113 */
void orc_derive(orc_params *p)
{
  p->dPhi     = (p->PhiYmax - p->PhiYmin) / p->M;
  p->MSIZE    = p->M + 3;
  p->TMSIZE   = p->M + 1;
  p->nu       = 1 + p->dt / 2;
  p->nu2      = p->nu * p->nu;
  p->nu_tilde = 1 - p->dt / 2;
  p->bdt      = p->B * p->dt / (4 * p->dPhi);
}

/* I_n(x) = sum_k (x/2)^(2k+n) / (k! (n+k)!), summed in long double.
 * GSL (unpinned version, not in this image) is what the reference calls; any
 * routine accurate to a few 1e-16 gives the same float a0 except on rounding
 * ties.  Orders far above x underflow to 0, which is also the float result. */
double orc_bessel_In(int n, double x)
{
  if (n < 0) n = -n;
  long double h = (long double)x / 2.0L;
  long double term = 1.0L;
  for (int i = 1; i <= n; i++) term *= h / (long double)i;   /* h^n / n! */
  long double sum = term;
  long double h2 = h * h;
  for (int k = 1; k < 100000; k++) {
    term *= h2 / ((long double)k * (long double)(n + k));
    sum += term;
    if (term <= 1e-22L * sum) break;
  }
  return (double)sum;
}

/* grid column -> phi_y, reference macro src/boltzmann_c_solver.c:69 */
static inline float orc_phi_y(const orc_params *p, int m)
{
  return p->PhiYmin + p->dPhi * (m - 1);
}

/* reference: src/boltzmann_c_solver.c:116-122.  The Gaussian factor depends on
 * m only, so it is evaluated once per column; the product the reference forms,
 * float a times long double expl(...), narrowed to float, is unchanged. */
void orc_init_a0(const orc_params *p, float *a0)
{
  const int cols = p->M + 3;
  long double *gauss = (long double *)malloc(sizeof(long double) * (size_t)cols);
  for (int m = 0; m < cols; m++)
    gauss[m] = expl(-p->mu * pow(orc_phi_y(p, m), 2) / 2);
  const double i0 = orc_bessel_In(0, p->mu);
  for (int n = 0; n <= p->N; n++) {
    float c = orc_bessel_In(n, p->mu) * (n == 0 ? 0.5 : 1) / (ORC_PI * i0)
              * sqrt(p->mu / (2 * ORC_PI * p->alpha));
    float *row = a0 + (size_t)n * p->MSIZE;
    for (int m = 0; m < cols; m++) row[m] = c * gauss[m];
  }
  free(gauss);
}

/* The point update shared by both sweeps.
 * reference: src/boltzmann_c_solver.c:363-378 (== :393-406). */
void orc_sweep(const orc_params *p, int pitch, int m_offset, int j_first, int j_last,
               const float *a0, const float *a_c, const float *b_c,
               const float *a_s, const float *b_s, float *a_out, float *b_out,
               float cos_omega_t, float cos_omega_t_plus_dt)
{
  const int   N = p->N;
  const float E_dc = p->E_dc, E_omega = p->E_omega, B = p->B, dt = p->dt;
  const float nu = p->nu, nu2 = p->nu2, nu_tilde = p->nu_tilde, bdt = p->bdt;
  const size_t P = (size_t)pitch;

  #pragma omp parallel for num_threads(orc_threads) schedule(static)
  for (int j = j_first; j <= j_last; j++) {
    const float phi = orc_phi_y(p, j + m_offset);
    const float mu_t_part        = (E_dc + E_omega * cos_omega_t         + B * phi) * dt / 2;
    const float mu_t_plus_1_part = (E_dc + E_omega * cos_omega_t_plus_dt + B * phi) * dt / 2;

    for (int n = 0; n < N; n++) {
      const size_t c  = (size_t)n * P + (size_t)j;     /* (n, j)   */
      const size_t up = c + P;                          /* (n+1, j) */
      const size_t dn = c - P;                          /* (n-1, j), only used when n >= 1 */
      const float mu_t        = n * mu_t_part;
      const float mu_t_plus_1 = n * mu_t_plus_1_part;

      float g = dt * a0[c] + a_c[c] * nu_tilde - b_c[c] * mu_t
              + bdt * (b_s[up + 1] - b_s[up - 1]
                       - (n < 2 ? 0 : (b_s[dn + 1] - b_s[dn - 1])));
      float h = b_c[c] * nu_tilde + a_c[c] * mu_t
              + bdt * ((n == 1 ? 2 : 1) * (n == 0 ? 0 : (a_s[dn + 1] - a_s[dn - 1]))
                       - a_s[up + 1] + a_s[up - 1]);

      float xi = nu2 + mu_t_plus_1 * mu_t_plus_1;
      a_out[c] = (g * nu - h * mu_t_plus_1) / xi;
      if (n > 0)
        b_out[c] = (g * mu_t_plus_1 + h * nu) / xi;
    }
  }
}

void orc_step_on_grid(const orc_params *p, const float *a0,
                      const float *a_current, const float *b_current,
                      float *a_next, float *b_next,
                      const float *a_current_hs, const float *b_current_hs,
                      float cos_omega_t, float cos_omega_t_plus_dt)
{
  /* columns 1..TMSIZE inclusive (reference :361) */
  orc_sweep(p, p->MSIZE, 0, 1, p->TMSIZE, a0, a_current, b_current,
            a_current_hs, b_current_hs, a_next, b_next,
            cos_omega_t, cos_omega_t_plus_dt);
}

void orc_step_on_half_grid(const orc_params *p, const float *a0,
                           const float *a_next, const float *b_next,
                           const float *a_current_hs, const float *b_current_hs,
                           float *a_next_hs, float *b_next_hs,
                           float cos_omega_t, float cos_omega_t_plus_dt)
{
  /* columns 1..TMSIZE-1 (reference :391); centre = hs lattice, stencil = next */
  orc_sweep(p, p->MSIZE, 0, 1, p->TMSIZE - 1, a0, a_current_hs, b_current_hs,
            a_next, b_next, a_next_hs, b_next_hs,
            cos_omega_t, cos_omega_t_plus_dt);
}

/* reference: src/boltzmann_c_solver.c:413-437 */
void orc_av(const orc_params *p, const float *a, const float *b, float *av_data, float t)
{
  const size_t P = (size_t)p->MSIZE;
  int av_count = av_data[0] + 1;

  float v_dr = 0, v_y = 0, mmx = 0;
  for (int m = 1; m < p->TMSIZE; m++) {
    v_dr += b[P + m] * p->dPhi;
    v_y  += a[m] * orc_phi_y(p, m) * p->dPhi;
    mmx  += a[P + m] * p->dPhi;
  }
  av_data[1] += (v_dr - av_data[1]) / av_count;
  av_data[2] += (v_y  - av_data[2]) / av_count;
  av_data[3] += (mmx  - av_data[3]) / av_count;
  av_data[4] += cos(p->omega * t) * v_dr * p->dt;
  av_data[5] += sin(p->omega * t) * v_dr * p->dt;
  av_data[0] += 1;
}

float orc_eval_norm(const orc_params *p, const float *a)
{
  float norm = 0;
  for (int m = 1; m < p->M + 1; m++) norm += a[m] * p->dPhi;
  norm *= 2 * ORC_PI * sqrt(p->alpha);
  return norm;
}

int orc_state_init(orc_state *s, const orc_params *p_in)
{
  memset(s, 0, sizeof(*s));
  s->p = *p_in;
  orc_derive(&s->p);
  const orc_params *p = &s->p;
  const size_t cells = (size_t)(p->N + 1) * (size_t)p->MSIZE;

  s->a0 = (float *)calloc(cells, sizeof(float));
  if (!s->a0) return -1;
  for (int i = 0; i < 4; i++) {
    s->a[i] = (float *)calloc(cells, sizeof(float));
    s->b[i] = (float *)calloc(cells, sizeof(float));
    if (!s->a[i] || !s->b[i]) return -1;
  }
  orc_init_a0(p, s->a0);
  s->current = 0; s->next = 1; s->current_hs = 2; s->next_hs = 3;
  /* size_t byte count: the reference's int count overflows past 2 GiB per
   * array (src/boltzmann_c_solver.c:106, :136) and then skips this copy. */
  memcpy(s->a[0], s->a0, cells * sizeof(float));

  /* tiptoe to the first half step (reference :141-145): centre and stencil are
   * both the t=0 lattice, the result lands in the hs lattice. */
  float c0 = 1;
  float c1 = cos(p->omega * (p->dt));
  orc_step_on_grid(p, s->a0, s->a[0], s->b[0], s->a[2], s->b[2], s->a[0], s->b[0], c0, c1);

  s->T     = p->omega > 0 ? (2 * ORC_PI / p->omega) : 0;
  s->t_max = p->t_start + s->T;
  s->t0 = 0; s->t = 0; s->frame_time = 0; s->steps_done = 0;
  return 0;
}

void orc_state_free(orc_state *s)
{
  free(s->a0);
  for (int i = 0; i < 4; i++) { free(s->a[i]); free(s->b[i]); }
  memset(s, 0, sizeof(*s));
}

static void orc_swap(orc_state *s)
{
  if (s->current == 0)    { s->current = 1;    s->next = 0; }    else { s->current = 0;    s->next = 1; }
  if (s->current_hs == 2) { s->current_hs = 3; s->next_hs = 2; } else { s->current_hs = 2; s->next_hs = 3; }
}

/* the two sweeps of one loop iteration at time s->t (reference :165-176) */
static void orc_two_sweeps(orc_state *s)
{
  const orc_params *p = &s->p;
  const float t = s->t, dt = p->dt, omega = p->omega;
  float t_hs = t + dt / 2;
  float c0 = cos(omega * t);
  float c1 = cos(omega * (t + dt));
  orc_step_on_grid(p, s->a0, s->a[s->current], s->b[s->current], s->a[s->next], s->b[s->next],
                   s->a[s->current_hs], s->b[s->current_hs], c0, c1);
  c0 = cos(omega * t_hs);
  c1 = cos(omega * (t_hs + dt));
  orc_step_on_half_grid(p, s->a0, s->a[s->next], s->b[s->next],
                        s->a[s->current_hs], s->b[s->current_hs],
                        s->a[s->next_hs], s->b[s->next_hs], c0, c1);
}

void orc_state_step(orc_state *s)
{
  orc_two_sweeps(s);
  orc_swap(s);
  s->t += s->p.dt;
  s->frame_time += s->p.dt;
  s->steps_done++;
}

static const char *orc_hdr4 =
  "#E_{dc}                \\tilde{E}_{\\omega}     \\tilde{\\omega}         mu                     "
  "v_{dr}/v_{p}         A(\\omega)              NORM     v_{y}/v_{p}    m/m_{x,k}   "
  "<v_{dr}/v_{p}>   <v_{y}/v_{p}>    <m/m_{x,k}>    Asin\n";
static const char *orc_hdr77 =
  "#E_{dc}                \\tilde{E}_{\\omega}     \\tilde{\\omega}         mu                     "
  "v_{dr}/v_{p}         A(\\omega)              NORM     v_{y}/v_{p}    m/m_{x,k}   "
  "<v_{dr}/v_{p}>   <v_{y}/v_{p}>    <m/m_{x,k}>  A_{inst}  t    Asin\n";

/* reference: src/boltzmann_c_solver.c:298-331.  The sums deliberately run over
 * the flat index 1..2M+1 with pitch M+3 exactly as the reference does, i.e.
 * they spill into the following row; av_data is rescaled in place. */
static void orc_print_time_evolution(orc_state *s, FILE *out, FILE *log,
                                     float norm, const float *a, const float *b, float t)
{
  const orc_params *p = &s->p;
  const size_t P = (size_t)p->MSIZE;
  float *av_data = s->av_data;
  fprintf(log, "\n# t=%0.20f norm=%0.20f\n", t, norm);
  float v_dr = 0, v_y = 0, mmx = 0;
  for (int m = 1; m < 2 * p->M + 2; m++) {
    v_dr += b[P + m] * p->dPhi;
    v_y  += a[m] * orc_phi_y(p, m) * p->dPhi;
    mmx  += a[P + m] * p->dPhi;
  }
  const double i0 = orc_bessel_In(0, p->mu), i1 = orc_bessel_In(1, p->mu);
  float k_dr = 2 * i0 * ORC_PI * sqrt(p->alpha) / i1;
  float k_y  = 4 * ORC_PI * i0 / i1;
  float k_m  = ORC_PI * p->alpha * sqrt(p->alpha);
  v_dr *= k_dr; v_y *= k_y; mmx *= k_m;

  av_data[1] *= k_dr;
  av_data[2] *= k_y;
  av_data[3] *= k_m;
  av_data[4] *= k_dr; av_data[4] /= t;
  av_data[5] *= k_dr; av_data[5] /= t;

  fputs(orc_hdr77, out);
  fprintf(out, "%0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f\n",
          p->E_dc, p->E_omega, p->omega, p->mu, v_dr, av_data[4], norm, v_y,
          mmx, av_data[1], av_data[2], av_data[3], cos(p->omega * t) * v_dr, t, av_data[4]);
}

/* reference: src/boltzmann_c_solver.c:236-268 */
static void orc_print_display4(orc_state *s, int display, FILE *out, FILE *log, float norm)
{
  const orc_params *p = &s->p;
  const size_t P = (size_t)p->MSIZE;
  const float *a = s->a[s->current], *b = s->b[s->current];
  float *av_data = s->av_data;
  fprintf(log, "\n# norm=%0.20f\n", norm);
  float v_dr = 0, v_y = 0, mmx = 0;
  for (int m = 1; m < p->M; m++) {
    v_dr += b[P + m] * p->dPhi;
    v_y  += a[m] * orc_phi_y(p, m) * p->dPhi;
    mmx  += a[P + m] * p->dPhi;
  }
  const double i0 = orc_bessel_In(0, p->mu), i1 = orc_bessel_In(1, p->mu);
  float k_dr = 2 * i0 * ORC_PI * sqrt(p->alpha) / i1;
  float k_y  = 4 * ORC_PI * i0 / i1;
  float k_m  = ORC_PI * p->alpha * sqrt(p->alpha);
  v_dr *= k_dr; v_y *= k_y; mmx *= k_m;

  av_data[1] *= k_dr;
  av_data[2] *= k_y;
  av_data[3] *= k_m;
  av_data[4] *= k_dr; av_data[4] /= s->T;
  av_data[5] *= k_dr; av_data[5] /= s->T;

  fprintf(out, "# display=%d E_dc=%0.20f E_omega=%0.20f omega=%0.20f mu=%0.20f alpha=%0.20f n-harmonics=%d PhiYmin=%0.20f PhiYmax=%0.20f B=%0.20f t-max=%0.20f dt=%0.20f g-grid=%d\n",
          display, p->E_dc, p->E_omega, p->omega, p->mu, p->alpha, p->N,
          p->PhiYmin, p->PhiYmax, p->B, p->t_start, p->dt, p->M);
  fputs(orc_hdr4, out);
  fprintf(out, "%0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f %0.20f\n",
          p->E_dc, p->E_omega, p->omega, p->mu, v_dr, av_data[4], norm, v_y,
          mmx, av_data[1], av_data[2], av_data[3], av_data[5]);
}

static float orc_frame_start = 0;      /* frame-start= of the command line (display=7) */
void orc_set_frame_start(float t) { orc_frame_start = t; }

static float frame_value(const orc_params *p, const float *a, const float *b, int m, float phi_x);

/* display=9: add this period's frame to the running strobe image and write it
 * (reference: src/boltzmann_solver.c:459-484; the norm there is the trapezoid-looking
 * sum over columns 1..M) */
static void orc_print_2d_strobe(const orc_params *p, FILE *out, FILE *log, float *strobe,
                                const float *a, const float *b, float t)
{
  float norm = 0;
  float dphi_over_2 = p->dPhi / 2.0;
  for (int m = 1; m < p->M + 1; m++) norm += (a[m] + a[m]) * dphi_over_2;
  norm *= 2 * ORC_PI * sqrt(p->alpha);
  size_t i = 0;
  for (float phi_x = -ORC_PI; phi_x < ORC_PI; phi_x += 0.01) {
    for (int m = 1; m < p->M + 2; m++) {
      strobe[i] = strobe[i] + frame_value(p, a, b, m, phi_x);
      fprintf(out, "%0.5f %0.5f %0.20f\n", phi_x, orc_phi_y(p, m), strobe[i]);
      i++;
    }
  }
  fprintf(out, "# norm=%0.20f\n", norm);
  fprintf(out, "# t=%0.20f\n", t);
  fprintf(log, "# norm=%0.20f\n", norm);
}

/* display=3: the distribution next to the equilibrium one, columns 1..M-1
 * (reference: src/boltzmann_c_solver.c:219-234) */
static void orc_print_display3(const orc_params *p, FILE *out, FILE *log, float norm,
                               const float *a0, const float *a, const float *b)
{
  const size_t P = (size_t)p->MSIZE;
  for (float phi_x = -ORC_PI; phi_x < ORC_PI; phi_x += 0.01) {
    for (int m = 1; m < p->M; m++) {
      float value = 0, value0 = 0;
      for (int n = 0; n < p->N + 1; n++) {
        value  += a[n * P + m] * cos(n * phi_x) + b[n * P + m] * sin(n * phi_x);
        value0 += a0[n * P + m] * cos(n * phi_x);
      }
      fprintf(out, "%0.5f %0.5f %0.20f %0.20f\n", phi_x, orc_phi_y(p, m),
              value < 0 ? 0 : value, value0 < 0 ? 0 : value0);
    }
  }
  fprintf(out, "# norm=%0.20f\n", norm);
  fprintf(log, "# norm=%0.20f\n", norm);
}

/* reference: src/boltzmann_c_solver.c:157-268 (one pass of the outer loop, no
 * stdin re-parameterisation).  display=7 and display=9 follow the CUDA host
 * (src/boltzmann_solver.c:255-287), the only place that has the strobe and the
 * frame-start test; its frame writer is the one that stays inside the arrays. */
int orc_solver_run(orc_state *s, int display, FILE *out, FILE *log, long max_steps)
{
  const orc_params *p = &s->p;
  int progress = 0, frame_number = 1;
  float last_tT_reminder = 0;
  float *strobe = NULL;
  char name[64];
  if (display == 9) {
    s->t_max = p->t_start + 101 * s->T;
    size_t size = 0;
    for (float phi_x = -ORC_PI; phi_x < ORC_PI; phi_x += 0.01) size += (size_t)p->M + 1;
    strobe = (float *)calloc(size, sizeof(float));
    if (!strobe) return -1;
  }
  fprintf(log, "# t_max = %0.20f\n", s->t_max);

  for (s->t = s->t0; s->t < s->t_max; s->t += p->dt) {
    if (max_steps >= 0 && s->steps_done >= max_steps) break;
    orc_two_sweeps(s);

    if (p->E_omega > 0 && display == 77 && s->frame_time >= 0.01) {
      orc_av(p, s->a[s->next], s->b[s->next], s->av_data, s->t);
      float norm = orc_eval_norm(p, s->a[s->current]);
      orc_print_time_evolution(s, out, log, norm, s->a[s->current], s->b[s->current], s->t);
      s->frame_time = 0;
    }
    if (p->E_omega > 0 && display != 7 && display != 77 && s->t >= p->t_start)
      orc_av(p, s->a[s->next], s->b[s->next], s->av_data, s->t);

    orc_swap(s);

    if (display == 9 && s->t >= p->t_start) {
      float tT = s->t / s->T;
      float tT_reminder = tT - ((int)tT);
      if (tT_reminder < last_tT_reminder) {
        sprintf(name, "strobe%08d.data", frame_number++);
        FILE *f = fopen(name, "w");
        if (!f) return -1;
        fprintf(log, "\nWriting strobe %s\n", name);
        orc_print_2d_strobe(p, f, log, strobe, s->a[s->current], s->b[s->current], s->t);
        fclose(f);
        s->frame_time = 0;
      }
      last_tT_reminder = tT_reminder;
    }
    if (display == 7 && s->frame_time >= 0.01 && s->t > orc_frame_start) {
      sprintf(name, "frame%08d.data", frame_number++);
      FILE *f = fopen(name, "w");
      if (!f) return -1;
      fprintf(log, "\nWriting frame %s\n", name);
      orc_print_2d_data(p, f, log, s->a[s->current], s->b[s->current], s->t);
      fclose(f);
      s->frame_time = 0;
    }

    if (out != stdout && display != 7) {
      if (++progress == 300) {
        fprintf(log, "\rt=%0.9f %0.2f%%", s->t, s->t / s->t_max * 100);
        fflush(log);
        progress = 0;
      }
    }
    s->frame_time += p->dt;
    s->steps_done++;
  }

  free(strobe);
  float norm = orc_eval_norm(p, s->a[s->current]);
  if (display == 3) orc_print_display3(p, out, log, norm, s->a0, s->a[s->current], s->b[s->current]);
  if (display == 4) orc_print_display4(s, display, out, log, norm);
  if (display == 8) {
    FILE *f = fopen("frame.data", "w");
    if (!f) return -1;
    fprintf(log, "\nWriting frame %s\n", "frame.data");
    orc_print_2d_data(p, f, log, s->a[s->current], s->b[s->current], s->t);
    fclose(f);
  }
  return 0;
}

/* read-from=stdin: one of E_dc, E_omega, omega, mu, alpha, B is replaced and the run goes
 * on from the present state for `timeout` more time units plus one period.  The two
 * reference programs disagree on what "replaced" means, so there are three forms, chosen by
 * the environment variable SLB_REPARAM (the same switch the GPU program reads):
 *   cuda  (default) the CUDA host, reference src/boltzmann_solver.c:384-393: the new value
 *         reaches the sweeps through load_data(), but the derived constant bdt = B dt / (4 dPhi)
 *         is NOT formed again (host_bdt is only computed at start-up, :113), so a new B acts
 *         through the field term only;
 *   cpu   the CPU solver, reference src/boltzmann_c_solver.c:270-280: scan_for_new_parameters()
 *         sets the host_* copies and nobody reads them again (no load_data()), so the new
 *         value is ignored; only the averaging window moves;
 *   fixed what the feature is for: the new value and every constant derived from it.
 * In all three t_max is formed with the OLD period, the averaging data are cleared, and a0
 * (which depends on mu and alpha) stays as it was built at start-up.
 * Returns 0, or -1 for an unknown name. */
int orc_solver_continue(orc_state *s, const char *name, float value, float timeout)
{
  orc_params *p = &s->p;
  const char *mode = getenv("SLB_REPARAM");
  const int cpu = mode && !strcmp(mode, "cpu"), fixed = mode && !strcmp(mode, "fixed");
  int known = 1;
  float ignored;
  if      (!strcmp(name, "E_dc"))    *(cpu ? &ignored : &p->E_dc) = value;
  else if (!strcmp(name, "E_omega")) *(cpu ? &ignored : &p->E_omega) = value;
  else if (!strcmp(name, "omega"))   *(cpu ? &ignored : &p->omega) = value;
  else if (!strcmp(name, "mu"))      *(cpu ? &ignored : &p->mu) = value;
  else if (!strcmp(name, "alpha"))   *(cpu ? &ignored : &p->alpha) = value;
  else if (!strcmp(name, "B"))       *(cpu ? &ignored : &p->B) = value;
  else known = 0;
  p->t_start = s->t + timeout;
  s->t_max = p->t_start + s->T;
  s->t0 = s->t + p->dt;
  s->T = p->omega > 0 ? (2 * ORC_PI / p->omega) : 0;
  if (fixed) orc_derive(p);
  memset(s->av_data, 0, sizeof(s->av_data));
  return known ? 0 : -1;
}

/* A float running sum in index order: the shape of every sum in av() and in the
 * display reductions (reference: src/boltzmann_c_solver.c:417-423). */
float orc_seq_sum(const float *x, long n)
{
  float acc = 0;
  for (long i = 0; i < n; i++) acc = acc + x[i];
  return acc;
}

/* one output value of the display=8 frame (reference: src/boltzmann_solver.c:497-502) */
static float frame_value(const orc_params *p, const float *a, const float *b, int m, float phi_x)
{
  const size_t P = (size_t)p->MSIZE;
  float value = 0;
  for (int n = 0; n < p->N + 1; n++)
    value += a[n * P + m] * cos(n * phi_x) + b[n * P + m] * sin(n * phi_x);
  return value < 0 ? 0 : value;
}

/* the frame's values without the text: out[i*(M+1) + (m-1)] for phi_x[i], m = 1..M+1 */
void orc_frame_values(const orc_params *p, const float *a, const float *b,
                      const float *phi_x, int nphi, float *out)
{
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nphi; i++)
    for (int m = 1; m < p->M + 2; m++)
      out[(size_t)i * (p->M + 1) + (m - 1)] = frame_value(p, a, b, m, phi_x[i]);
}

/* reference: src/boltzmann_solver.c:487-507, read with the CPU row pitch M+3.
 * The norm there runs over the flat index 1..2M+1 (spills into row 1). */
void orc_print_2d_data(const orc_params *p, FILE *out, FILE *log,
                       const float *a, const float *b, float t)
{
  fprintf(out, "# t=%0.20f\n", t);
  float norm = 0;
  for (int m = 1; m < 2 * p->M + 2; m++) norm += a[m] * p->dPhi;
  norm *= 2 * ORC_PI * sqrt(p->alpha);

  for (float phi_x = -ORC_PI; phi_x < ORC_PI; phi_x += 0.01) {
    for (int m = 1; m < p->M + 2; m++)
      fprintf(out, "%0.5f %0.5f %0.20f\n", phi_x, orc_phi_y(p, m), frame_value(p, a, b, m, phi_x));
  }
  fprintf(out, "# norm=%0.20f\n", norm);
  fprintf(log, "# norm=%0.20f\n", norm);
}
