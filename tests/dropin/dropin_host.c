/*
 * dropin_host.c -- a host program in the shape of the reference's: it DEFINES the parameters
 * the stepping library imports (reference: src/boltzmann_cli.c:20-45 and
 * src/boltzmann_solver.c:47-65 define them, src/boltzmann_gpu.cu:40-44 imports them), owns all
 * device memory, and drives the library through the five boundary functions only
 * (reference: src/boltzmann_gpu.h:4-29) -- linked against libslb2d.so at build time, the way
 * the reference links boltzmann_gpu.o.  Test infrastructure: what it leaves in its output
 * file is compared with the CPU oracle by tests/test_dropin_gpu.py.
 *
 * usage: dropin_host N M B B2 omega steps1 steps2 a0.bin out.bin
 *   a0.bin : (N+1) x (M+3) floats, the equilibrium coefficients (formed by the test)
 *   steps1 loop iterations with B, then host_B = B2 and a second load_data() (the reference
 *   does this after new parameters arrive, src/boltzmann_solver.c:391: host_bdt keeps its
 *   value), then steps2 more.  av() after every iteration.
 *   out.bin: a[current], b[current], a[current_hs], b[current_hs] as (N+1) x (M+3) floats
 *   each, then av_data[6], then the loop time.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the boundary, as the reference's host declares it */
void load_data(void);
void step_on_grid(int blocks, float *a0, float *a_current, float *b_current, float *a_next, float *b_next,
                  float *a_current_hs, float *b_current_hs, float t, float t_hs, float cos_omega_t,
                  float cos_omega_t_plus_dt);
void step_on_half_grid(int blocks, float *a0, float *a_current, float *b_current, float *a_next, float *b_next,
                       float *a_current_hs, float *b_current_hs, float *a_next_hs, float *b_next_hs, float t,
                       float t_hs, float cos_omega_t, float cos_omega_t_plus_dt);
void av(int blocks, float *a, float *b, float *av_data, float t);
void HandleError(int err, const char *file, int line);
#define HANDLE_ERROR(err) (HandleError((int)(err), __FILE__, __LINE__))

/* the parameters: this executable's own definitions */
float host_E_dc = 1.0f, host_E_omega = 0.1f, host_omega, host_mu = 5.0f, host_alpha = 1.0f;
float PhiYmin = -4.0f, PhiYmax = 4.0f, host_B, t_start = 1.0f;
float host_dPhi, host_dt = 0.001f, host_bdt, host_nu_tilde, host_nu2, host_nu;
int   host_M, host_N, MSIZE, MP1, NSIZE, host_TMSIZE, PADDED_MSIZE;

int main(int argc, char **argv)
{
  if (argc != 10) { fprintf(stderr, "usage: %s N M B B2 omega steps1 steps2 a0.bin out.bin\n", argv[0]); return 2; }
  host_N = atoi(argv[1]); host_M = atoi(argv[2]); host_B = (float)atof(argv[3]);
  const float B2 = (float)atof(argv[4]);
  host_omega = (float)atof(argv[5]);
  const int steps1 = atoi(argv[6]), steps2 = atoi(argv[7]);

  /* derived values, as the reference's main() forms them (src/boltzmann_solver.c:97-115) */
  host_dPhi = (PhiYmax - PhiYmin) / host_M;
  NSIZE = host_N + 1; MSIZE = host_M + 3;
  PADDED_MSIZE = (MSIZE * (int)sizeof(float) + 127) / 128 * 128 / (int)sizeof(float);
  MP1 = host_M + 1; host_TMSIZE = host_M + 1;
  host_nu = 1 + host_dt / 2; host_nu2 = host_nu * host_nu; host_nu_tilde = 1 - host_dt / 2;
  host_bdt = host_B * host_dt / (4 * host_dPhi);
  load_data();

  const size_t cells = (size_t)NSIZE * PADDED_MSIZE, bytes = cells * sizeof(float);
  float *h = (float *)calloc(cells, sizeof(float)), *row = (float *)malloc(sizeof(float) * MSIZE);
  FILE *f = fopen(argv[8], "rb");
  if (!f || !h || !row) { perror("a0"); return 2; }
  for (int n = 0; n < NSIZE; n++) {
    if (fread(row, sizeof(float), MSIZE, f) != (size_t)MSIZE) { fprintf(stderr, "short a0 file\n"); return 2; }
    memcpy(h + (size_t)n * PADDED_MSIZE, row, sizeof(float) * MSIZE);
  }
  fclose(f);

  float *a0, *a[4], *b[4], *av_data;
  HANDLE_ERROR(hipMalloc((void **)&a0, bytes));
  HANDLE_ERROR(hipMemcpy(a0, h, bytes, hipMemcpyHostToDevice));
  for (int i = 0; i < 4; i++) {
    HANDLE_ERROR(hipMalloc((void **)&a[i], bytes));
    HANDLE_ERROR(hipMalloc((void **)&b[i], bytes));
    HANDLE_ERROR(hipMemset(a[i], 0, bytes));
    HANDLE_ERROR(hipMemset(b[i], 0, bytes));
  }
  HANDLE_ERROR(hipMemcpy(a[0], h, bytes, hipMemcpyHostToDevice));
  HANDLE_ERROR(hipMalloc((void **)&av_data, 6 * sizeof(float)));
  HANDLE_ERROR(hipMemset(av_data, 0, 6 * sizeof(float)));

  int current = 0, next = 1, current_hs = 2, next_hs = 3;
  const int blocks = (host_M + 3) / 128;
  /* the first half step (reference: src/boltzmann_solver.c:161-165) */
  step_on_grid(blocks, a0, a[current], b[current], a[current_hs], b[current_hs], a[current], b[current],
               0, 0, 1, cos(host_omega * host_dt));
  float t = 0;
  for (int k = 0; k < steps1 + steps2; k++, t += host_dt) {
    if (k == steps1) { host_B = B2; load_data(); }
    const float t_hs = t + host_dt / 2;
    float c0 = cos(host_omega * t), c1 = cos(host_omega * (t + host_dt));
    step_on_grid(blocks, a0, a[current], b[current], a[next], b[next], a[current_hs], b[current_hs], t, t_hs, c0, c1);
    HANDLE_ERROR(hipDeviceSynchronize());
    c0 = cos(host_omega * t_hs); c1 = cos(host_omega * (t_hs + host_dt));
    step_on_half_grid(blocks, a0, a[current], b[current], a[next], b[next], a[current_hs], b[current_hs],
                      a[next_hs], b[next_hs], t, t_hs, c0, c1);
    av(blocks, a[next], b[next], av_data, t);
    int x = current; current = next; next = x;
    x = current_hs; current_hs = next_hs; next_hs = x;
  }
  HANDLE_ERROR(hipDeviceSynchronize());

  f = fopen(argv[9], "wb");
  if (!f) { perror("out"); return 2; }
  float *src[4] = {a[current], b[current], a[current_hs], b[current_hs]};
  for (int q = 0; q < 4; q++) {
    HANDLE_ERROR(hipMemcpy(h, src[q], bytes, hipMemcpyDeviceToHost));
    for (int n = 0; n < NSIZE; n++) fwrite(h + (size_t)n * PADDED_MSIZE, sizeof(float), MSIZE, f);
  }
  float av6[6];
  HANDLE_ERROR(hipMemcpy(av6, av_data, sizeof(av6), hipMemcpyDeviceToHost));
  fwrite(av6, sizeof(float), 6, f);
  fwrite(&t, sizeof(float), 1, f);
  fclose(f);
  HANDLE_ERROR(hipFree(a0));
  for (int i = 0; i < 4; i++) { HANDLE_ERROR(hipFree(a[i])); HANDLE_ERROR(hipFree(b[i])); }
  HANDLE_ERROR(hipFree(av_data));
  free(h); free(row);
  return 0;
}
