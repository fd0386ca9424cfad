/*
 * slb_solver.c -- host engine of the stepping loop: derived constants, device
 * arrays, initial state, tiptoe, the two sweeps per iteration with their
 * ping-pong bookkeeping, halo columns for phi_y slabs, and the host-side
 * display reductions.  Plays the part of the reference's main() up to and
 * including its time loop (reference: src/boltzmann_solver.c:97-253), split
 * into calls so that a single-GPU program, a benchmark and a one-process-per-GPU
 * driver can all use it.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slb_host.h"

#define HIP_OK(call) HandleError((int)(call), __FILE__, __LINE__)

/* reference: src/boltzmann_solver.c:97-117 */
void slb_derive_and_load(void)
{
  host_dPhi = (PhiYmax - PhiYmin) / host_M;

  NSIZE = host_N + 1;
  MSIZE = host_M + 3;
  /* rows start on 128-byte boundaries (32 floats) */
  PADDED_MSIZE = (MSIZE + 31) / 32 * 32;
  MP1 = host_M + 1;
  host_TMSIZE = host_M + 1;

  host_nu = 1 + host_dt / 2;
  host_nu2 = host_nu * host_nu;
  host_nu_tilde = 1 - host_dt / 2;
  host_bdt = host_B * host_dt / (4 * host_dPhi);

  load_data();
}

/* Lattice arrays are allocated one by one, and the k-th array of a run starts k*4096 bytes
 * into its allocation: the sweeps read the same cell of five arrays and write it in two or
 * four more at the same moment, and large allocations all start at the same offset within
 * the memory channels' interleaving pattern.  Measured on MI355X at n-harmonics 128,
 * g-grid 2^20: a two-sweep launch takes 0.670 ms with 4096 bytes between the starts
 * (0.675-0.684 with 1024-8192) against 0.708 ms with none; the double step is unaffected.
 * SLB_STAGGER=<bytes> (a multiple of 16) sets another distance. */
/* (user pointer, allocation) pairs of the live lattice arrays of this process; grows as needed */
static struct slb_alloc { void *user, *base; } *g_allocs = NULL;
static int g_allocs_cap = 0;
static int g_alloc_seq = 0;

static hipError_t lattice_alloc(float **out, size_t bytes)
{
  const char *e = getenv("SLB_STAGGER");
  long want = e ? atol(e) : 4096;
  if (want < 0) want = 0;
  if (want > (1L << 20)) want = 1L << 20;             /* 16 arrays: at most 15 MiB of padding each */
  const size_t stagger = (size_t)want / 16 * 16;
  const size_t off = stagger * (size_t)(g_alloc_seq++ % 16);
  int slot = -1;
  for (int i = 0; i < g_allocs_cap && slot < 0; i++) if (!g_allocs[i].user) slot = i;
  if (slot < 0) {
    const int cap = g_allocs_cap ? 2 * g_allocs_cap : 64;
    struct slb_alloc *t = (struct slb_alloc *)realloc(g_allocs, sizeof(*t) * (size_t)cap);
    if (!t) return hipErrorOutOfMemory;
    memset(t + g_allocs_cap, 0, sizeof(*t) * (size_t)(cap - g_allocs_cap));
    slot = g_allocs_cap;
    g_allocs = t; g_allocs_cap = cap;
  }
  void *base = NULL;
  hipError_t rc = hipMalloc(&base, bytes + off);
  if (rc != hipSuccess) return rc;
  *out = (float *)((char *)base + off);
  g_allocs[slot].user = *out; g_allocs[slot].base = base;
  return hipSuccess;
}

static void lattice_free(float *p)
{
  if (!p) return;
  for (int i = 0; i < g_allocs_cap; i++)
    if (g_allocs[i].user == p) { hipFree(g_allocs[i].base); g_allocs[i].user = g_allocs[i].base = NULL; return; }
  hipFree(p);
}

static float phi_y_of(int m) { return PhiYmin + host_dPhi * (m - 1); }

/* Equilibrium coefficients for global columns m_offset .. m_offset+cols-1
 * (reference: src/boltzmann_solver.c:119-126).  exp(-mu*phi_y^2/2) depends on
 * the column only and is evaluated once per column in long double; the
 * reference's product (float coefficient) x (long double expl) -> float is kept.
 * rowf / colf (may be NULL): the two factors as doubles, N+1 and `cols` of them. */
static void fill_a0(float *h, int pitch, int m_offset, int cols, double *rowf, double *colf)
{
  long double *gauss = (long double *)malloc(sizeof(long double) * (size_t)cols);
  for (int j = 0; j < cols; j++) {
    gauss[j] = expl(-host_mu * pow(phi_y_of(m_offset + j), 2) / 2);
    if (colf) colf[j] = (double)gauss[j];
  }
  const double i0 = slb_bessel_In(0, host_mu);
  for (int n = 0; n <= host_N; n++) {
    const float c = slb_bessel_In(n, host_mu) * (n == 0 ? 0.5 : 1) / (SLB_PI * i0)
                    * sqrt(host_mu / (2 * SLB_PI * host_alpha));
    if (rowf) rowf[n] = (double)c;
    float *row = h + (size_t)n * pitch;
    for (int j = 0; j < cols; j++) row[j] = c * gauss[j];
  }
  free(gauss);
}

/* a0 in separable form for the double-step kernel: the factors go to the device, the kernel's
 * own product is compared with the stored a0 in every cell, and a row with a cell that differs
 * (the double product was rounded once more than the reference's long double one) gets the
 * neighbouring doubles tried as its factor.  Only when every cell of every row agrees do the
 * factors stay; otherwise a0 is loaded as before.  SLB_A0=load skips all this. */
static void factor_a0(slb_solver *s, const double *rowf, const double *colf, int cols)
{
  const char *mode = getenv("SLB_A0");
  s->a0_row = s->a0_col = NULL;
  s->a0_rows_adjusted = -1;
  if (mode && strcmp(mode, "load") == 0) return;
  const int rows = s->N + 1;
  int *bad = NULL, *h_bad = (int *)malloc(sizeof(int) * (size_t)rows);
  double *colp = (double *)calloc((size_t)s->pitch, sizeof(double));
  if (!h_bad || !colp) { free(h_bad); free(colp); return; }
  memcpy(colp, colf, sizeof(double) * (size_t)cols);
  HIP_OK(hipMalloc((void **)&s->a0_row, sizeof(double) * (size_t)rows));
  HIP_OK(hipMalloc((void **)&s->a0_col, sizeof(double) * (size_t)s->pitch));
  HIP_OK(hipMalloc((void **)&bad, sizeof(int) * (size_t)rows));
  HIP_OK(hipMemcpy(s->a0_row, rowf, sizeof(double) * (size_t)rows, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(s->a0_col, colp, sizeof(double) * (size_t)s->pitch, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(bad, 0, sizeof(int) * (size_t)rows));
  free(colp);
  /* (the check runs on the caller's launch stream, the copies on the null stream: whole-device
   * waits keep them in order whatever kind of stream that is; this is set-up, not the loop) */
  HIP_OK(hipDeviceSynchronize());
  int ok = slb_a0_factors_check(s->a0, s->pitch, 0, rows, cols, s->a0_row, s->a0_col, bad) == 0;
  HIP_OK(hipDeviceSynchronize());
  if (ok) HIP_OK(hipMemcpy(h_bad, bad, sizeof(int) * (size_t)rows, hipMemcpyDeviceToHost));
  int adjusted = 0;
  for (int n = 0; ok && n < rows; n++) {
    if (h_bad[n] == 0) continue;
    /* the doubles next to the factor, nearest first, on both sides */
    int fixed = 0;
    double up = rowf[n], down = rowf[n];
    for (int k = 0; k < 32 && !fixed; k++) {
      up = nextafter(up, INFINITY); down = nextafter(down, -INFINITY);
      const double cand[2] = {up, down};
      for (int q = 0; q < 2 && !fixed; q++) {
        int count = 0;
        HIP_OK(hipMemcpy(s->a0_row + n, &cand[q], sizeof(double), hipMemcpyHostToDevice));
        HIP_OK(hipMemset(bad + n, 0, sizeof(int)));
        HIP_OK(hipDeviceSynchronize());
        if (slb_a0_factors_check(s->a0, s->pitch, n, 1, cols, s->a0_row, s->a0_col, bad) != 0) { ok = 0; break; }
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(&count, bad + n, sizeof(int), hipMemcpyDeviceToHost));
        fixed = count == 0;
      }
    }
    if (!fixed) ok = 0; else adjusted++;
  }
  HIP_OK(hipFree(bad));
  free(h_bad);
  if (!ok) {
    HIP_OK(hipFree(s->a0_row)); HIP_OK(hipFree(s->a0_col));
    s->a0_row = s->a0_col = NULL;
    return;
  }
  s->a0_rows_adjusted = adjusted;
}

slb_solver *slb_solver_create(int rank, int nranks)
{
  if (nranks < 1 || rank < 0 || rank >= nranks || host_M < nranks || host_N < 1) {
    fprintf(stderr, "slb_solver_create: bad partition (rank %d of %d, g-grid %d, n-harmonics %d)\n",
            rank, nranks, host_M, host_N);
    return NULL;
  }
  slb_solver *s = (slb_solver *)calloc(1, sizeof(*s));
  if (!s) return NULL;
  s->N = host_N; s->M = host_M; s->rank = rank; s->nranks = nranks;

  /* columns 1..M are dealt out in contiguous runs; the last slab also carries
   * column M+1 of the main lattice and the boundary column M+2.  A slab keeps
   * G ghost columns per side: 1 for a single slab (the reference layout, the
   * ghosts are the boundary columns 0 and M+2), 6 when there are neighbours, so
   * that a launch of up to three steps can recompute its earlier sweeps up to five
   * columns into the ghost zone and needs one exchange per launch. */
  const int base = host_M / nranks, rem = host_M % nranks;
  s->W = base + (rank < rem ? 1 : 0);
  s->G = nranks == 1 ? 1 : 6;
  if (nranks > 1 && s->W < 12) {
    fprintf(stderr, "slb_solver_create: slabs need at least 12 columns each\n");
    free(s);
    return NULL;
  }
  const int first = 1 + rank * base + (rank < rem ? rank : rem);
  s->m_offset = first - s->G;
  const int cols = s->W + 2 * s->G + 1;
  s->pitch = nranks == 1 ? PADDED_MSIZE : (cols + 31) / 32 * 32;

  const int last = rank == nranks - 1;
  const int G = s->G;
  s->main_slab = (slb_slab){s->pitch, s->m_offset, G, G + s->W - 1 + (last ? 1 : 0)};
  s->half_slab = (slb_slab){s->pitch, s->m_offset, G, G + s->W - 1};
  s->av_slab   = (slb_slab){s->pitch, s->m_offset, G, G + s->W - 1};

  const size_t cells = (size_t)(s->N + 1) * (size_t)s->pitch;
  const size_t bytes = cells * sizeof(float);
  float *h = (float *)calloc(cells, sizeof(float));
  if (!h) { fprintf(stderr, "slb_solver_create: out of host memory\n"); free(s); return NULL; }
  double *rowf = (double *)malloc(sizeof(double) * (size_t)(s->N + 1));
  double *colf = (double *)malloc(sizeof(double) * (size_t)cols);
  if (!rowf || !colf) { fprintf(stderr, "slb_solver_create: out of host memory\n"); free(h); free(rowf); free(colf); free(s); return NULL; }
  fill_a0(h, s->pitch, s->m_offset, cols, rowf, colf);

  HIP_OK(lattice_alloc(&s->a0, bytes));
  HIP_OK(hipMemcpy(s->a0, h, bytes, hipMemcpyHostToDevice));
  free(h);
  factor_a0(s, rowf, colf, cols);
  free(rowf); free(colf);
  for (int i = 0; i < 4; i++) {
    HIP_OK(lattice_alloc(&s->a[i], bytes));
    HIP_OK(lattice_alloc(&s->b[i], bytes));
    HIP_OK(hipMemset(s->a[i], 0, bytes));
    HIP_OK(hipMemset(s->b[i], 0, bytes));
  }
  HIP_OK(hipMemcpy(s->a[0], s->a0, bytes, hipMemcpyDeviceToDevice));
  HIP_OK(hipMalloc((void **)&s->av_data, 6 * sizeof(float)));
  HIP_OK(hipMemset(s->av_data, 0, 6 * sizeof(float)));
  HIP_OK(hipMalloc((void **)&s->sums, 4 * sizeof(double)));
  HIP_OK(hipMalloc((void **)&s->halo_buf, (size_t)48 * (s->N + 1) * sizeof(float)));
  HIP_OK(hipDeviceSynchronize());

  s->current = 0; s->next = 1; s->current_hs = 2; s->next_hs = 3;
  s->t = 0; s->frame_time = 0; s->steps = 0;
  return s;
}

void slb_solver_destroy(slb_solver *s)
{
  if (!s) return;
  lattice_free(s->a0);
  for (int i = 0; i < 4; i++) { lattice_free(s->a[i]); lattice_free(s->b[i]); }
  hipFree(s->av_data); hipFree(s->sums); hipFree(s->halo_buf);
  if (s->a0_row) { hipFree(s->a0_row); hipFree(s->a0_col); }
  lattice_free(s->spare_a); lattice_free(s->spare_b); lattice_free(s->spare_ah); lattice_free(s->spare_bh);
  free(s);
}

void slb_solver_set_stream(slb_solver *s, void *hip_stream)
{
  s->stream = hip_stream;
  s->has_stream = 1;
}

/* a launch of the library on this solver's stream: ON(s, rc = slb_sweep(...)); */
#define ON(s, statement) \
  do { if ((s)->has_stream) slb_push_stream((s)->stream); statement; if ((s)->has_stream) slb_pop_stream(); } while (0)
/* the stream this solver's launches go to */
static hipStream_t stream_of(const slb_solver *s)
{
  return (hipStream_t)(s->has_stream ? s->stream : slb_get_stream());
}

static void check_sweep(int rc, int line)
{
  if (rc > 0) HandleError(rc, __FILE__, line);
  if (rc < 0) {
    fprintf(stderr, "ERROR: a launch was refused (arguments or tuning variant; see the message above), %s line %d\n",
            __FILE__, line);
    exit(EXIT_FAILURE);
  }
}

/* centre and stencil are both the t=0 lattice; the result is the first
 * half-step lattice (reference: src/boltzmann_solver.c:161-165) */
void slb_solver_tiptoe(slb_solver *s)
{
  const float c0 = 1;
  const float c1 = cos(host_omega * (host_dt));
  { int rc_; ON(s, rc_ = slb_sweep(&s->main_slab, s->a0, s->a[s->current], s->b[s->current],
                        s->a[s->current], s->b[s->current],
                        s->a[s->current_hs], s->b[s->current_hs], c0, c1)); check_sweep(rc_, __LINE__); }
}

void slb_solver_sweep_main(slb_solver *s, float c0, float c1)
{
  { int rc_; ON(s, rc_ = slb_sweep(&s->main_slab, s->a0, s->a[s->current], s->b[s->current],
                        s->a[s->current_hs], s->b[s->current_hs],
                        s->a[s->next], s->b[s->next], c0, c1)); check_sweep(rc_, __LINE__); }
}

void slb_solver_sweep_half(slb_solver *s, float c0, float c1)
{
  { int rc_; ON(s, rc_ = slb_sweep(&s->half_slab, s->a0, s->a[s->current_hs], s->b[s->current_hs],
                        s->a[s->next], s->b[s->next],
                        s->a[s->next_hs], s->b[s->next_hs], c0, c1)); check_sweep(rc_, __LINE__); }
}

/* Part of a sweep: lattice 0 = main, 1 = half-step; local columns
 * j_first..j_last, clipped to what that sweep owns.  Lets a driver update the
 * slab's edge columns first, ship them, and update the interior meanwhile. */
void slb_solver_sweep_range(slb_solver *s, int lattice, int j_first, int j_last, float c0, float c1)
{
  slb_slab part = lattice == 0 ? s->main_slab : s->half_slab;
  if (j_first > part.j_first) part.j_first = j_first;
  if (j_last < part.j_last)   part.j_last = j_last;
  if (part.j_last < part.j_first) return;
  if (lattice == 0)
    { int rc_; ON(s, rc_ = slb_sweep(&part, s->a0, s->a[s->current], s->b[s->current],
                          s->a[s->current_hs], s->b[s->current_hs],
                          s->a[s->next], s->b[s->next], c0, c1)); check_sweep(rc_, __LINE__); }
  else
    { int rc_; ON(s, rc_ = slb_sweep(&part, s->a0, s->a[s->current_hs], s->b[s->current_hs],
                          s->a[s->next], s->b[s->next],
                          s->a[s->next_hs], s->b[s->next_hs], c0, c1)); check_sweep(rc_, __LINE__); }
}

void slb_solver_advance(slb_solver *s)
{
  s->t += host_dt;
  s->frame_time += host_dt;
  s->steps++;
}

void slb_solver_swap(slb_solver *s)
{
  const int c = s->current, h = s->current_hs;
  s->current = s->next;       s->next = c;
  s->current_hs = s->next_hs; s->next_hs = h;
}

/* float products and sums, double cos, float results
 * (reference: src/boltzmann_solver.c:204-214) */
void slb_step_cosines(float t, float c[4])
{
  const float t_hs = t + host_dt / 2;
  c[0] = cos(host_omega * t);
  c[1] = cos(host_omega * (t + host_dt));
  c[2] = cos(host_omega * t_hs);
  c[3] = cos(host_omega * (t_hs + host_dt));
}

void slb_solver_step(slb_solver *s)
{
  float c[4];
  slb_step_cosines(s->t, c);
  slb_solver_sweep_main(s, c[0], c[1]);
  slb_solver_sweep_half(s, c[2], c[3]);
  slb_solver_swap(s);
  slb_solver_advance(s);
}

/* Both sweeps in one launch for half-step-lattice columns j2_first..j2_last
 * (clipped to the slab's own); the main sweep is formed one column further out
 * on each side, as far as this slab may form it: up to the ghost column next to
 * a neighbour (recomputed here, bit-identical to the neighbour's own value),
 * not into a global boundary column.  Main-sweep columns outside this launch
 * are taken from memory, so a step may be issued as several column ranges. */
void slb_solver_step_fused_range(slb_solver *s, int j2_first, int j2_last, const float c[4])
{
  const int G = s->G;
  const int j1_lo = s->rank == 0 ? G : G - 1, j1_hi = G + s->W;
  slb_step_geom g;
  g.pitch = s->pitch; g.m_offset = s->m_offset;
  g.j2_first = j2_first > s->half_slab.j_first ? j2_first : s->half_slab.j_first;
  g.j2_last  = j2_last  < s->half_slab.j_last  ? j2_last  : s->half_slab.j_last;
  if (g.j2_last < g.j2_first) return;
  g.j1_first = g.j2_first - 1 > j1_lo ? g.j2_first - 1 : j1_lo;
  g.j1_last  = g.j2_last + 1 < j1_hi ? g.j2_last + 1 : j1_hi;
  { int rc_; ON(s, rc_ = slb_step_fused(&g, s->a0, s->a[s->current], s->b[s->current],
                             s->a[s->current_hs], s->b[s->current_hs],
                             s->a[s->next], s->b[s->next], s->a[s->next_hs], s->b[s->next_hs], c)); check_sweep(rc_, __LINE__); }
}

/* one loop iteration as a single fused launch; same results as slb_solver_step */
void slb_solver_step_fused(slb_solver *s)
{
  float c[4];
  slb_step_cosines(s->t, c);
  slb_solver_step_fused_range(s, s->half_slab.j_first, s->half_slab.j_last, c);
  slb_solver_swap(s);
  slb_solver_advance(s);
}

/* (re)build the spare arrays as full copies of the current lattices, so that every
 * cell the sweeps never write is what the reference would find in these buffers */
static void sync_spares(slb_solver *s)
{
  const size_t bytes = (size_t)(s->N + 1) * (size_t)s->pitch * sizeof(float);
  if (!s->spare_a) {
    HIP_OK(lattice_alloc(&s->spare_a, bytes));
    HIP_OK(lattice_alloc(&s->spare_b, bytes));
    HIP_OK(lattice_alloc(&s->spare_ah, bytes));
    HIP_OK(lattice_alloc(&s->spare_bh, bytes));
  }
  hipStream_t st = stream_of(s);
  HIP_OK(hipMemcpyAsync(s->spare_a,  s->a[s->current],    bytes, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(s->spare_b,  s->b[s->current],    bytes, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(s->spare_ah, s->a[s->current_hs], bytes, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(s->spare_bh, s->b[s->current_hs], bytes, hipMemcpyDeviceToDevice, st));
  s->spare_of = s->current; s->spare_of_hs = s->current_hs;
}

/* Double step for the columns j_first..j_last of the LAST sweep (half-step
 * lattice at t+5dt/2), clipped to the slab's own; each earlier sweep is formed
 * one column further out per stage, as far as this slab may: into its ghost
 * zone next to a neighbour, not into a global boundary column.  Results go to
 * the spare arrays; slb_solver_double_step_finish() makes them current.
 * c[0..3] = cosines of the first iteration, c[4..7] of the second. */
/* Bring the spare arrays up to date now, on the current stream, if the next double step
 * would have to (first use, or single steps since the last one): a caller that spreads
 * the ranges of a double step over several streams does this first. */
void slb_solver_prepare_double_step(slb_solver *s)
{
  if (!s->spare_a || s->spare_of != s->current || s->spare_of_hs != s->current_hs) sync_spares(s);
}

void slb_solver_double_step_range(slb_solver *s, int j_first, int j_last, const float c[8])
{
  slb_solver_double_step_range_rows(s, j_first, j_last, c, NULL, NULL);
}

/* the same, also writing the rows av() reads of both new main lattices (see
 * slb_solver_double_step_rows) at the columns this range forms */
void slb_solver_double_step_range_rows(slb_solver *s, int j_first, int j_last, const float c[8],
                                       float *rows_first, float *rows_second)
{
  if (!s->spare_a || s->spare_of != s->current || s->spare_of_hs != s->current_hs) sync_spares(s);
  const int G = s->G, first = s->rank == 0, last = s->rank == s->nranks - 1;
  /* outermost columns stage q (0 = A .. 3 = D) may form in this slab */
  int lo[4], hi[4];
  for (int q = 0; q < 4; q++) {
    lo[q] = first ? G : G - 3 + q;
    const int main_sweep = (q % 2) == 0;
    hi[q] = last ? G + s->W - 1 + (main_sweep ? 1 : 0) : G + s->W - 1 + 3 - q;
  }
  slb_step2_geom g;
  g.pitch = s->pitch; g.m_offset = s->m_offset;
  g.j_first[3] = j_first > lo[3] ? j_first : lo[3];
  g.j_last[3]  = j_last  < hi[3] ? j_last  : hi[3];
  if (g.j_last[3] < g.j_first[3]) return;
  for (int q = 2; q >= 0; q--) {
    g.j_first[q] = g.j_first[q + 1] - 1 > lo[q] ? g.j_first[q + 1] - 1 : lo[q];
    g.j_last[q]  = g.j_last[q + 1] + 1 < hi[q] ? g.j_last[q + 1] + 1 : hi[q];
  }
  if (s->a0_row) {
    int rc_; ON(s, rc_ = slb_step_fused2_factored(&g, s->a0, s->a0_row, s->a0_col, s->a[s->current], s->b[s->current],
                                   s->a[s->current_hs], s->b[s->current_hs],
                                   s->a[s->next], s->b[s->next], s->a[s->next_hs], s->b[s->next_hs],
                                   s->spare_a, s->spare_b, s->spare_ah, s->spare_bh, c,
                                   rows_first, rows_second)); check_sweep(rc_, __LINE__);
  } else {
    int rc_; ON(s, rc_ = slb_step_fused2_rows(&g, s->a0, s->a[s->current], s->b[s->current],
                                   s->a[s->current_hs], s->b[s->current_hs],
                                   s->a[s->next], s->b[s->next], s->a[s->next_hs], s->b[s->next_hs],
                                   s->spare_a, s->spare_b, s->spare_ah, s->spare_bh, c,
                                   rows_first, rows_second)); check_sweep(rc_, __LINE__);
  }
}

/* after two iterations the ping-pong indices are back where they were; the
 * arrays that now hold those lattices are the former spares */
void slb_solver_double_step_finish(slb_solver *s)
{
  float *x;
  x = s->a[s->current];    s->a[s->current]    = s->spare_a;  s->spare_a  = x;
  x = s->b[s->current];    s->b[s->current]    = s->spare_b;  s->spare_b  = x;
  x = s->a[s->current_hs]; s->a[s->current_hs] = s->spare_ah; s->spare_ah = x;
  x = s->b[s->current_hs]; s->b[s->current_hs] = s->spare_bh; s->spare_bh = x;
  slb_solver_advance(s);
  slb_solver_advance(s);
}

void slb_step2_cosines(float t, float c[8])
{
  slb_step_cosines(t, c);
  const float t2 = t + host_dt;                 /* the loop variable after one iteration */
  slb_step_cosines(t2, c + 4);
}

void slb_solver_double_step(slb_solver *s)
{
  float c[8];
  slb_step2_cosines(s->t, c);
  slb_solver_double_step_range(s, s->half_slab.j_first, s->half_slab.j_last, c);
  slb_solver_double_step_finish(s);
}

/* A double step that also leaves, in two [3][pitch] device buffers, the rows av() reads
 * (a rows 0 and 1, b row 1) of the main lattice after the first and after the second
 * iteration: averaging can then go on while two iterations share a launch. */
void slb_solver_double_step_rows(slb_solver *s, float *rows_first, float *rows_second)
{
  float c[8];
  slb_step2_cosines(s->t, c);
  slb_solver_double_step_range_rows(s, s->half_slab.j_first, s->half_slab.j_last, c, rows_first, rows_second);
  slb_solver_double_step_finish(s);
}

/* S loop iterations in one launch for the columns j_first..j_last of the LAST
 * sweep; every earlier sweep reaches one column further out per stage, as far
 * as this slab may form it.  Odd S: results go to a/b[next], a/b[next_hs];
 * even S: to the spare arrays.  slb_solver_multi_step_finish() makes them
 * current.  c = the cosines of the S loop times, 4 per iteration. */
void slb_solver_multi_step_range(slb_solver *s, int S, int j_first, int j_last, const float *c)
{
  slb_solver_multi_step_range_rows(s, S, j_first, j_last, c, NULL);
}

/* the same, also writing the rows av() reads of the S new main lattices into rows[0..S-1]
 * ([3][pitch] device buffers; S <= 3) at the columns this range forms; rows == NULL: none */
void slb_solver_multi_step_range_rows(slb_solver *s, int S, int j_first, int j_last, const float *c, float *const *rows)
{
  if (S < 1 || S > 4) { fprintf(stderr, "slb_solver_multi_step: 1 to 4 steps per launch\n"); exit(EXIT_FAILURE); }
  const int Q = 2 * S, even = (S % 2) == 0;
  if (even && (!s->spare_a || s->spare_of != s->current || s->spare_of_hs != s->current_hs)) sync_spares(s);
  const int G = s->G, first = s->rank == 0, last = s->rank == s->nranks - 1;
  if (s->nranks > 1 && Q > G) { fprintf(stderr, "slb_solver_multi_step: %d steps need %d ghost columns\n", S, Q); exit(EXIT_FAILURE); }
  slb_multi_geom g;
  memset(&g, 0, sizeof(g));
  g.pitch = s->pitch; g.m_offset = s->m_offset; g.steps = S;
  int lo[8], hi[8];
  for (int q = 0; q < Q; q++) {
    lo[q] = first ? G : G - (Q - 1 - q);
    const int main_sweep = (q % 2) == 0;
    hi[q] = last ? G + s->W - 1 + (main_sweep ? 1 : 0) : G + s->W - 1 + (Q - 1 - q);
  }
  g.j_first[Q - 1] = j_first > lo[Q - 1] ? j_first : lo[Q - 1];
  g.j_last[Q - 1]  = j_last  < hi[Q - 1] ? j_last  : hi[Q - 1];
  if (g.j_last[Q - 1] < g.j_first[Q - 1]) return;
  for (int q = Q - 2; q >= 0; q--) {
    g.j_first[q] = g.j_first[q + 1] - 1 > lo[q] ? g.j_first[q + 1] - 1 : lo[q];
    g.j_last[q]  = g.j_last[q + 1] + 1 < hi[q] ? g.j_last[q + 1] + 1 : hi[q];
  }
  const float *in[4] = {s->a[s->current], s->b[s->current], s->a[s->current_hs], s->b[s->current_hs]};
  float *out[4];
  if (even) { out[0] = s->spare_a; out[1] = s->spare_b; out[2] = s->spare_ah; out[3] = s->spare_bh; }
  else      { out[0] = s->a[s->next]; out[1] = s->b[s->next]; out[2] = s->a[s->next_hs]; out[3] = s->b[s->next_hs]; }
  /* sweep q writes iteration i = q/2; the reference's iteration i writes the main lattice into
   * `next` for even i and into `current` for odd i (likewise the half-step lattice) */
  const float *pa[8], *pb[8];
  for (int q = 0; q < Q - 1; q++) {
    const int i = q / 2, into_next = (i % 2) == 0;
    const int idx = (q % 2) == 0 ? (into_next ? s->next : s->current) : (into_next ? s->next_hs : s->current_hs);
    pa[q] = s->a[idx]; pb[q] = s->b[idx];
  }
  pa[Q - 2] = out[0]; pb[Q - 2] = out[1];
  if (rows) { int rc_; ON(s, rc_ = slb_step_multi_rows(&g, s->a0, in, pa, pb, out, c, rows)); check_sweep(rc_, __LINE__); }
  else      { int rc_; ON(s, rc_ = slb_step_multi(&g, s->a0, in, pa, pb, out, c)); check_sweep(rc_, __LINE__); }
}

void slb_solver_multi_step_finish(slb_solver *s, int S)
{
  if (S % 2 == 0) {
    float *x;
    x = s->a[s->current];    s->a[s->current]    = s->spare_a;  s->spare_a  = x;
    x = s->b[s->current];    s->b[s->current]    = s->spare_b;  s->spare_b  = x;
    x = s->a[s->current_hs]; s->a[s->current_hs] = s->spare_ah; s->spare_ah = x;
    x = s->b[s->current_hs]; s->b[s->current_hs] = s->spare_bh; s->spare_bh = x;
  } else {
    slb_solver_swap(s);
  }
  for (int i = 0; i < S; i++) slb_solver_advance(s);
}

void slb_steps_cosines(float t, int S, float *c)
{
  for (int i = 0; i < S; i++) {
    slb_step_cosines(t, c + 4 * i);
    t = t + host_dt;                          /* the loop variable, accumulated in float */
  }
}

void slb_solver_multi_step(slb_solver *s, int S)
{
  float c[16];
  slb_steps_cosines(s->t, S, c);
  slb_solver_multi_step_range(s, S, s->half_slab.j_first, s->half_slab.j_last, c);
  slb_solver_multi_step_finish(s, S);
}

void slb_solver_multi_step_rows(slb_solver *s, int S, float *const *rows)
{
  float c[16];
  slb_steps_cosines(s->t, S, c);
  slb_solver_multi_step_range_rows(s, S, s->half_slab.j_first, s->half_slab.j_last, c, rows);
  slb_solver_multi_step_finish(s, S);
}

/* How many loop iterations a launch of the time loop should take on this slab: 3 where the
 * three-step kernel has the lattice to fill the device, else 2.  Measured on MI355X (updates per
 * second, two against three steps per launch): 2.4e11 / 2.9e11 at 64 harmonics x 2^19 columns,
 * 2.2e11 / 3.4e11 at 128 x 2^19, 2.25e11 / 3.0e11 at 32 x 2^20, 2.6e11 / 3.9e11 at 512 x 2^20,
 * 2.2e11 / 2.5e11 at 32 x 2^19; among the narrower lattices 2.4e11 / 2.6e11 at 128 x 2^18,
 * 2.3e11 / 2.4e11 at 96 x 2^18, 2.3e11 / 2.6e11 at 256 x 2^17, 2.2e11 / 2.5e11 at 512 x 2^16 and
 * 2.2e11 / 2.7e11 at 2048 x 2^15, but a tie at 64 x 2^18 and at 128 x 2^17 (2.1e11 both) and
 * 2.6e11 / 2.3e11 at 16 x 2^21: three from 32 harmonics up where the lattice has 2^19 columns
 * or 24 million points.  SLB_STEPS_PER_LAUNCH=1..3 overrides. */
int slb_solver_steps_per_launch(const slb_solver *s)
{
  const char *e = getenv("SLB_STEPS_PER_LAUNCH");
  if (e && e[0] >= '1' && e[0] <= '3' && e[1] == 0) return e[0] - '0';
  const long waves = ((long)s->W + 239) / 240;
  return s->N >= 32 && (waves >= 2048 || (long)s->N * (long)s->W >= 24000000L) ? 3 : 2;
}

int slb_solver_run_timed(slb_solver *s, int steps, float *ms)
{
  hipStream_t st = stream_of(s);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, st));
  int i = 0;
  if (s->fused == 3) {                       /* three steps per launch, the rest in twos and ones */
    for (; i + 2 < steps; i += 3) slb_solver_multi_step(s, 3);
    for (; i + 1 < steps; i += 2) slb_solver_double_step(s);
  } else if (s->fused == 2) {
    for (; i + 1 < steps; i += 2) slb_solver_double_step(s);
  } else if (s->fused >= 10) {               /* 10 + S: the general S-steps-per-launch kernel */
    const int S = s->fused - 10;
    for (; i + S <= steps; i += S) slb_solver_multi_step(s, S);
  }
  for (; i < steps; i++) {
    if (s->fused) slb_solver_step_fused(s); else slb_solver_step(s);
  }
  HIP_OK(hipEventRecord(e1, st));
  HIP_OK(hipEventSynchronize(e1));
  HIP_OK(hipEventElapsedTime(ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

/* which: 0 a/b[current_hs], 1 a/b[next], 2 a/b[next_hs], 3 a/b[current],
 *        4 spare main lattice, 5 spare half-step lattice */
static void halo_arrays(const slb_solver *s, int which, float **a, float **b)
{
  if (which == 4)      { *a = s->spare_a;  *b = s->spare_b; }
  else if (which == 5) { *a = s->spare_ah; *b = s->spare_bh; }
  else {
    const int idx = which == 0 ? s->current_hs : which == 1 ? s->next : which == 2 ? s->next_hs : s->current;
    *a = s->a[idx]; *b = s->b[idx];
  }
}

int slb_solver_halo_pack(slb_solver *s, int which, int side, int ncols, float *buf)
{
  float *a, *b;
  halo_arrays(s, which, &a, &b);
  if (ncols < 1 || ncols > s->G || !a) return -1;
  const int col = side == 0 ? s->G : s->G + s->W - ncols;       /* my outermost owned columns */
  int rc;
  ON(s, rc = slb_halo_pack(a, b, s->pitch, s->N + 1, col, ncols, buf));
  return rc;
}

int slb_solver_halo_unpack(slb_solver *s, int which, int side, int ncols, const float *buf)
{
  float *a, *b;
  halo_arrays(s, which, &a, &b);
  if (ncols < 1 || ncols > s->G || !a) return -1;
  const int col = side == 0 ? s->G - ncols : s->G + s->W;       /* the ghost columns next to them */
  int rc;
  ON(s, rc = slb_halo_unpack(a, b, s->pitch, s->N + 1, col, ncols, buf));
  return rc;
}

/* After a double step's edge ranges: the outermost owned columns of the new main lattice
 * (3 columns) and of the new half-step lattice (4 columns), both still in the spare arrays,
 * gathered for each neighbour in one launch.  A side's buffer (NULL = no neighbour there)
 * holds 2*3*(N+1) floats of the main lattice followed by 2*4*(N+1) of the half-step one. */
/* The same after the edge ranges of a launch of S steps (before slb_solver_multi_step_finish): the
 * outermost nmain columns of the new main lattice and nhalf of the new half-step lattice -- which
 * are in the spare arrays for even S, in a/b[next] and a/b[next_hs] for odd S.  A side's buffer
 * holds 2*nmain*(N+1) floats of the main lattice followed by 2*nhalf*(N+1) of the half-step one. */
static int halo_multi(slb_solver *s, int S, int nmain, int nhalf, float *buf_left, float *buf_right, int pack)
{
  const int nrows = s->N + 1, even = (S % 2) == 0;
  if (!buf_left && !buf_right) return 0;          /* a slab without neighbours */
  if (S < 1 || S > 3 || (even && !s->spare_a) || nmain < 1 || nhalf < 1 || nmain > s->G || nhalf > s->G) return -1;
  slb_halo_seg seg[4];
  int k = 0;
  for (int side = 0; side < 2; side++) {
    float *buf = side == 0 ? buf_left : buf_right;
    if (!buf) continue;
    for (int lat = 0; lat < 2; lat++) {
      const int ncols = lat == 0 ? nmain : nhalf;
      seg[k].a = lat == 0 ? (even ? s->spare_a : s->a[s->next]) : (even ? s->spare_ah : s->a[s->next_hs]);
      seg[k].b = lat == 0 ? (even ? s->spare_b : s->b[s->next]) : (even ? s->spare_bh : s->b[s->next_hs]);
      seg[k].ncols = ncols;
      if (pack) seg[k].col = side == 0 ? s->G : s->G + s->W - ncols;      /* outermost owned columns */
      else      seg[k].col = side == 0 ? s->G - ncols : s->G + s->W;      /* the ghost columns next to them */
      seg[k].buf = buf + (lat == 0 ? 0 : 2 * nmain * nrows);
      k++;
    }
  }
  if (k == 0) return 0;
  int rc;
  ON(s, rc = pack ? slb_halo_pack_segs(seg, k, s->pitch, nrows) : slb_halo_unpack_segs(seg, k, s->pitch, nrows));
  return rc;
}

/* full depth: G-1 main and G half-step columns, what ANY launch that follows may read */
int slb_solver_halo_pack_multi(slb_solver *s, int S, float *buf_left, float *buf_right)
{
  return halo_multi(s, S, s->G - 1, s->G, buf_left, buf_right, 1);
}

int slb_solver_halo_unpack_multi(slb_solver *s, int S, float *buf_left, float *buf_right)
{
  return halo_multi(s, S, s->G - 1, s->G, buf_left, buf_right, 0);
}

/* the depth another double step reads: 3 and 4 columns */
int slb_solver_halo_pack_double(slb_solver *s, float *buf_left, float *buf_right)
{
  return halo_multi(s, 2, 3, 4, buf_left, buf_right, 1);
}

int slb_solver_halo_unpack_double(slb_solver *s, float *buf_left, float *buf_right)
{
  return halo_multi(s, 2, 3, 4, buf_left, buf_right, 0);
}

/* this slab's share of the three av() sums over its own columns of the CURRENT main
 * lattice (the one the reference passes as a/b[next] before it swaps); sums = 3
 * doubles on the device, to be added across slabs and given to slb_av_update() */
int slb_solver_av_partial(slb_solver *s, double *sums)
{
  int rc;
  ON(s, rc = slb_av_partial(&s->av_slab, s->a[s->current], s->b[s->current], sums));
  return rc;
}

/* the rows av() reads of the CURRENT main lattice (a rows 0 and 1, b row 1) copied into a
 * [3][pitch] device buffer on the launch stream: the form in which the double step hands
 * them out, for a single step in between */
int slb_solver_copy_av_rows(slb_solver *s, float *rows)
{
  const size_t row_bytes = sizeof(float) * (size_t)s->pitch;
  hipStream_t st = stream_of(s);
  hipError_t e = hipMemcpyAsync(rows, s->a[s->current], 2 * row_bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(rows + 2 * (size_t)s->pitch, s->b[s->current] + s->pitch, row_bytes,
                       hipMemcpyDeviceToDevice, st);
  return (int)e;
}

/* Frame values (display=3/7/8/9) of this slab's main-lattice columns (main_slab.j_first..
 * j_last; 1..M+1 for a single slab) for nphi values of phi_x: out (host) receives
 * nphi x ncols floats, row p = phi_x[p].  source 0: the current lattice a, b;
 * source 1: the equilibrium coefficients a0, cosine terms only.  The cos/sin factors come from the host's
 * libm with the argument n*phi_x formed in float, as in the reference
 * (src/boltzmann_solver.c:495-504); the sums run on the device. */
int slb_solver_frame_values_device(slb_solver *s, int source, const float *phi_x, int nphi, float *d_out)
{
  const int nrows = s->N + 1;
  const int col_first = s->main_slab.j_first;
  const int ncols = s->main_slab.j_last - s->main_slab.j_first + 1;
  if (nphi < 1 || ncols < 1) return -1;
  const size_t tab_bytes = sizeof(double) * 2 * (size_t)nphi * nrows;
  double *tab = (double *)malloc(tab_bytes);
  if (!tab) return -1;
  for (int p = 0; p < nphi; p++)
    for (int n = 0; n < nrows; n++) {
      tab[2 * ((size_t)p * nrows + n)]     = cos(n * phi_x[p]);
      tab[2 * ((size_t)p * nrows + n) + 1] = sin(n * phi_x[p]);
    }
  double *d_tab = NULL;
  HIP_OK(hipMalloc((void **)&d_tab, tab_bytes));
  HIP_OK(hipMemcpy(d_tab, tab, tab_bytes, hipMemcpyHostToDevice));
  free(tab);
  int rc;
  ON(s, rc = source == 1
      ? slb_frame_values(s->a0, NULL, s->pitch, nrows, col_first, ncols, d_tab, nphi, d_out)
      : slb_frame_values(s->a[s->current], s->b[s->current], s->pitch, nrows, col_first, ncols, d_tab, nphi, d_out));
  if (rc == 0) HIP_OK(hipStreamSynchronize(stream_of(s)));
  HIP_OK(hipFree(d_tab));
  return rc;
}

int slb_solver_frame_values(slb_solver *s, int source, const float *phi_x, int nphi, float *out)
{
  const int ncols = s->main_slab.j_last - s->main_slab.j_first + 1;
  if (nphi < 1 || ncols < 1) return -1;
  const size_t out_bytes = sizeof(float) * (size_t)nphi * ncols;
  float *d_out = NULL;
  HIP_OK(hipMalloc((void **)&d_out, out_bytes));
  const int rc = slb_solver_frame_values_device(s, source, phi_x, nphi, d_out);
  if (rc == 0) HIP_OK(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
  HIP_OK(hipFree(d_out));
  return rc;
}

int slb_solver_get_rows(slb_solver *s, int arr, int row0, int nrows, float *dst)
{
  if (arr < 0 || arr > 8 || row0 < 0 || nrows < 1 || row0 + nrows > s->N + 1) return -1;
  const float *src = arr == 0 ? s->a0 : (arr <= 4 ? s->a[arr - 1] : s->b[arr - 5]);
  HIP_OK(hipMemcpy(dst, src + (size_t)row0 * s->pitch, (size_t)nrows * s->pitch * sizeof(float),
                   hipMemcpyDeviceToHost));
  return 0;
}

int slb_solver_get_window(slb_solver *s, int arr, int col0, int ncols, float *dst)
{
  if (arr < 0 || arr > 8 || col0 < 0 || ncols < 1 || col0 + ncols > s->pitch) return -1;
  const float *src = arr == 0 ? s->a0 : (arr <= 4 ? s->a[arr - 1] : s->b[arr - 5]);
  HIP_OK(hipMemcpy2D(dst, (size_t)ncols * sizeof(float), src + col0, (size_t)s->pitch * sizeof(float),
                     (size_t)ncols * sizeof(float), (size_t)s->N + 1, hipMemcpyDeviceToHost));
  return 0;
}

/* ---- display reductions, evaluated on rows copied to the host ------------ */

/* reference: src/boltzmann_c_solver.c:289-296 */
float slb_host_norm(const float *a_row0, int M, float dPhi, float alpha)
{
  float norm = 0;
  for (int m = 1; m < M + 1; m++) norm += a_row0[m] * dPhi;
  norm *= 2 * SLB_PI * sqrt(alpha);
  return norm;
}

/* reference: src/boltzmann_solver.c:350-357 */
void slb_host_display4_sums(const float *a_rows01, const float *b_row1, int pitch, float out3[3])
{
  float v_dr = 0, v_y = 0, mmx = 0;
  for (int m = 1; m < host_M; m++) {
    v_dr += b_row1[m] * host_dPhi;
    v_y  += a_rows01[m] * phi_y_of(m) * host_dPhi;
    mmx  += a_rows01[pitch + m] * host_dPhi;
  }
  out3[0] = v_dr; out3[1] = v_y; out3[2] = mmx;
}

/* The CPU solver sums over its flat index 1..2M+1 with a row pitch of M+3, so
 * the sums run off the end of the row into the next one, and phi_y keeps
 * growing with the flat index (reference: src/boltzmann_c_solver.c:303-310).
 * The same elements are picked out of the padded device rows here. */
void slb_host_display77_sums(const float *a_rows012, const float *b_rows12, int pitch, float out3[3])
{
  const int cpu_pitch = host_M + 3;
  float v_dr = 0, v_y = 0, mmx = 0;
  for (int m = 1; m < 2 * host_M + 2; m++) {
    const int r = m >= cpu_pitch ? 1 : 0, c = m - r * cpu_pitch;
    v_dr += b_rows12[(size_t)r * pitch + c] * host_dPhi;
    v_y  += a_rows012[(size_t)r * pitch + c] * phi_y_of(m) * host_dPhi;
    mmx  += a_rows012[(size_t)(r + 1) * pitch + c] * host_dPhi;
  }
  out3[0] = v_dr; out3[1] = v_y; out3[2] = mmx;
}

/* reference: src/boltzmann_solver.c:359-361 */
void slb_host_multipliers(float k3[3])
{
  const double i0 = slb_bessel_In(0, host_mu), i1 = slb_bessel_In(1, host_mu);
  k3[0] = 2 * i0 * SLB_PI * sqrt(host_alpha) / i1;
  k3[1] = 4 * SLB_PI * i0 / i1;
  k3[2] = SLB_PI * host_alpha * sqrt(host_alpha);
}
