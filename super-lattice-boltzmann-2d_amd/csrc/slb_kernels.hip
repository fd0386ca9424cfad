// slb_kernels.hip -- gfx950 (MI355X) kernels and the C boundary of libslb2d.so.
//
// What runs here: the two lattice sweeps of the Boltzmann stepping loop (one per
// launch, both of a time step in one launch, two or more time steps in one launch),
// the harmonic reductions (exact-order float sums evaluated in parallel, or a tree),
// the display=8 frame sums and the halo movers for phi_y slabs, behind the entry
// points declared in include/slb_abi.h.  The kernels themselves are in the slb_*.inc files beside
// this one, included below in the order they build on each other; this file holds the state
// load_data() fills, the launch schedules, the device workspace and the entry points.
// Arithmetic is IEEE binary32, one rounding per operation, in the association
// order of the reference CPU sweeps (reference: src/boltzmann_c_solver.c:363-378),
// so that fields are bit-identical to the CPU solver.  This file must be compiled
// with -ffp-contract=off and without any fast-math option.
//
// Layout: every array is [N+1][pitch] floats, row n holds harmonic n, column m
// the phi_y grid point.  Lanes run along m (coalesced); a lane owns V adjacent
// columns and walks n upward, keeping the differences X[r][m+1]-X[r][m-1] of the
// two stencil rows it will need again in registers.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/slb_abi.h"

// ---------------------------------------------------------------------------
// parameters snapshotted by load_data()
// ---------------------------------------------------------------------------
namespace {

struct Physics {
  float E_dc, E_omega, omega, B, dt, dPhi, PhiYmin, nu, nu2, nu_tilde, bdt;
  int   N, M, pitch, TMSIZE;
  bool  loaded;
};
Physics     g_phys = {};
hipStream_t g_stream = nullptr;      // slb_set_stream(): the default launch stream of the process
// slb_push_stream(): the launch stream of the calling thread while the push is in force
constexpr int kStreamStack = 8;
thread_local hipStream_t t_stream[kStreamStack];
thread_local int         t_stream_depth = 0;
inline hipStream_t launch_stream() { return t_stream_depth > 0 ? t_stream[t_stream_depth - 1] : g_stream; }
int         g_av_mode = SLB_AV_EXACT;
int         g_variant = 0, g_n_chunk = 0, g_block = 0;
unsigned long long *g_stamps = nullptr;   // timing builds: slb_debug_set_stamps()
double     *g_ws = nullptr;          // reduction workspace, allocated once
int         g_ws_device = -1;
constexpr int kMaxAvBlocks = 1024;
constexpr int kAvColsPerBlock = 256 * 8;   // columns a block of the tree sums takes per pass

struct SweepArgs {
  const float *a0, *a_c, *b_c, *a_s, *b_s;
  float *a_out, *b_out;
  float E_dc, E_omega, B, dt, dPhi, PhiYmin, nu, nu2, nu_tilde, bdt, cos_t, cos_t1;
  int N, pitch, j_first, j_last, m_offset, n_chunk;
  int jj0;          // first column that gets a lane (multiple of V, <= j_first)
};

// The kernels, in the order they build on each other (each file continues this translation unit,
// inside the unnamed namespace):
#include "slb_point.inc"    // the point update and the row loads every kernel shares
#include "slb_sweep.inc"    // one sweep per launch: sweep_kernel, sweep_pipe_kernel
#include "slb_fused.inc"    // both sweeps of a time step in one launch (fused_step_kernel) and the shorter, bit-identical forms of the point update
#include "slb_fused2.inc"   // two time steps per launch: fused2_step_kernel
#include "slb_multi.inc"    // S time steps per launch, general forms: multi_step_kernel, pipe_step_kernel
#include "slb_ring.inc"     // three time steps per launch with the stage outputs in an LDS ring: ring_step_kernel
#include "slb_halo.inc"     // halo movers for phi_y slabs
#include "slb_reduce.inc"   // harmonic reductions (tree and exact-order sums) and the display=8 frame kernel

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
constexpr int kMaxDevices = 64;
constexpr size_t kWsDoubles = kMaxAvBlocks * 3 + 4;      // block partials, then 4 doubles for totals
double *g_ws_of[kMaxDevices] = {};                       // one workspace per device, kept until exit

// the reduction workspace of the current device (allocated on first use there) -> g_ws
int ensure_workspace()
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) { fprintf(stderr, "libslb2d: device index %d not supported\n", dev); return -1; }
  if (!g_ws_of[dev]) {
    e = hipMalloc((void **)&g_ws_of[dev], sizeof(double) * kWsDoubles);
    if (e != hipSuccess) { g_ws_of[dev] = nullptr; return (int)e; }
  }
  g_ws = g_ws_of[dev];
  g_ws_device = dev;
  return 0;
}

// ---------------------------------------------------------------------------
// av() for callers that cannot allocate (code under HIP graph capture, serving
// loops, harnesses that forbid hipMalloc): the reduction workspace lives in a
// statically allocated device array instead of a hipMalloc'ed one.  Every GPU
// the library runs on holds its own copy of the array (kWsDoubles * 8 = 24 608
// bytes) until the process ends.  In host code the array's name is only a
// handle: its device address is looked up once per device and kept.
// ---------------------------------------------------------------------------
static __device__ __align__(256) double scratch[kWsDoubles];
double *g_scratch_of[kMaxDevices] = {};                  // device address of `scratch`, per device

int static_workspace(double **ws)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) { fprintf(stderr, "libslb2d: device index %d not supported\n", dev); return -1; }
  if (!g_scratch_of[dev]) {
    double *ptr = nullptr;
    e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!ptr) return -1;
    g_scratch_of[dev] = ptr;
  }
  *ws = g_scratch_of[dev];
  return 0;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Rows per workgroup when the caller leaves it to the library.  A wave walks its rows one
// after the other, so a launch lasts at least as long as one chunk's row loop: wide
// lattices (a thousand workgroups across already) take the chunk length measured fastest
// at full size, narrower ones are cut into shorter chunks -- down to 4 rows -- until about
// a thousand workgroups exist, because there the row loop is all the launch costs.
int auto_chunk(int N, int cols, int cols_per_wave, int block, int full_size_chunk, int target_wgs = 1024)
{
  const int waves = (cols + cols_per_wave - 1) / cols_per_wave;
  const int wgs_x = (waves + block / 64 - 1) / (block / 64);
  int chunks = (target_wgs + wgs_x - 1) / wgs_x;
  const int most = N / 4 > 0 ? N / 4 : 1;
  if (chunks > most) chunks = most;
  int nc = (N + chunks - 1) / chunks;
  if (nc > full_size_chunk) nc = full_size_chunk;
  return nc < 1 ? 1 : nc;
}

template <int V, int XCH>
int launch_sweep(SweepArgs &args, int block)
{
  args.jj0 = args.j_first / V * V;
  const int cols   = args.j_last + 1 - args.jj0;          // columns jj0..j_last get a lane
  const int lanes  = (cols + V - 1) / V;
  dim3 grid((lanes + block - 1) / block, (args.N + args.n_chunk - 1) / args.n_chunk);
  hipLaunchKernelGGL((sweep_kernel<V, XCH>), grid, dim3(block), 0, launch_stream(), args);
  return (int)hipGetLastError();
}

template <bool NT, bool NTS = NT>
int launch_pipe(SweepArgs &args, int block)
{
  args.jj0 = args.j_first / 4 * 4;
  const int cols  = args.j_last + 1 - args.jj0;
  const int lanes = (cols + 3) / 4;
  dim3 grid((lanes + block - 1) / block, (args.N + args.n_chunk - 1) / args.n_chunk);
  hipLaunchKernelGGL((sweep_pipe_kernel<NT, NTS>), grid, dim3(block), 0, launch_stream(), args);
  return (int)hipGetLastError();
}

template <bool NT, int MINW, bool DIAG = false, bool NTS = NT>
int launch_fused(FusedArgs &args, int block)
{
  const int lo = args.j1_first < args.j2_first ? args.j1_first : args.j2_first;
  const int hi = args.j1_last > args.j2_last ? args.j1_last : args.j2_last;
  args.jj0 = lo / 4 * 4;
  const int per_wave = 62 * 4;                          // columns a wave's interior lanes cover
  const int waves = (hi + 1 - args.jj0 + per_wave - 1) / per_wave;
  const int wpb = block / 64;
  dim3 grid((waves + wpb - 1) / wpb, (args.N + args.n_chunk - 1) / args.n_chunk);
  hipLaunchKernelGGL((fused_step_kernel<NT, MINW, DIAG, NTS>), grid, dim3(block), 0, launch_stream(), args);
  return (int)hipGetLastError();
}

// workgroups of `block` threads of a kernel that the current device holds at once (0 if unknown)
template <class K>
int resident_workgroups(K kernel, int block)
{
  static int cache[kMaxDevices][5] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
  int &slot = cache[dev][block / 64];
  if (slot == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    slot = per_cu * cus > 0 ? per_cu * cus : -1;
  }
  return slot > 0 ? slot : 0;
}

// n_chunk > 0: every strip group is cut into chunks of n_chunk rows (one phase).
// n_chunk == 0 (the library chooses): on a lattice of at least one round of workgroups, as
// many strip groups as fill the device a whole number of times take all their rows in one
// workgroup each -- such a workgroup reads every input row once and forms no row twice --
// and only the groups left over are cut into chunks (`tail_chunk` rows, or shorter ones when
// the lattice has few rounds); narrower lattices are cut into chunks of `tail_chunk` rows
// throughout.
// `slots`: workgroups of the kernel the device holds at once (0: unknown).  H: halo lanes per side.
void schedule_fused2(Fused2Args &args, int H, int block, int n_chunk, int tail_chunk, int slots)
{
  int lo = args.jf[0], hi = args.jl[0];
  for (int q = 1; q < 4; q++) { lo = args.jf[q] < lo ? args.jf[q] : lo; hi = args.jl[q] > hi ? args.jl[q] : hi; }
  args.jj0 = H == 1 ? lo / 4 * 4 : lo / 32 * 32;         // wide halos: strips start on line boundaries
  const int per_wave = (64 - 2 * H) * 4;
  const int waves = (hi + 1 - args.jj0 + per_wave - 1) / per_wave;
  const int wpb = block / 64;
  const int groups = (waves + wpb - 1) / wpb;
  auto chunks_of = [&](int len) { return (args.N + len - 1) / len; };
  if (n_chunk > 0) slots = 0;
  // (measured on MI355X with two halo lanes and non-temporal stores: whole groups first wins
  // from one round of workgroups up -- 2.02 against 2.11 ms at 128 rows x 2^21 columns, 2.06
  // against 2.15 at 256 x 2^20, 0.53 against 0.57 at 64 x 2^20, 1.02-1.04 against 1.03-1.08 at
  // 128 x 2^20 -- provided the strips left over are cut finely enough to fill the device)
  // (tuning variant 34: the earlier rule -- below four rounds of workgroups every strip is cut)
  const int min_rounds = g_variant == 34 ? 4 : 1;
  const int whole = slots > 0 && groups >= min_rounds * slots ? groups / slots * slots : 0;
  if (whole > 0 && whole < groups && groups < 4 * slots && g_variant != 34) {
    // few rounds: the strips left over are cut into as many chunks as make their workgroups
    // fill the device once, three at least
    int c = (slots + (groups - whole) - 1) / (groups - whole);
    c = c < 3 ? 3 : c;
    tail_chunk = (args.N + c - 1) / c;
    if (tail_chunk < 8) tail_chunk = args.N < 8 ? args.N : 8;
  }
  if (n_chunk <= 0 && whole == 0) n_chunk = tail_chunk;
  if (n_chunk > 0) {
    if (n_chunk > args.N) n_chunk = args.N;
    args.n_phase = 1;
    args.g0[0] = 0; args.ng[0] = groups; args.len[0] = n_chunk;
    args.wg_begin[0] = 0; args.wg_begin[1] = groups * chunks_of(n_chunk);
  } else {
    if (tail_chunk > args.N) tail_chunk = args.N;
    args.n_phase = whole < groups ? 2 : 1;
    args.g0[0] = 0; args.ng[0] = whole; args.len[0] = args.N;
    args.wg_begin[0] = 0; args.wg_begin[1] = (whole + 7) / 8 * 8;
    args.g0[1] = whole; args.ng[1] = groups - whole; args.len[1] = tail_chunk;
    args.wg_begin[2] = args.wg_begin[1] + (groups - whole) * chunks_of(tail_chunk);
  }
  for (int k = args.n_phase; k < 3; k++) { args.g0[k] = 0; args.ng[k] = 1; args.len[k] = args.N; args.wg_begin[k + 1] = args.wg_begin[k]; }
}

template <int MINW, bool DIAG = false, bool ROWS = false, int H = 2, bool NTS = true, bool NTL = false, int A0F = 0, int OPT = 0>
int launch_fused2(Fused2Args &args, int block, int n_chunk, int tail_chunk)
{
  // (tuning variant 39: as if the occupancy query had failed -- every strip is cut into chunks)
  const int slots = n_chunk > 0 || g_variant == 39 ? 0 : resident_workgroups(fused2_step_kernel<MINW, DIAG, ROWS, H, NTS, NTL, A0F, OPT>, block);
  schedule_fused2(args, H, block, n_chunk, tail_chunk, slots);
  hipLaunchKernelGGL((fused2_step_kernel<MINW, DIAG, ROWS, H, NTS, NTL, A0F, OPT>), dim3(args.wg_begin[args.n_phase]), dim3(block), 0,
                     launch_stream(), args);
  return (int)hipGetLastError();
}

// The scheduled general kernels: whole strip groups first, the rest in chunks (see launch_fused2).
// `slots`: workgroups of the kernel the device holds at once (0: unknown, or n_chunk given).
// edge_first: the two strip groups at the ends of the lattice go first, in short chunks.  Their outermost
// waves fetch never-written cells stage by stage (a memory round trip each, five to a pass at three steps
// per launch), which makes a pass of theirs several times as long as anyone else's; as whole strips they
// would end long after the rest of the launch.
template <int S>
PipeArgs<S> make_pipe_args(const slb_multi_geom *g, const float *a0, const float *const *in, const float *const *pers_a,
                           const float *const *pers_b, float *const *out, const float *cosines, int H, int block,
                           int n_chunk, int slots, bool edge_first = false, int fewer_rounds = 0, int rest_chunk = 0,
                           bool sub_round_whole = false)
{
  constexpr int Q = 2 * S;
  PipeArgs<S> s;
  s.a0 = a0; s.a_c = in[0]; s.b_c = in[1]; s.a_h = in[2]; s.b_h = in[3];
  for (int q = 0; q < Q - 1; q++) { s.pers_a[q] = pers_a[q]; s.pers_b[q] = pers_b[q]; }
  s.a_o = out[0]; s.b_o = out[1]; s.a_ho = out[2]; s.b_ho = out[3];
  s.E_dc = g_phys.E_dc; s.E_omega = g_phys.E_omega; s.B = g_phys.B; s.dt = g_phys.dt;
  s.dPhi = g_phys.dPhi; s.PhiYmin = g_phys.PhiYmin; s.nu = g_phys.nu; s.nu2 = g_phys.nu2;
  s.nu_tilde = g_phys.nu_tilde; s.bdt = g_phys.bdt;
  for (int q = 0; q < 2 * Q; q++) s.c[q] = cosines[q];
  s.N = g_phys.N; s.pitch = g->pitch; s.m_offset = g->m_offset;
  int lo = g->j_first[0], hi = g->j_last[0];
  for (int q = 0; q < Q; q++) {
    s.jf[q] = g->j_first[q]; s.jl[q] = g->j_last[q];
    lo = s.jf[q] < lo ? s.jf[q] : lo; hi = s.jl[q] > hi ? s.jl[q] : hi;
  }
  s.jj0 = lo / 32 * 32;
  const int per_wave = (64 - 2 * H) * 4;
  const int waves = (hi + 1 - s.jj0 + per_wave - 1) / per_wave;
  const int wpb = block / 64;
  const int groups = (waves + wpb - 1) / wpb;
  auto chunks_of = [&](int len) { return (s.N + len - 1) / len; };
  if (n_chunk > 0) slots = 0;
  s.edge_last = 0;
  // (tuning: on a lattice of less than one round of workgroups, whole strips all the same -- no row is
  // formed twice, but the device is not full)
  const bool sub = sub_round_whole && edge_first && slots > 0 && n_chunk <= 0 && groups >= 3 && groups < slots + 2;
  const bool edges = edge_first && slots > 0 && (groups >= slots + 2 || sub);
  const int inner = edges ? groups - 2 : groups;          // strip groups other than the two at the ends
  int whole = slots > 0 && inner >= slots ? inner / slots * slots : (sub ? inner : 0);
  int tail_chunk = s.N >= 128 ? (7 * s.N + 15) / 16 : 32;
  // (tuning: fewer_rounds of whole strips less, the strips left over in chunks of rest_chunk rows)
  if (fewer_rounds > 0 && whole >= (fewer_rounds + 1) * slots) {
    whole -= fewer_rounds * slots;
    tail_chunk = rest_chunk > 0 ? rest_chunk : (s.N + 1) / 2;
  } else if (whole > 0 && whole < inner && inner < 4 * slots) {
    int c = (slots + (inner - whole) - 1) / (inner - whole);
    c = c < 3 ? 3 : c;
    tail_chunk = (s.N + c - 1) / c;
    if (tail_chunk < 4 * Q) tail_chunk = s.N < 4 * Q ? s.N : 4 * Q;     // a chunk re-forms Q-1 rows above itself
  }
  if (n_chunk <= 0 && whole == 0) n_chunk = auto_chunk(s.N, hi - lo + 1, per_wave, block, tail_chunk, 2 * (slots > 0 ? slots : 768));
  if (n_chunk > 0) {
    if (n_chunk > s.N) n_chunk = s.N;
    s.n_phase = 1;
    s.g0[0] = 0; s.ng[0] = groups; s.len[0] = n_chunk;
    s.wg_begin[0] = 0; s.wg_begin[1] = groups * chunks_of(n_chunk);
  } else if (edges) {
    if (tail_chunk > s.N) tail_chunk = s.N;
    const int edge_chunk = s.N < 16 ? s.N : 16;
    s.n_phase = whole < inner ? 3 : 2;
    s.edge_last = groups - 1;
    s.g0[0] = 0; s.ng[0] = 2; s.len[0] = edge_chunk;
    s.wg_begin[0] = 0; s.wg_begin[1] = (2 * chunks_of(edge_chunk) + 7) / 8 * 8;
    s.g0[1] = 1; s.ng[1] = whole; s.len[1] = s.N;
    s.wg_begin[2] = s.wg_begin[1] + (whole + 7) / 8 * 8;
    s.g0[2] = 1 + whole; s.ng[2] = inner - whole; s.len[2] = tail_chunk;
    s.wg_begin[3] = s.wg_begin[2] + (inner - whole) * chunks_of(tail_chunk);
  } else {
    if (tail_chunk > s.N) tail_chunk = s.N;
    s.n_phase = whole < groups ? 2 : 1;
    s.g0[0] = 0; s.ng[0] = whole; s.len[0] = s.N;
    s.wg_begin[0] = 0; s.wg_begin[1] = (whole + 7) / 8 * 8;
    s.g0[1] = whole; s.ng[1] = groups - whole; s.len[1] = tail_chunk;
    s.wg_begin[2] = s.wg_begin[1] + (groups - whole) * chunks_of(tail_chunk);
  }
  for (int k = s.n_phase; k < 3; k++) { s.g0[k] = 0; s.ng[k] = 1; s.len[k] = s.N; s.wg_begin[k + 1] = s.wg_begin[k]; }
  static const bool show = getenv("SLB_SHOW_SCHEDULE") != nullptr;
  if (show)
    fprintf(stderr, "slb schedule: S=%d slots=%d groups=%d phases=%d  (groups x rows) %d x %d, %d x %d, %d x %d  edge_last=%d  workgroups=%d\n",
            S, slots, groups, s.n_phase, s.ng[0], s.len[0], s.n_phase > 1 ? s.ng[1] : 0, s.n_phase > 1 ? s.len[1] : 0,
            s.n_phase > 2 ? s.ng[2] : 0, s.n_phase > 2 ? s.len[2] : 0, s.edge_last, s.wg_begin[s.n_phase]);
  return s;
}

template <int S, int MINW, int H, bool NTS, bool DIV2, bool PF, bool RING = false>
int launch_pipe_sched(const slb_multi_geom *g, const float *a0, const float *const *in, const float *const *pers_a,
                      const float *const *pers_b, float *const *out, const float *cosines, int block, int n_chunk)
{
  const int slots = n_chunk > 0 ? 0 : resident_workgroups(pipe_step_kernel<S, MINW, H, NTS, DIV2, PF, RING>, block);
  const PipeArgs<S> s = make_pipe_args<S>(g, a0, in, pers_a, pers_b, out, cosines, H, block, n_chunk, slots);
  hipLaunchKernelGGL((pipe_step_kernel<S, MINW, H, NTS, DIV2, PF, RING>), dim3(s.wg_begin[s.n_phase]), dim3(block), 0, launch_stream(), s);
  return (int)hipGetLastError();
}

// rows: S buffers of [3][pitch] floats when ROWS, else ignored
// May the two quotients of a point share one reciprocal (div2_shared)?  Their denominator is
// nu^2 + mu_t1^2 with |mu_t1| <= N (|E_dc| + |E_omega| + |B| max|phi_y|) dt / 2; the shared form equals
// the compiler's division for denominators from 0.5 to 2^100 (slb_selftest_division).
bool shared_division_ok(int m_offset, int pitch)
{
  const double phi_lo = (double)g_phys.PhiYmin + (double)g_phys.dPhi * (double)(m_offset - 1);
  const double phi_hi = (double)g_phys.PhiYmin + (double)g_phys.dPhi * (double)(m_offset + pitch);
  const double phi = fabs(phi_lo) > fabs(phi_hi) ? fabs(phi_lo) : fabs(phi_hi);
  const double mu1 = (double)g_phys.N * (fabs((double)g_phys.E_dc) + fabs((double)g_phys.E_omega) + fabs((double)g_phys.B) * phi) *
                     fabs((double)g_phys.dt) * 0.5 * 1.0001;
  const double lo = (double)g_phys.nu2, hi = (double)g_phys.nu2 + mu1 * mu1;
  return lo >= 0.5 && hi <= 0x1p100 && hi == hi;      // (NaN parameters fail the last test)
}

template <int S, int MINW, bool ROWS = false, bool DIV2 = true, int X = 0>
int launch_ring(const slb_multi_geom *g, const float *a0, const float *const *in, const float *const *pers_a,
                const float *const *pers_b, float *const *out, const float *cosines, int n_chunk, float *const *rows = nullptr)
{
  const int block = 256;                                  // the ring is laid out for four waves
  const int slots = n_chunk > 0 && g_variant != 2 ? 0 : resident_workgroups(ring_step_kernel<S, MINW, ROWS, DIV2, X>, block);
  RingArgs<S> s;
  if (g_variant == 2) s.p = make_pipe_args<S>(g, a0, in, pers_a, pers_b, out, cosines, 2, block, 0, slots, true, 1, n_chunk);
  else                s.p = make_pipe_args<S>(g, a0, in, pers_a, pers_b, out, cosines, 2, block, n_chunk, slots, true, 0, 0, g_variant == 3);
  for (int i = 0; i < S; i++) s.rows[i] = ROWS && rows ? rows[i] : nullptr;
  s.stamps = g_stamps;
  hipLaunchKernelGGL((ring_step_kernel<S, MINW, ROWS, DIV2, X>), dim3(s.p.wg_begin[s.p.n_phase]), dim3(block), 0, launch_stream(), s);
  return (int)hipGetLastError();
}

template <int S, int V, int MINW, bool DIAG = false, bool ROWS = false>
int launch_multi(const slb_multi_geom *g, const float *a0, const float *const *in, const float *const *pers_a,
                 const float *const *pers_b, float *const *out, const float *cosines, int block, int n_chunk,
                 float *rows_first = nullptr, float *rows_second = nullptr)
{
  constexpr int Q = 2 * S, H = (Q + V - 1) / V, INNER = 64 - 2 * H;
  MultiArgs<S> s;
  s.a0 = a0; s.a_c = in[0]; s.b_c = in[1]; s.a_h = in[2]; s.b_h = in[3];
  for (int q = 0; q < Q - 1; q++) { s.pers_a[q] = pers_a[q]; s.pers_b[q] = pers_b[q]; }
  s.a_o = out[0]; s.b_o = out[1]; s.a_ho = out[2]; s.b_ho = out[3];
  s.rows1 = rows_first; s.rows2 = rows_second;
  s.E_dc = g_phys.E_dc; s.E_omega = g_phys.E_omega; s.B = g_phys.B; s.dt = g_phys.dt;
  s.dPhi = g_phys.dPhi; s.PhiYmin = g_phys.PhiYmin; s.nu = g_phys.nu; s.nu2 = g_phys.nu2;
  s.nu_tilde = g_phys.nu_tilde; s.bdt = g_phys.bdt;
  for (int q = 0; q < 2 * Q; q++) s.c[q] = cosines[q];
  s.N = g_phys.N; s.pitch = g->pitch; s.m_offset = g->m_offset;
  int lo = g->j_first[0], hi = g->j_last[0];
  for (int q = 0; q < Q; q++) {
    s.jf[q] = g->j_first[q]; s.jl[q] = g->j_last[q];
    lo = s.jf[q] < lo ? s.jf[q] : lo; hi = s.jl[q] > hi ? s.jl[q] : hi;
  }
  s.jj0 = lo / V * V;
  s.n_chunk = n_chunk > s.N ? s.N : n_chunk;
  const int per_wave = INNER * V;
  const int waves = (hi + 1 - s.jj0 + per_wave - 1) / per_wave;
  const int wpb = block / 64;
  dim3 grid((waves + wpb - 1) / wpb, (s.N + s.n_chunk - 1) / s.n_chunk);
  hipLaunchKernelGGL((multi_step_kernel<S, V, MINW, DIAG, ROWS>), grid, dim3(block), 0, launch_stream(), s);
  return (int)hipGetLastError();
}

}  // namespace

// ---------------------------------------------------------------------------
// C boundary
// ---------------------------------------------------------------------------
extern "C" {

void HandleError(int err, const char *file, int line)
{
  if (err != (int)hipSuccess) {
    printf("%s in %s at line %d\n", hipGetErrorString((hipError_t)err), file, line);
    exit(EXIT_FAILURE);
  }
}

void load_data(void)
{
  g_phys.E_dc = host_E_dc;       g_phys.E_omega = host_E_omega;  g_phys.omega = host_omega;
  g_phys.B = host_B;             g_phys.dt = host_dt;            g_phys.dPhi = host_dPhi;
  g_phys.PhiYmin = PhiYmin;      g_phys.nu = host_nu;            g_phys.nu2 = host_nu2;
  g_phys.nu_tilde = host_nu_tilde; g_phys.bdt = host_bdt;
  g_phys.N = host_N;             g_phys.M = host_M;
  g_phys.pitch = PADDED_MSIZE;   g_phys.TMSIZE = host_TMSIZE;
  g_phys.loaded = true;
}

void slb_set_stream(void *s)           { g_stream = (hipStream_t)s; }
void *slb_get_stream(void)             { return (void *)launch_stream(); }
void slb_push_stream(void *s)
{
  if (t_stream_depth >= kStreamStack) { fprintf(stderr, "slb_push_stream: more than %d nested\n", kStreamStack); exit(EXIT_FAILURE); }
  t_stream[t_stream_depth++] = (hipStream_t)s;
}
void slb_pop_stream(void)
{
  if (t_stream_depth > 0) t_stream_depth--;
}
void slb_set_av_mode(int mode)         { g_av_mode = (mode == SLB_AV_TREE || mode == SLB_AV_CHAIN) ? mode : SLB_AV_EXACT; }
void slb_set_tuning(int v, int nc, int b) { g_variant = v; g_n_chunk = nc; g_block = b; }
#ifdef SLB_TIMING_VARIANTS
void slb_debug_set_stamps(unsigned long long *buf) { g_stamps = buf; }
#endif
int slb_debug_schedule(int kind, int steps, int N, int col_first, int col_last, int block, int n_chunk, int slots,
                       int *info, int *cover, int cover_len)
{
  if (!info || !cover || N < 1 || col_first < 1 || col_last < col_first || block < 64 || block > 256 || block % 64 != 0 ||
      kind < 2 || kind > 5 || (kind != 2 && steps != 2 && steps != 3))
    return -1;
  const int H = 2, per_group = (64 - 2 * H) * 4 * (block / 64);
  int groups = 0, total = 0;
  auto tally = [&](auto &sched, int edge_last, int jj0) -> int {
    groups = (col_last + 1 - jj0 + per_group - 1) / per_group;
    if ((long long)groups * N > cover_len) return -3;
    for (int i = 0; i < groups * N; i++) cover[i] = 0;
    total = sched.wg_begin[sched.n_phase];
    for (int id = 0; id < total; id++) {
      const WgPlace wg = place_workgroup(sched, id, edge_last);
      if (!wg.live) continue;
      if (wg.bx < 0 || wg.bx >= groups || wg.len < 1) return -2;
      for (int n = wg.n0; n < wg.n0 + wg.len && n < N; n++) cover[wg.bx * N + n]++;
    }
    info[0] = groups; info[1] = sched.n_phase; info[2] = per_group; info[3] = jj0;
    return total;
  };
  if (kind == 2) {
    Fused2Args a;
    memset(&a, 0, sizeof(a));
    a.N = N;
    for (int q = 0; q < 4; q++) { a.jf[q] = col_first; a.jl[q] = col_last; }
    schedule_fused2(a, H, block, n_chunk, N >= 128 ? (7 * N + 15) / 16 : 32, slots);
    return tally(a, 0, a.jj0);
  }
  slb_multi_geom g;
  memset(&g, 0, sizeof(g));
  g.pitch = (col_last + 2 + 31) / 32 * 32; g.steps = steps;
  for (int q = 0; q < 2 * steps; q++) { g.j_first[q] = col_first; g.j_last[q] = col_last; }
  const float *in[4] = {nullptr, nullptr, nullptr, nullptr}, *pers[8] = {};
  float *out[4] = {nullptr, nullptr, nullptr, nullptr};
  const float cosines[16] = {};
  const int saved_N = g_phys.N;
  g_phys.N = N;
  int rc;
  if (steps == 2) { PipeArgs<2> a = make_pipe_args<2>(&g, nullptr, in, pers, pers, out, cosines, H, block, n_chunk, slots, kind != 4, 0, 0, kind == 5); rc = tally(a, a.edge_last, a.jj0); }
  else            { PipeArgs<3> a = make_pipe_args<3>(&g, nullptr, in, pers, pers, out, cosines, H, block, n_chunk, slots, kind != 4, 0, 0, kind == 5); rc = tally(a, a.edge_last, a.jj0); }
  g_phys.N = saved_N;
  return rc;
}

const char *slb_version(void)          { return "slb2d-gfx950 0.1"; }

int slb_sweep(const slb_slab *slab,
              const float *a0, const float *a_centre, const float *b_centre,
              const float *a_stencil, const float *b_stencil,
              float *a_out, float *b_out,
              float cos_omega_t, float cos_omega_t_plus_dt)
{
  if (!g_phys.loaded) { fprintf(stderr, "slb_sweep: load_data() has not been called\n"); return -1; }
  if (!slab || slab->pitch < 3 || slab->j_first < 1 || slab->j_last > slab->pitch - 2 ||
      g_phys.N < 1) {
    fprintf(stderr, "slb_sweep: bad geometry (pitch=%d j_first=%d j_last=%d N=%d)\n",
            slab ? slab->pitch : -1, slab ? slab->j_first : -1, slab ? slab->j_last : -1, g_phys.N);
    return -1;
  }
  if (slab->j_last < slab->j_first) return 0;   // empty slab: nothing to update

  SweepArgs s;
  s.a0 = a0; s.a_c = a_centre; s.b_c = b_centre; s.a_s = a_stencil; s.b_s = b_stencil;
  s.a_out = a_out; s.b_out = b_out;
  s.E_dc = g_phys.E_dc; s.E_omega = g_phys.E_omega; s.B = g_phys.B; s.dt = g_phys.dt;
  s.dPhi = g_phys.dPhi; s.PhiYmin = g_phys.PhiYmin; s.nu = g_phys.nu; s.nu2 = g_phys.nu2;
  s.nu_tilde = g_phys.nu_tilde; s.bdt = g_phys.bdt;
  s.cos_t = cos_omega_t; s.cos_t1 = cos_omega_t_plus_dt;
  s.N = g_phys.N; s.pitch = slab->pitch;
  s.j_first = slab->j_first; s.j_last = slab->j_last; s.m_offset = slab->m_offset;

  const bool can_vec = (slab->pitch % 4 == 0) && aligned16(a0) && aligned16(a_centre) &&
                       aligned16(b_centre) && aligned16(a_stencil) && aligned16(b_stencil) &&
                       aligned16(a_out) && aligned16(b_out);
  int variant = g_variant;
  if (variant == 0) variant = can_vec ? 5 : 1;
  if (variant != 1 && !can_vec) {
    fprintf(stderr, "slb_sweep: variant %d needs a row pitch that is a multiple of 4 floats and 16-byte aligned arrays\n", variant);
    return -1;
  }
  const int block = g_block > 0 ? g_block : 256;
  if (block % 64 != 0 || block > 256) { fprintf(stderr, "slb_sweep: block must be 64, 128, 192 or 256\n"); return -1; }

  // rows per workgroup: each chunk re-reads two stencil rows, but chunks of 16 rows measured
  // fastest on MI355X at every large shape tried (waves of different chunks run out of phase),
  // and narrow lattices need the extra workgroups to fill 256 CUs
  int n_chunk = g_n_chunk;
  if (n_chunk <= 0) n_chunk = auto_chunk(s.N, s.j_last - s.j_first + 1, variant == 1 ? 64 : 256, block, 16);
  if (n_chunk > s.N) n_chunk = s.N;
  s.n_chunk = n_chunk;

  switch (variant) {
    case 1:  return launch_sweep<1, 0>(s, block);
    case 2:  return launch_sweep<4, 0>(s, block);
    case 3:  return launch_sweep<4, 1>(s, block);
    case 4:  return launch_pipe<false>(s, block);
    case 5:  return launch_pipe<true>(s, block);
    case 9:  return launch_pipe<false, true>(s, block);   // nontemporal stores only
    default:
      fprintf(stderr, "slb_sweep: unknown variant %d\n", variant);
      return -1;
  }
}

int slb_step_fused(const slb_step_geom *g,
                   const float *a0, const float *a_current, const float *b_current,
                   const float *a_current_hs, const float *b_current_hs,
                   float *a_next, float *b_next, float *a_next_hs, float *b_next_hs,
                   const float cosines[4])
{
  if (!g_phys.loaded) { fprintf(stderr, "slb_step_fused: load_data() has not been called\n"); return -1; }
  if (!g || g->pitch < 4 || g->pitch % 4 != 0 || g->j1_first < 1 || g->j2_first < 1 ||
      g->j1_last > g->pitch - 2 || g->j2_last > g->pitch - 2 || g_phys.N < 1) {
    fprintf(stderr, "slb_step_fused: bad geometry\n");
    return -1;
  }
  // the half-step sweep may only read main-sweep columns that are produced here or are
  // genuinely never written (global boundary columns hold their old value)
  const void *ptrs[9] = {a0, a_current, b_current, a_current_hs, b_current_hs, a_next, b_next, a_next_hs, b_next_hs};
  for (int i = 0; i < 9; i++)
    if (!aligned16(ptrs[i])) { fprintf(stderr, "slb_step_fused: arrays must be 16-byte aligned\n"); return -1; }
  if (g->j1_last < g->j1_first && g->j2_last < g->j2_first) return 0;

  FusedArgs s;
  s.a0 = a0; s.a_c = a_current; s.b_c = b_current; s.a_h = a_current_hs; s.b_h = b_current_hs;
  s.a_n = a_next; s.b_n = b_next; s.a_hn = a_next_hs; s.b_hn = b_next_hs;
  s.E_dc = g_phys.E_dc; s.E_omega = g_phys.E_omega; s.B = g_phys.B; s.dt = g_phys.dt;
  s.dPhi = g_phys.dPhi; s.PhiYmin = g_phys.PhiYmin; s.nu = g_phys.nu; s.nu2 = g_phys.nu2;
  s.nu_tilde = g_phys.nu_tilde; s.bdt = g_phys.bdt;
  s.c0 = cosines[0]; s.c1 = cosines[1]; s.c2 = cosines[2]; s.c3 = cosines[3];
  s.N = g_phys.N; s.pitch = g->pitch; s.m_offset = g->m_offset;
  s.j1_first = g->j1_first; s.j1_last = g->j1_last; s.j2_first = g->j2_first; s.j2_last = g->j2_last;

  const int block = g_block > 0 ? g_block : 256;
  if (block % 64 != 0 || block > 256) { fprintf(stderr, "slb_step_fused: block must be 64, 128, 192 or 256\n"); return -1; }
  // Rows per workgroup.  Each chunk re-forms one main-sweep row above and below itself,
  // so short chunks cost reads; but on MI355X chunks of 16 rows measured fastest at every
  // large shape tried (waves of different chunks run out of phase and keep HBM busy), and
  // narrow lattices need the extra workgroups anyway.
  int n_chunk = g_n_chunk;
  if (n_chunk <= 0) {
    const int lo = s.j1_first < s.j2_first ? s.j1_first : s.j2_first;
    const int hi = s.j1_last > s.j2_last ? s.j1_last : s.j2_last;
    n_chunk = auto_chunk(s.N, hi - lo + 1, 248, block, 16);
  }
  if (n_chunk > s.N) n_chunk = s.N;
  s.n_chunk = n_chunk;
  switch (g_variant) {
    case 6:  return launch_fused<false, 4>(s, block);
    case 7:  return launch_fused<true, 4>(s, block);      // nontemporal accesses for data touched once
    case 8:  return launch_fused<false, 4, false, true>(s, block);   // nontemporal stores only
#ifdef SLB_TIMING_VARIANTS
    case 99: return launch_fused<false, 4, true>(s, block);   // arithmetic removed: memory floor, wrong results
#endif
    default: return launch_fused<false, 4>(s, block);
  }
}

int slb_step_fused2(const slb_step2_geom *g, const float *a0,
                    const float *a_current, const float *b_current,
                    const float *a_current_hs, const float *b_current_hs,
                    const float *a_next, const float *b_next,
                    const float *a_next_hs, const float *b_next_hs,
                    float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                    const float cosines[8])
{
  return slb_step_fused2_rows(g, a0, a_current, b_current, a_current_hs, b_current_hs, a_next, b_next,
                              a_next_hs, b_next_hs, a_out, b_out, a_out_hs, b_out_hs, cosines, nullptr, nullptr);
}

static int step_fused2_impl(const slb_step2_geom *g, const float *a0, const double *a0_row, const double *a0_col,
                            const float *a_current, const float *b_current,
                            const float *a_current_hs, const float *b_current_hs,
                            const float *a_next, const float *b_next,
                            const float *a_next_hs, const float *b_next_hs,
                            float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                            const float cosines[8], float *rows_first, float *rows_second);

int slb_step_fused2_rows(const slb_step2_geom *g, const float *a0,
                         const float *a_current, const float *b_current,
                         const float *a_current_hs, const float *b_current_hs,
                         const float *a_next, const float *b_next,
                         const float *a_next_hs, const float *b_next_hs,
                         float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                         const float cosines[8], float *rows_first, float *rows_second)
{
  return step_fused2_impl(g, a0, nullptr, nullptr, a_current, b_current, a_current_hs, b_current_hs, a_next, b_next,
                          a_next_hs, b_next_hs, a_out, b_out, a_out_hs, b_out_hs, cosines, rows_first, rows_second);
}

int slb_step_fused2_factored(const slb_step2_geom *g, const float *a0, const double *a0_row, const double *a0_col,
                             const float *a_current, const float *b_current,
                             const float *a_current_hs, const float *b_current_hs,
                             const float *a_next, const float *b_next,
                             const float *a_next_hs, const float *b_next_hs,
                             float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                             const float cosines[8], float *rows_first, float *rows_second)
{
  if (!a0_row || !a0_col || !aligned16(a0_col)) {
    fprintf(stderr, "slb_step_fused2_factored: the two factor arrays are needed, the column factors 16-byte aligned\n");
    return -1;
  }
  return step_fused2_impl(g, a0, a0_row, a0_col, a_current, b_current, a_current_hs, b_current_hs, a_next, b_next,
                          a_next_hs, b_next_hs, a_out, b_out, a_out_hs, b_out_hs, cosines, rows_first, rows_second);
}

int slb_selftest_division(unsigned long long count, unsigned seed, unsigned long long *differ, float *first_bad)
{
  if (!differ || !first_bad) return -1;
  hipLaunchKernelGGL(division_selftest_kernel, dim3(4096), dim3(256), 0, launch_stream(), count, (uint32_t)seed, differ, first_bad);
  return (int)hipGetLastError();
}

int slb_a0_factors_check(const float *a0, int pitch, int row_first, int nrows, int ncols,
                         const double *a0_row, const double *a0_col, int *bad)
{
  if (!a0 || !a0_row || !a0_col || !bad || pitch < 1 || row_first < 0 || nrows < 0 || ncols < 0 || ncols > pitch) {
    fprintf(stderr, "slb_a0_factors_check: bad arguments\n");
    return -1;
  }
  if (nrows == 0 || ncols == 0) return 0;
  int bx = (ncols + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(a0_factors_check_kernel, dim3(bx, nrows), dim3(256), 0, launch_stream(), a0, pitch, row_first, nrows,
                     ncols, a0_row, a0_col, bad);
  return (int)hipGetLastError();
}

static int step_fused2_impl(const slb_step2_geom *g, const float *a0, const double *a0_row, const double *a0_col,
                            const float *a_current, const float *b_current,
                            const float *a_current_hs, const float *b_current_hs,
                            const float *a_next, const float *b_next,
                            const float *a_next_hs, const float *b_next_hs,
                            float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                            const float cosines[8], float *rows_first, float *rows_second)
{
  if ((rows_first == nullptr) != (rows_second == nullptr)) {
    fprintf(stderr, "slb_step_fused2_rows: both row buffers or neither\n");
    return -1;
  }
  if (!g_phys.loaded) { fprintf(stderr, "slb_step_fused2: load_data() has not been called\n"); return -1; }
  if (!g || g->pitch < 4 || g->pitch % 4 != 0 || g_phys.N < 1) { fprintf(stderr, "slb_step_fused2: bad geometry\n"); return -1; }
  for (int q = 0; q < 4; q++) {
    if (g->j_first[q] < 1 || g->j_last[q] > g->pitch - 2) { fprintf(stderr, "slb_step_fused2: bad column range\n"); return -1; }
    // every stage must be able to take its m+-1 neighbours from the stage before it or from
    // never-written cells: ranges may only shrink by at most... they are the caller's contract
  }
  const void *ptrs[13] = {a0, a_current, b_current, a_current_hs, b_current_hs, a_next, b_next,
                          a_next_hs, b_next_hs, a_out, b_out, a_out_hs, b_out_hs};
  for (int i = 0; i < 13; i++)
    if (!aligned16(ptrs[i])) { fprintf(stderr, "slb_step_fused2: arrays must be 16-byte aligned\n"); return -1; }
  if (a_out == a_current || b_out == b_current || a_out_hs == a_current_hs || b_out_hs == b_current_hs) {
    fprintf(stderr, "slb_step_fused2: outputs must not alias the inputs\n");
    return -1;
  }
  if (g->j_last[3] < g->j_first[3] && g->j_last[2] < g->j_first[2]) return 0;

  Fused2Args s;
  s.a0 = a0; s.a_c = a_current; s.b_c = b_current; s.a_h = a_current_hs; s.b_h = b_current_hs;
  s.p1a = a_next; s.p1b = b_next; s.p2a = a_next_hs; s.p2b = b_next_hs;
  s.a_o = a_out; s.b_o = b_out; s.a_ho = a_out_hs; s.b_ho = b_out_hs;
  s.rows1 = rows_first; s.rows2 = rows_second;
  s.a0_row = a0_row; s.a0_col = a0_col;
  s.E_dc = g_phys.E_dc; s.E_omega = g_phys.E_omega; s.B = g_phys.B; s.dt = g_phys.dt;
  s.dPhi = g_phys.dPhi; s.PhiYmin = g_phys.PhiYmin; s.nu = g_phys.nu; s.nu2 = g_phys.nu2;
  s.nu_tilde = g_phys.nu_tilde; s.bdt = g_phys.bdt;
  for (int q = 0; q < 8; q++) s.c[q] = cosines[q];
  s.N = g_phys.N; s.pitch = g->pitch; s.m_offset = g->m_offset;
  for (int q = 0; q < 4; q++) { s.jf[q] = g->j_first[q]; s.jl[q] = g->j_last[q]; }

  const int block = g_block > 0 ? g_block : 256;
  if (block % 64 != 0 || block > 256) { fprintf(stderr, "slb_step_fused2: block must be 64, 128, 192 or 256\n"); return -1; }
  // Each chunk of rows re-forms three rows of stage A above and below itself (two of B, one
  // of C) and re-reads their inputs, so chunks are a cost; they are there to fill the device.
  int lo = s.jf[0], hi = s.jl[0];
  for (int q = 1; q < 4; q++) { lo = s.jf[q] < lo ? s.jf[q] : lo; hi = s.jl[q] > hi ? s.jl[q] : hi; }
  // Rows per workgroup where strip groups have to be cut (see launch_fused2): 7/16 of the rows,
  // so that two long chunks are followed by a short one (56, 56, 16 at 128 rows).  Workgroups
  // are dispatched chunk after chunk; the short ones at the end fill the compute units that
  // equal chunks leave idle while the last of them finish (measured at 2^20 columns with every
  // group cut: 1.113 ms against 1.146 ms for four chunks of 32; 224 rows at 512, 112 at 256
  // likewise).  Lattices narrower than about a thousand workgroups get shorter chunks.
  int cap = 32;
  if (s.N >= 128) cap = (7 * s.N + 15) / 16;
  // (two rounds of workgroups rather than one: at n-harmonics 64, g-grid 2^18 chunks of 11-12
  // rows take 0.176-0.178 ms per launch, 16 rows 0.183, 22 rows 0.191, 32 rows 0.204)
  int tail_chunk = auto_chunk(s.N, hi - lo + 1, 240, block, cap, 1536);
  // A launch too narrow to fill the device is a matter of how long one wave takes per pass of
  // its row loop, and that is set by the instructions of a pass: the general kernel with ONE
  // column per lane has a quarter of them (and four times the waves).  Measured on MI355X:
  // 14.2 against 23.7 us per double step at 32 x 4096, 16.4 against 23.8 at 16 x 20000; from
  // about 500 wave-chunks of the four-column kernel on it is slower (0.119 against 0.109 ms at
  // 128 x 65536, 0.207 against 0.172 at 64 x 2^18).  Same stages, same point update, same bits.
  // (Tuning variant 38 keeps the four-column kernel.)
  if (g_variant == 0 && g_n_chunk <= 0) {
    const int waves4 = (hi - lo + 1 + 239) / 240;
    const int chunks4 = (s.N + tail_chunk - 1) / tail_chunk;
    if (waves4 * chunks4 <= 512) {
      slb_multi_geom mg;
      mg.pitch = g->pitch; mg.m_offset = g->m_offset; mg.steps = 2;
      for (int q = 0; q < 4; q++) { mg.j_first[q] = g->j_first[q]; mg.j_last[q] = g->j_last[q]; }
      const float *in[4] = {a_current, b_current, a_current_hs, b_current_hs};
      const float *pa[3] = {a_next, a_next_hs, a_out}, *pb[3] = {b_next, b_next_hs, b_out};
      float *out[4] = {a_out, b_out, a_out_hs, b_out_hs};
      const int nc = auto_chunk(s.N, hi - lo + 1, 56, 64, 8, 2048);
      if (rows_first) return launch_multi<2, 1, 4, false, true>(&mg, a0, in, pa, pb, out, cosines, 64, nc, rows_first, rows_second);
      return launch_multi<2, 1, 4>(&mg, a0, in, pa, pb, out, cosines, 64, nc);
    }
  }
  // tuning variant 26: every strip group cut into chunks, as for a narrow lattice
  int n_chunk = g_n_chunk > 0 ? g_n_chunk : (g_variant == 26 || s.N < 64 ? tail_chunk : 0);
  // a0 formed from its factors instead of loaded: tuning variants 50 (a row's values formed once
  // and handed from stage to stage) and 51 (every stage forms its own row's).  Not the default:
  // measured on MI355X at 128 x 2^20 the launch takes 1.19-1.21 ms with a0 formed against 1.02-1.04
  // with a0 loaded, although it moves 32 instead of 36 bytes per cell -- and 1.08 ms with a0
  // neither loaded nor formed (a timing probe), so it is the schedule of the row loop that the
  // change upsets, not the double-precision products (profiles/r03/a0_factored_ab.txt).
  if (a0_row && (g_variant == 50 || g_variant == 51 || g_variant == 53 || g_variant == 54)) {
    if (g_variant == 51)
      return rows_first ? launch_fused2<3, false, true, 2, true, false, 2>(s, block, n_chunk, tail_chunk)
                        : launch_fused2<3, false, false, 2, true, false, 2>(s, block, n_chunk, tail_chunk);
#ifdef SLB_TIMING_VARIANTS
    if (g_variant == 53) return launch_fused2<3, false, false, 2, true, false, 3>(s, block, n_chunk, tail_chunk);
    if (g_variant == 54) return launch_fused2<3, false, false, 2, true, false, 4>(s, block, n_chunk, tail_chunk);
#endif
    return rows_first ? launch_fused2<3, false, true, 2, true, false, 1>(s, block, n_chunk, tail_chunk)
                      : launch_fused2<3, false, false, 2, true, false, 1>(s, block, n_chunk, tail_chunk);
  }
  // tuning variants 60 / 61: the point update with fewer instructions (61: and the shared division)
  if (g_variant == 60)
    return rows_first ? launch_fused2<3, false, true, 2, true, false, 0, 1>(s, block, n_chunk, tail_chunk)
                      : launch_fused2<3, false, false, 2, true, false, 0, 1>(s, block, n_chunk, tail_chunk);
  if (g_variant == 61)
    return rows_first ? launch_fused2<3, false, true, 2, true, false, 0, 3>(s, block, n_chunk, tail_chunk)
                      : launch_fused2<3, false, false, 2, true, false, 0, 3>(s, block, n_chunk, tail_chunk);
  if (g_variant == 62)
    return rows_first ? launch_fused2<3, false, true, 2, true, false, 0, 5>(s, block, n_chunk, tail_chunk)
                      : launch_fused2<3, false, false, 2, true, false, 0, 5>(s, block, n_chunk, tail_chunk);
  if (g_variant == 63)
    return rows_first ? launch_fused2<3, false, true, 2, true, false, 0, 7>(s, block, n_chunk, tail_chunk)
                      : launch_fused2<3, false, false, 2, true, false, 0, 7>(s, block, n_chunk, tail_chunk);
  // default: two halo lanes per side (the 60 interior lanes' stores start on 64-byte boundaries),
  // non-temporal stores, registers for 3 waves per SIMD, the point update two columns wide with one
  // reciprocal for the two quotients of a point where the denominators allow it (measured on MI355X:
  // 1.004-1.009 against 1.014-1.042 ms at 128 x 2^20, 0.150 against 0.174-0.184 ms at 64 x 2^18;
  // tuning variant 64 = the compiler's own point update).  The rest are tuning variants.
  const bool fast_point = g_variant != 64 && shared_division_ok(g->m_offset, g->pitch);
  if (rows_first) return fast_point ? launch_fused2<3, false, true, 2, true, false, 0, 7>(s, block, n_chunk, tail_chunk)
                                    : launch_fused2<3, false, true>(s, block, n_chunk, tail_chunk);
  if (fast_point && (g_variant == 0 || g_variant == 26 || g_variant == 34 || g_variant == 38 || g_variant == 39))
    return launch_fused2<3, false, false, 2, true, false, 0, 7>(s, block, n_chunk, tail_chunk);
  switch (g_variant) {
    case 20: return launch_fused2<3, false, false, 1, false>(s, block, n_chunk, tail_chunk);  // one halo lane, plain stores
    case 21: return launch_fused2<2, false, false, 1, false>(s, block, n_chunk, tail_chunk);  // register budget of 2, 4 waves per SIMD
    case 25: return launch_fused2<4, false, false, 1, false>(s, block, n_chunk, tail_chunk);
    case 27: return launch_fused2<3, false, false, 1, true>(s, block, n_chunk, tail_chunk);   // one halo lane
    case 29: return launch_fused2<3, false, false, 2, false>(s, block, n_chunk, tail_chunk);  // plain stores
    case 30: return launch_fused2<3, false, false, 4, false>(s, block, n_chunk, tail_chunk);  // four halo lanes, plain stores
    case 23: return launch_fused2<3, false, false, 4, true>(s, block, n_chunk, tail_chunk);   // four halo lanes
    case 24: return launch_fused2<3, false, false, 2, true, true>(s, block, n_chunk, tail_chunk);   // non-temporal loads too
#ifdef SLB_TIMING_VARIANTS
    case 98: return launch_fused2<3, true>(s, block, n_chunk, tail_chunk);   // arithmetic removed: memory floor, wrong results
#endif
    default: return launch_fused2<3>(s, block, n_chunk, tail_chunk);
  }
}

static int step_multi_impl(const slb_multi_geom *g, const float *a0, const float *const in[4],
                           const float *const pers_a[], const float *const pers_b[], float *const out[4],
                           const float cosines[], float *const rows[]);

int slb_step_multi(const slb_multi_geom *g, const float *a0, const float *const in[4],
                   const float *const pers_a[], const float *const pers_b[], float *const out[4],
                   const float cosines[])
{
  return step_multi_impl(g, a0, in, pers_a, pers_b, out, cosines, nullptr);
}

int slb_step_multi_rows(const slb_multi_geom *g, const float *a0, const float *const in[4],
                        const float *const pers_a[], const float *const pers_b[], float *const out[4],
                        const float cosines[], float *const rows[])
{
  if (!g || !rows) { fprintf(stderr, "slb_step_multi_rows: one row buffer per iteration is needed\n"); return -1; }
  if (g->steps < 1 || g->steps > 3) { fprintf(stderr, "slb_step_multi_rows: 1 to 3 steps per launch\n"); return -1; }
  for (int i = 0; i < g->steps; i++)
    if (!rows[i]) { fprintf(stderr, "slb_step_multi_rows: one row buffer per iteration is needed\n"); return -1; }
  return step_multi_impl(g, a0, in, pers_a, pers_b, out, cosines, rows);
}

static int step_multi_impl(const slb_multi_geom *g, const float *a0, const float *const in[4],
                           const float *const pers_a[], const float *const pers_b[], float *const out[4],
                           const float cosines[], float *const rows[])
{
  if (!g_phys.loaded) { fprintf(stderr, "slb_step_multi: load_data() has not been called\n"); return -1; }
  if (!g || g->steps < 1 || g->steps > 4 || g->pitch < 4 || g->pitch % 4 != 0 || g_phys.N < 1) {
    fprintf(stderr, "slb_step_multi: bad geometry (1 to 4 steps per launch)\n");
    return -1;
  }
  const int Q = 2 * g->steps;
  for (int q = 0; q < Q; q++)
    if (g->j_first[q] < 1 || g->j_last[q] > g->pitch - 2) { fprintf(stderr, "slb_step_multi: bad column range\n"); return -1; }
  for (int i = 0; i < 4; i++) {
    if (!aligned16(in[i]) || !aligned16(out[i])) { fprintf(stderr, "slb_step_multi: arrays must be 16-byte aligned\n"); return -1; }
    for (int k = 0; k < 4; k++)
      if (out[i] == in[k]) { fprintf(stderr, "slb_step_multi: outputs must not alias the inputs\n"); return -1; }
  }
  for (int q = 0; q < Q - 1; q++)
    if (!aligned16(pers_a[q]) || !aligned16(pers_b[q])) { fprintf(stderr, "slb_step_multi: arrays must be 16-byte aligned\n"); return -1; }
  if (!aligned16(a0)) return -1;
  if (g->j_last[Q - 1] < g->j_first[Q - 1] && g->j_last[Q - 2] < g->j_first[Q - 2]) return 0;
  const int block = g_block > 0 ? g_block : 256;
  if (block % 64 != 0 || block > 256) { fprintf(stderr, "slb_step_multi: block must be 64, 128, 192 or 256\n"); return -1; }
  const int n_chunk = g_n_chunk > 0 ? g_n_chunk : 64;
  // waves per SIMD the register allocator is held to: measured best per S on MI355X.
  // tuning variants 32..34: two columns per lane (half the registers, twice the waves)
  // with a budget of 2, 3, 4 waves per SIMD
  const int v = g_variant;
  if (v >= 32 && v <= 34) {
    switch (g->steps * 10 + (v - 32)) {
      case 20: return launch_multi<2, 2, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 21: return launch_multi<2, 2, 3>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 22: return launch_multi<2, 2, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 30: return launch_multi<3, 2, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 31: return launch_multi<3, 2, 3>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 32: return launch_multi<3, 2, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 40: return launch_multi<4, 2, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 41: return launch_multi<4, 2, 3>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 42: return launch_multi<4, 2, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      default: break;
    }
  }
  // tuning variants 70..75: the scheduled kernel (pipe_step_kernel), four columns per lane, two halo
  // lanes per side, non-temporal stores, packed point update with the shared division:
  //   70 / 71: S as given, register budget of 3 / 2 waves per SIMD;  72 / 73: the same with stage 0's
  //   loads a pass ahead;  74 / 75: 71 / 73 with the compiler's own division
  if (v >= 70 && v <= 75 && (g->steps == 2 || g->steps == 3)) {
    const int nc = g_n_chunk;
    switch (g->steps * 100 + v) {
      case 270: return launch_pipe_sched<2, 3, 2, true, true, false>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 271: return launch_pipe_sched<2, 2, 2, true, true, false>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 272: return launch_pipe_sched<2, 3, 2, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 273: return launch_pipe_sched<2, 2, 2, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 370: return launch_pipe_sched<3, 3, 2, true, true, false>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 371: return launch_pipe_sched<3, 2, 2, true, true, false>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 372: return launch_pipe_sched<3, 3, 2, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 373: return launch_pipe_sched<3, 2, 2, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 374: return launch_pipe_sched<3, 2, 2, true, false, false>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 375: return launch_pipe_sched<3, 2, 2, true, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      default: break;
    }
  }
  // tuning variants 80..85: the same kernel with the stage outputs kept in an LDS ring per wave:
  //   80 / 81: register budget of 3 / 2 waves per SIMD;  82 / 83: the same with stage 0's loads a
  //   pass ahead;  84: budget of 4 waves per SIMD (S = 2)
  if (v >= 80 && v <= 85 && (g->steps == 2 || g->steps == 3)) {
    const int nc = g_n_chunk;
    if (block != 256) { fprintf(stderr, "slb_step_multi: the LDS ring is laid out for 256 threads\n"); return -1; }
    switch (g->steps * 100 + v) {
      case 280: return launch_pipe_sched<2, 3, 2, true, true, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 281: return launch_pipe_sched<2, 2, 2, true, true, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 282: return launch_pipe_sched<2, 3, 2, true, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 284: return launch_pipe_sched<2, 4, 2, true, true, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 380: return launch_pipe_sched<3, 3, 2, true, true, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 381: return launch_pipe_sched<3, 2, 2, true, true, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 382: return launch_pipe_sched<3, 3, 2, true, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      case 383: return launch_pipe_sched<3, 2, 2, true, true, true, true>(g, a0, in, pers_a, pers_b, out, cosines, block, nc);
      default: break;
    }
  }
  // tuning variants 90 / 91: the ring kernel with the branch-free steady passes (ring_step_kernel),
  // register budget of 2 / 3 waves per SIMD
#ifdef SLB_TIMING_VARIANTS
  // timing builds: 92 = the rows of a pass requested in that pass, 93 = no stores, 95 = time stamps per wave,
  // 96 = a strip's last passes left out, 94 = those passes in the branch-free body (wrong results in the last rows)
  if (v == 92 && g->steps == 3) return launch_ring<3, 2, false, true, 1>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
  if (v == 93 && g->steps == 3) return launch_ring<3, 2, false, true, 2>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
  if (v == 95 && g->steps == 3) return launch_ring<3, 2, false, true, 4>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
  if (v == 96 && g->steps == 3) return launch_ring<3, 2, false, true, 8>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
  if (v == 94 && g->steps == 3) return launch_ring<3, 2, false, true, 16>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
#endif
  if ((v == 90 || v == 91) && (g->steps == 2 || g->steps == 3)) {
    if (block != 256) { fprintf(stderr, "slb_step_multi: the LDS ring is laid out for 256 threads\n"); return -1; }
    const int nc = g_n_chunk;
    switch (g->steps * 100 + v) {
      case 290: return launch_ring<2, 2>(g, a0, in, pers_a, pers_b, out, cosines, nc);
      case 291: return launch_ring<2, 3>(g, a0, in, pers_a, pers_b, out, cosines, nc);
      case 390: return launch_ring<3, 2>(g, a0, in, pers_a, pers_b, out, cosines, nc);
      default: break;
    }
  }
  // tuning variants 35..37: one column per lane, register budget of 2, 4, 8 waves per SIMD
  if (v >= 35 && v <= 37) {
    switch (g->steps * 10 + (v - 35)) {
      case 20: return launch_multi<2, 1, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 21: return launch_multi<2, 1, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 22: return launch_multi<2, 1, 8>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 30: return launch_multi<3, 1, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 31: return launch_multi<3, 1, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 40: return launch_multi<4, 1, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      case 41: return launch_multi<4, 1, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
      default: break;
    }
  }
  // Three steps per launch: the LDS-ring kernel (ring_step_kernel), the fastest form on MI355X from about
  // one round of workgroups up (tuning variant 1 keeps the register-history kernel, multi_step_kernel).
  // With rows, the ring kernel is the only three-step form that hands them out.
  if (g->steps == 3 && block == 256 && (v == 0 || v == 2 || v == 3 || rows)) {
    const int nc = g_n_chunk;
    const bool d2 = shared_division_ok(g->m_offset, g->pitch);
    if (rows) return d2 ? launch_ring<3, 2, true, true>(g, a0, in, pers_a, pers_b, out, cosines, nc, rows)
                        : launch_ring<3, 2, true, false>(g, a0, in, pers_a, pers_b, out, cosines, nc, rows);
    return d2 ? launch_ring<3, 2, false, true>(g, a0, in, pers_a, pers_b, out, cosines, nc)
              : launch_ring<3, 2, false, false>(g, a0, in, pers_a, pers_b, out, cosines, nc);
  }
  if (rows) {
    if (g->steps == 1) return launch_multi<1, 4, 4, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk, nullptr, rows[0]);
    if (g->steps == 2) return launch_multi<2, 4, 3, false, true>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk, rows[0], rows[1]);
    fprintf(stderr, "slb_step_multi_rows: three steps per launch hand out rows with 256-thread workgroups only\n");
    return -1;
  }
  switch (g->steps) {
    case 1: return launch_multi<1, 4, 4>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
    case 2: return launch_multi<2, 4, 3>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
    case 3: return launch_multi<3, 4, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
    default: return launch_multi<4, 4, 2>(g, a0, in, pers_a, pers_b, out, cosines, block, n_chunk);
  }
}

void step_on_grid(int blocks, float *a0, float *a_current, float *b_current,
                  float *a_next, float *b_next,
                  float *a_current_hs, float *b_current_hs,
                  float t, float t_hs, float cos_omega_t, float cos_omega_t_plus_dt)
{
  (void)blocks; (void)t; (void)t_hs;
  const slb_slab slab = {g_phys.pitch, 0, 1, g_phys.TMSIZE};
  int rc = slb_sweep(&slab, a0, a_current, b_current, a_current_hs, b_current_hs,
                     a_next, b_next, cos_omega_t, cos_omega_t_plus_dt);
  if (rc > 0) HandleError(rc, __FILE__, __LINE__);
  if (rc < 0) exit(EXIT_FAILURE);
}

void step_on_half_grid(int blocks, float *a0, float *a_current, float *b_current,
                       float *a_next, float *b_next,
                       float *a_current_hs, float *b_current_hs,
                       float *a_next_hs, float *b_next_hs,
                       float t, float t_hs, float cos_omega_t, float cos_omega_t_plus_dt)
{
  (void)blocks; (void)t; (void)t_hs; (void)a_current; (void)b_current;
  const slb_slab slab = {g_phys.pitch, 0, 1, g_phys.TMSIZE - 1};
  int rc = slb_sweep(&slab, a0, a_current_hs, b_current_hs, a_next, b_next,
                     a_next_hs, b_next_hs, cos_omega_t, cos_omega_t_plus_dt);
  if (rc > 0) HandleError(rc, __FILE__, __LINE__);
  if (rc < 0) exit(EXIT_FAILURE);
}

int slb_halo_pack(const float *a, const float *b, int pitch, int nrows, int col, int ncols, float *buf)
{
  if (nrows < 1 || ncols < 1 || col < 0 || col + ncols > pitch) return -1;
  const int n = 2 * ncols * nrows;
  hipLaunchKernelGGL(halo_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, launch_stream(), a, b, pitch, nrows, col, ncols, buf);
  return (int)hipGetLastError();
}

int slb_halo_unpack(float *a, float *b, int pitch, int nrows, int col, int ncols, const float *buf)
{
  if (nrows < 1 || ncols < 1 || col < 0 || col + ncols > pitch) return -1;
  const int n = 2 * ncols * nrows;
  hipLaunchKernelGGL(halo_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, launch_stream(), a, b, pitch, nrows, col, ncols, buf);
  return (int)hipGetLastError();
}

static int halo_segs(const slb_halo_seg *segs, int nsegs, int pitch, int nrows, bool pack)
{
  if (!segs || nsegs < 1 || nsegs > 4 || nrows < 1 || pitch < 1) return -1;
  HaloSegs h;
  h.nsegs = nsegs; h.pitch = pitch; h.nrows = nrows; h.start[0] = 0;
  for (int k = 0; k < nsegs; k++) {
    if (segs[k].ncols < 1 || segs[k].col < 0 || segs[k].col + segs[k].ncols > pitch ||
        !segs[k].a || !segs[k].b || !segs[k].buf) return -1;
    h.a[k] = segs[k].a; h.b[k] = segs[k].b; h.buf[k] = segs[k].buf;
    h.col[k] = segs[k].col; h.ncols[k] = segs[k].ncols;
    h.start[k + 1] = h.start[k] + 2 * segs[k].ncols * nrows;
  }
  const int n = h.start[nsegs];
  if (pack) hipLaunchKernelGGL(halo_segs_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, launch_stream(), h);
  else      hipLaunchKernelGGL(halo_segs_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, launch_stream(), h);
  return (int)hipGetLastError();
}

int slb_halo_pack_segs(const slb_halo_seg *segs, int nsegs, int pitch, int nrows)
{
  return halo_segs(segs, nsegs, pitch, nrows, true);
}

int slb_halo_unpack_segs(const slb_halo_seg *segs, int nsegs, int pitch, int nrows)
{
  return halo_segs(segs, nsegs, pitch, nrows, false);
}

int slb_frame_values(const float *a, const float *b, int pitch, int nrows, int col_first, int ncols,
                     const double *cs, int nphi, float *out)
{
  if (nrows < 1 || pitch < 1 || col_first < 0 || ncols < 0 || col_first + ncols > pitch || nphi < 0) {
    fprintf(stderr, "slb_frame_values: bad geometry\n");
    return -1;
  }
  if (ncols == 0 || nphi == 0) return 0;
  if (!aligned16(cs)) { fprintf(stderr, "slb_frame_values: table must be 16-byte aligned\n"); return -1; }
  constexpr int P = 8;
  dim3 grid((ncols + 255) / 256, (nphi + P - 1) / P);
  if (b) hipLaunchKernelGGL((frame_kernel<P, true>), grid, dim3(256), 0, launch_stream(), a, b, pitch, nrows, col_first,
                            ncols, reinterpret_cast<const double2 *>(cs), nphi, out);
  else   hipLaunchKernelGGL((frame_kernel<P, false>), grid, dim3(256), 0, launch_stream(), a, b, pitch, nrows, col_first,
                            ncols, reinterpret_cast<const double2 *>(cs), nphi, out);
  return (int)hipGetLastError();
}

int slb_seq_sum_stats(const float *x, int n, float *out, int *stats)
{
  if (n < 0 || (n > 0 && !x) || !out) return -1;
  // slb_set_tuning variants 41..47 select other workgroup shapes (waves x terms per thread)
#define SLB_SEQ(W, T) hipLaunchKernelGGL((seq_sum_kernel<W, T>), dim3(1), dim3(64 * W), 0, launch_stream(), x, n, out, stats)
  switch (g_variant) {
    case 41: SLB_SEQ(4, 4); break;
    case 42: SLB_SEQ(8, 4); break;
    case 43: SLB_SEQ(16, 2); break;
    case 44: SLB_SEQ(4, 8); break;
    case 45: SLB_SEQ(16, 4); break;
    case 46: SLB_SEQ(1, 16); break;
    case 47: SLB_SEQ(2, 8); break;
    default: SLB_SEQ(kAvScanWaves, kAvScanPerLane); break;
  }
#undef SLB_SEQ
  return (int)hipGetLastError();
}

int slb_seq_sum(const float *x, int n, float *out) { return slb_seq_sum_stats(x, n, out, nullptr); }

static AvArgs make_av_args(const slb_slab *slab, const float *a, const float *b)
{
  AvArgs p;
  p.a = a; p.b = b; p.pitch = slab->pitch; p.m_offset = slab->m_offset;
  p.j_first = slab->j_first; p.j_last = slab->j_last;
  p.dPhi = g_phys.dPhi; p.PhiYmin = g_phys.PhiYmin;
  return p;
}

// tree sums of a slab with the block partials in ws (kMaxAvBlocks * 3 doubles)
static int av_partial_ws(const slb_slab *slab, const float *a, const float *b, double *ws, double *sums)
{
  const AvArgs p = make_av_args(slab, a, b);
  const int cols = slab->j_last - slab->j_first + 1;
  int nb = cols > 0 ? (cols + kAvColsPerBlock - 1) / kAvColsPerBlock : 1;
  if (nb > kMaxAvBlocks) nb = kMaxAvBlocks;
  hipLaunchKernelGGL(av_partial_kernel, dim3(nb), dim3(256), 0, launch_stream(), p, ws);
  hipLaunchKernelGGL(av_combine_kernel, dim3(1), dim3(64), 0, launch_stream(), (const double *)ws, nb, sums);
  return (int)hipGetLastError();
}

int slb_av_partial(const slb_slab *slab, const float *a, const float *b, double *sums)
{
  if (!g_phys.loaded || !slab || slab->j_first < 0 || slab->j_last >= slab->pitch) return -1;
  int rc = ensure_workspace();
  if (rc) return rc;
  return av_partial_ws(slab, a, b, g_ws, sums);
}

int slb_av_update(const double *sums, float *av_data, float t)
{
  if (!g_phys.loaded) return -1;
  const float wt = g_phys.omega * t;            // float product, as the reference forms it
  hipLaunchKernelGGL(av_update_kernel, dim3(1), dim3(64), 0, launch_stream(), sums, av_data,
                     cos((double)wt), sin((double)wt), g_phys.dt);
  return (int)hipGetLastError();
}

int slb_av_sums_exact_from_n(const slb_slab *slab, int n, const float *const a[], const float *const b[],
                             const float *const init3[], float *const tot3[])
{
  if (!g_phys.loaded || !slab || slab->j_first < 0 || slab->j_last >= slab->pitch ||
      slab->j_last < slab->j_first || n < 1 || n > kAvMaxSets || !a || !b || !tot3) return -1;
  AvSets sets;
  sets.geom = make_av_args(slab, nullptr, nullptr);
  for (int k = 0; k < kAvMaxSets; k++) {
    const int src = k < n ? k : 0;
    if (!a[src] || !b[src] || !tot3[src]) return -1;
    sets.a[k] = a[src]; sets.b[k] = b[src]; sets.tot[k] = tot3[src];
    sets.init[k] = init3 ? init3[src] : nullptr;
  }
  // both kernels give the CPU's bits; below a few ten thousand columns the one-lane chain is
  // done before the parallel form has paid for its passes
  if (g_av_mode == SLB_AV_CHAIN || slab->j_last - slab->j_first < 32768)
    hipLaunchKernelGGL(av_exact_kernel, dim3(3 * n), dim3(64), 0, launch_stream(), sets);
  else
    hipLaunchKernelGGL(av_exact_scan_kernel, dim3(3 * n), dim3(64 * kAvScanWaves), 0, launch_stream(), sets);
  return (int)hipGetLastError();
}

int slb_av_sums_exact_n(const slb_slab *slab, int n, const float *const a[], const float *const b[],
                        float *const tot3[])
{
  return slb_av_sums_exact_from_n(slab, n, a, b, nullptr, tot3);
}

int slb_av_sums_exact(const slb_slab *slab, const float *a, const float *b, float *tot3)
{
  return slb_av_sums_exact_n(slab, 1, &a, &b, &tot3);
}

int slb_av_update_exact(const float *tot3, float *av_data, float t)
{
  if (!g_phys.loaded || !tot3 || !av_data) return -1;
  const float wt = g_phys.omega * t;            // float product, as the reference forms it
  hipLaunchKernelGGL(av_update_exact_kernel, dim3(1), dim3(64), 0, launch_stream(), tot3, av_data,
                     cos((double)wt), sin((double)wt), g_phys.dt);
  return (int)hipGetLastError();
}

// av() with its workspace given: ws[0 .. 3*kMaxAvBlocks) block partials (tree sums),
// then 4 doubles for the totals handed from the sums to the update
static int av_with_workspace(double *ws, const float *a, const float *b, float *av_data, float t)
{
  // columns 1..M, i.e. m < TMSIZE (reference: src/boltzmann_c_solver.c:417)
  const slb_slab slab = {g_phys.pitch, 0, 1, g_phys.TMSIZE - 1};
  int rc;
  if (g_av_mode == SLB_AV_TREE) {
    rc = av_partial_ws(&slab, a, b, ws, ws + kMaxAvBlocks * 3);
    if (rc == 0) rc = slb_av_update(ws + kMaxAvBlocks * 3, av_data, t);
  } else {
    float *tot = reinterpret_cast<float *>(ws + kMaxAvBlocks * 3);
    rc = slb_av_sums_exact(&slab, a, b, tot);
    if (rc == 0) rc = slb_av_update_exact(tot, av_data, t);
  }
  return rc;
}

void av(int blocks, float *a, float *b, float *av_data, float t)
{
  (void)blocks;
  int rc = g_phys.loaded ? ensure_workspace() : -1;
  if (rc == 0) rc = av_with_workspace(g_ws, a, b, av_data, t);
  if (rc > 0) HandleError(rc, __FILE__, __LINE__);
  if (rc < 0) exit(EXIT_FAILURE);
}

int slb_static_init(void)
{
  double *ws = nullptr;
  return static_workspace(&ws);
}

int slb_av_static(float *a, float *b, float *av_data, float t)
{
  if (!g_phys.loaded) { fprintf(stderr, "slb_av_static: load_data() has not been called\n"); return -1; }
  if (!a || !b || !av_data) { fprintf(stderr, "slb_av_static: null array\n"); return -1; }
  // No limit on g-grid: the workspace does not grow with it.  The tree sums use at most
  // kMaxAvBlocks blocks whose threads stride over the columns, the exact sums keep no per-block
  // partials at all; either way the kWsDoubles entries of the static array are all that is used.
  double *ws = nullptr;
  int rc = static_workspace(&ws);
  if (rc == 0) rc = av_with_workspace(ws, a, b, av_data, t);
  return rc;
}

}  // extern "C"
