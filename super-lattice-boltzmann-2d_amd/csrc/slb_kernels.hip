// slb_kernels.hip -- gfx950 (MI355X) kernels and the C boundary of libslb2d.so.
//
// What runs here: the two lattice sweeps of the Boltzmann stepping loop (one per
// launch, both of a time step in one launch, two or more time steps in one launch),
// the harmonic reductions (exact-order float sums evaluated in parallel, or a tree),
// the display=8 frame sums and the halo movers for phi_y slabs, behind the entry
// points declared in include/slb_abi.h.
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

// ---------------------------------------------------------------------------
// the point update
// ---------------------------------------------------------------------------
// Ap/Am (Bp/Bm): stencil arrays at row n+1, columns m+1 / m-1.
// dA_lo/dB_lo:   A[n-1][m+1]-A[n-1][m-1] and the same for B (any value if n==0).
template <class Params>
__device__ __forceinline__ void point_update(
    const Params &p, int n, float fn, float mu_part, float mu1_part,
    float a0v, float ac, float bc, float Ap, float Am, float Bp, float Bm,
    float dA_lo, float dB_lo, float &a_new, float &b_new)
{
  const float mu_t  = fn * mu_part;
  const float mu_t1 = fn * mu1_part;
  // g: ((dt*a0 + a*nu~) - b*mu_t) + bdt*((Bp - Bm) - [n>=2] dB_lo)
  const float sb = (Bp - Bm) - (n < 2 ? 0.0f : dB_lo);
  const float g  = ((p.dt * a0v + ac * p.nu_tilde) - bc * mu_t) + p.bdt * sb;
  // h: (b*nu~ + a*mu_t) + bdt*(((chi*[n>=1] dA_lo) - Ap) + Am), chi = 2 at n==1
  const float chi = (n == 1) ? 2.0f : 1.0f;
  const float lo  = (n == 0) ? 0.0f : dA_lo;
  const float sa  = ((chi * lo) - Ap) + Am;
  const float h   = (bc * p.nu_tilde + ac * mu_t) + p.bdt * sa;
  const float xi  = p.nu2 + mu_t1 * mu_t1;
  a_new = (g * p.nu - h * mu_t1) / xi;
  b_new = (g * mu_t1 + h * p.nu) / xi;
}

template <int V>
__device__ __forceinline__ void load_v(const float *p, float (&r)[V]) {
  if constexpr (V == 4) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else {
    r[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void store_v(float *p, const float (&r)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4 *>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    *p = r[0];
  }
}

// One stencil row -> e[0..V+1] = columns j-1 .. j+V.
// XCH == 0: the two outer columns are re-read from memory (they hit L1/L2).
// XCH == 1: they come from the neighbouring lanes; only lane 0 / lane 63 of a
//           wave read memory for them.
template <int V, int XCH>
__device__ __forceinline__ void load_stencil_row(const float *row, int j, int jl, int jr,
                                                 int lane, float (&e)[V + 2])
{
  float body[V];
  load_v<V>(row + j, body);
#pragma unroll
  for (int k = 0; k < V; k++) e[k + 1] = body[k];
  if constexpr (XCH == 0) {
    e[0]     = row[jl];
    e[V + 1] = row[jr];
  } else {
    float edge = 0.0f;
    if (lane == 0)  edge = row[jl];
    if (lane == 63) edge = row[jr];
    const float from_left  = __shfl_up(body[V - 1], 1, 64);
    const float from_right = __shfl_down(body[0], 1, 64);
    e[0]     = (lane == 0)  ? edge : from_left;
    e[V + 1] = (lane == 63) ? edge : from_right;
  }
}

// ---------------------------------------------------------------------------
// sweep kernel: grid.x covers columns (V per lane), grid.y covers n-chunks
// ---------------------------------------------------------------------------
template <int V, int XCH>
__global__ void __launch_bounds__(256)
sweep_kernel(const SweepArgs p)
{
  const int lane = threadIdx.x & 63;
  const int jj   = p.jj0 + (blockIdx.x * blockDim.x + threadIdx.x) * V;   // first own column
  if (XCH == 0 && jj > p.j_last) return;
  // lanes past the end stay alive for the cross-lane exchange; keep their
  // addresses inside the row (pitch is a multiple of V when V > 1)
  const int j  = jj <= p.pitch - V ? jj : p.pitch - V;
  const int jl = j > 0 ? j - 1 : 0;
  const int jr = j + V < p.pitch ? j + V : p.pitch - 1;

  const int n0 = blockIdx.y * p.n_chunk;
  const int n1 = n0 + p.n_chunk < p.N ? n0 + p.n_chunk : p.N;

  // per-column field terms (reference: src/boltzmann_c_solver.c:363-364, macro :69)
  float mu_part[V], mu1_part[V];
  bool  wr[V];
  bool  wr_all = true;
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int   m   = p.m_offset + jj + k;
    const float phi = p.PhiYmin + p.dPhi * (float)(m - 1);
    mu_part[k]  = ((p.E_dc + p.E_omega * p.cos_t)  + p.B * phi) * p.dt / 2.0f;
    mu1_part[k] = ((p.E_dc + p.E_omega * p.cos_t1) + p.B * phi) * p.dt / 2.0f;
    wr[k]  = (jj + k >= p.j_first) && (jj + k <= p.j_last);
    wr_all = wr_all && wr[k];
  }

  const size_t P = (size_t)p.pitch;
  const float *a0r = p.a0  + (size_t)n0 * P + j;
  const float *acr = p.a_c + (size_t)n0 * P + j;
  const float *bcr = p.b_c + (size_t)n0 * P + j;
  float       *aor = p.a_out + (size_t)n0 * P + j;
  float       *bor = p.b_out + (size_t)n0 * P + j;
  const float *asr = p.a_s + (size_t)n0 * P;     // stencil rows are indexed with j, jl, jr
  const float *bsr = p.b_s + (size_t)n0 * P;

  // dA[q][k], dB[q][k]: column differences of the stencil row whose parity
  // relative to n0 is q^1 ... set 0 holds row n0-1, set 1 holds row n0.
  float dA[2][V], dB[2][V];
  {
    float e[V + 2];
    if (n0 >= 1) {
      load_stencil_row<V, XCH>(asr - P, j, jl, jr, lane, e);
#pragma unroll
      for (int k = 0; k < V; k++) dA[0][k] = e[k + 2] - e[k];
      load_stencil_row<V, XCH>(bsr - P, j, jl, jr, lane, e);
#pragma unroll
      for (int k = 0; k < V; k++) dB[0][k] = e[k + 2] - e[k];
    } else {
#pragma unroll
      for (int k = 0; k < V; k++) { dA[0][k] = 0.0f; dB[0][k] = 0.0f; }
    }
    load_stencil_row<V, XCH>(asr, j, jl, jr, lane, e);
#pragma unroll
    for (int k = 0; k < V; k++) dA[1][k] = e[k + 2] - e[k];
    load_stencil_row<V, XCH>(bsr, j, jl, jr, lane, e);
#pragma unroll
    for (int k = 0; k < V; k++) dB[1][k] = e[k + 2] - e[k];
  }

  auto row_update = [&](int n, float (&dAq)[V], float (&dBq)[V]) {
    // on entry dAq/dBq hold row n-1; on exit row n+1
    float ea[V + 2], eb[V + 2], a0v[V], ac[V], bc[V], an[V], bn[V];
    load_stencil_row<V, XCH>(asr + P, j, jl, jr, lane, ea);
    load_stencil_row<V, XCH>(bsr + P, j, jl, jr, lane, eb);
    load_v<V>(a0r, a0v);
    load_v<V>(acr, ac);
    load_v<V>(bcr, bc);
    const float fn = (float)n;
#pragma unroll
    for (int k = 0; k < V; k++) {
      point_update(p, n, fn, mu_part[k], mu1_part[k], a0v[k], ac[k], bc[k],
                   ea[k + 2], ea[k], eb[k + 2], eb[k], dAq[k], dBq[k], an[k], bn[k]);
      dAq[k] = ea[k + 2] - ea[k];
      dBq[k] = eb[k + 2] - eb[k];
    }
    if (wr_all) {
      store_v<V>(aor, an);
      if (n > 0) store_v<V>(bor, bn);
    } else {
#pragma unroll
      for (int k = 0; k < V; k++) {
        if (wr[k]) {
          aor[k] = an[k];
          if (n > 0) bor[k] = bn[k];
        }
      }
    }
    a0r += P; acr += P; bcr += P; aor += P; bor += P; asr += P; bsr += P;
  };

  int n = n0;
  for (; n + 1 < n1; n += 2) {
    row_update(n,     dA[0], dB[0]);
    row_update(n + 1, dA[1], dB[1]);
  }
  if (n < n1) row_update(n, dA[0], dB[0]);
}

// ---------------------------------------------------------------------------
// software-pipelined sweep: same arithmetic, four columns per lane; the loads of
// row n+1 are issued before row n is computed, so a wave keeps two rows of
// loads in flight.  Row bases are wave-uniform (SGPR base + 32-bit lane offset).
// NT: centre arrays / outputs, which are touched once per sweep, bypass the
// cache hierarchy's retention (nontemporal).
// ---------------------------------------------------------------------------
struct RowRegs {
  float ea[6], eb[6], a0v[4], ac[4], bc[4];
};

template <bool NT>
__device__ __forceinline__ float4 ld4(const float *base, unsigned off)
{
  const float4 *q = reinterpret_cast<const float4 *>(base + off);
  if constexpr (NT) {
    float4 v;
    v.x = __builtin_nontemporal_load(&q->x); v.y = __builtin_nontemporal_load(&q->y);
    v.z = __builtin_nontemporal_load(&q->z); v.w = __builtin_nontemporal_load(&q->w);
    return v;
  } else {
    return *q;
  }
}

template <bool NT>
__device__ __forceinline__ void st4(float *base, unsigned off, const float (&r)[4])
{
  float4 *q = reinterpret_cast<float4 *>(base + off);
  if constexpr (NT) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {r[0], r[1], r[2], r[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(q));
  } else {
    *q = make_float4(r[0], r[1], r[2], r[3]);
  }
}

template <bool NT, bool NTS = NT>
__global__ void __launch_bounds__(256)
sweep_pipe_kernel(const SweepArgs p)
{
  constexpr int V = 4;
  const int jj = p.jj0 + (blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (jj > p.j_last) return;
  const unsigned j  = (unsigned)jj;                       // jj <= j_last <= pitch-2, pitch % 4 == 0
  const unsigned jl = jj > 0 ? j - 1 : 0;
  const unsigned jr = jj + V < p.pitch ? j + V : (unsigned)p.pitch - 1;

  const int n0 = blockIdx.y * p.n_chunk;
  const int n1 = n0 + p.n_chunk < p.N ? n0 + p.n_chunk : p.N;

  float mu_part[V], mu1_part[V];
  bool  wr[V];
  bool  wr_all = true;
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int   m   = p.m_offset + jj + k;
    const float phi = p.PhiYmin + p.dPhi * (float)(m - 1);
    mu_part[k]  = ((p.E_dc + p.E_omega * p.cos_t)  + p.B * phi) * p.dt / 2.0f;
    mu1_part[k] = ((p.E_dc + p.E_omega * p.cos_t1) + p.B * phi) * p.dt / 2.0f;
    wr[k]  = (jj + k >= p.j_first) && (jj + k <= p.j_last);
    wr_all = wr_all && wr[k];
  }

  const size_t P = (size_t)p.pitch;

  auto stencil_diffs = [&](int r, float (&dAq)[V], float (&dBq)[V]) {
    const float *as = p.a_s + (size_t)r * P, *bs = p.b_s + (size_t)r * P;
    const float4 va = ld4<false>(as, j), vb = ld4<false>(bs, j);
    const float al = as[jl], ar = as[jr], bl = bs[jl], br = bs[jr];
    dAq[0] = va.y - al;   dAq[1] = va.z - va.x; dAq[2] = va.w - va.y; dAq[3] = ar - va.z;
    dBq[0] = vb.y - bl;   dBq[1] = vb.z - vb.x; dBq[2] = vb.w - vb.y; dBq[3] = br - vb.z;
  };

  float dA[2][V], dB[2][V];
  if (n0 >= 1) {
    stencil_diffs(n0 - 1, dA[0], dB[0]);
  } else {
#pragma unroll
    for (int k = 0; k < V; k++) { dA[0][k] = 0.0f; dB[0][k] = 0.0f; }
  }
  stencil_diffs(n0, dA[1], dB[1]);

  auto issue = [&](int n, RowRegs &r) {
    const size_t row = (size_t)n * P;                      // wave-uniform
    const float *as = p.a_s + row + P, *bs = p.b_s + row + P;
    const float4 va = ld4<false>(as, j), vb = ld4<false>(bs, j);
    r.ea[0] = as[jl]; r.ea[1] = va.x; r.ea[2] = va.y; r.ea[3] = va.z; r.ea[4] = va.w; r.ea[5] = as[jr];
    r.eb[0] = bs[jl]; r.eb[1] = vb.x; r.eb[2] = vb.y; r.eb[3] = vb.z; r.eb[4] = vb.w; r.eb[5] = bs[jr];
    const float4 v0 = ld4<NT>(p.a0 + row, j), v1 = ld4<NT>(p.a_c + row, j), v2 = ld4<NT>(p.b_c + row, j);
    r.a0v[0] = v0.x; r.a0v[1] = v0.y; r.a0v[2] = v0.z; r.a0v[3] = v0.w;
    r.ac[0] = v1.x;  r.ac[1] = v1.y;  r.ac[2] = v1.z;  r.ac[3] = v1.w;
    r.bc[0] = v2.x;  r.bc[1] = v2.y;  r.bc[2] = v2.z;  r.bc[3] = v2.w;
  };

  auto finish = [&](int n, const RowRegs &r, float (&dAq)[V], float (&dBq)[V]) {
    float an[V], bn[V];
    const float fn = (float)n;
#pragma unroll
    for (int k = 0; k < V; k++) {
      point_update(p, n, fn, mu_part[k], mu1_part[k], r.a0v[k], r.ac[k], r.bc[k],
                   r.ea[k + 2], r.ea[k], r.eb[k + 2], r.eb[k], dAq[k], dBq[k], an[k], bn[k]);
      dAq[k] = r.ea[k + 2] - r.ea[k];
      dBq[k] = r.eb[k + 2] - r.eb[k];
    }
    const size_t row = (size_t)n * P;
    float *ao = p.a_out + row, *bo = p.b_out + row;
    if (wr_all) {
      st4<NTS>(ao, j, an);
      if (n > 0) st4<NTS>(bo, j, bn);
    } else {
#pragma unroll
      for (int k = 0; k < V; k++) {
        if (wr[k]) {
          ao[j + k] = an[k];
          if (n > 0) bo[j + k] = bn[k];
        }
      }
    }
  };

  RowRegs r0, r1;
  issue(n0, r0);
  int n = n0;
  while (n + 1 < n1) {
    issue(n + 1, r1);
    finish(n, r0, dA[0], dB[0]);
    if (n + 2 < n1) issue(n + 2, r0);
    finish(n + 1, r1, dA[1], dB[1]);
    n += 2;
  }
  if (n < n1) finish(n, r0, dA[0], dB[0]);
}

// ---------------------------------------------------------------------------
// fused step: both sweeps of one loop iteration in one pass over the lattice.
//
// The half-step sweep at (n, m) reads the main sweep's output at (n+-1, m+-1)
// (reference: src/boltzmann_c_solver.c:399-402 read a_next/b_next written at
// :375-377).  A lane owns four columns and walks n upward; when it has the main
// sweep's row r it immediately forms the half-step row r-1, taking the m+-1
// values of row r from the adjacent lanes (DPP wave shifts) instead of from
// memory.  Lanes 0 and 63 of every wave are halo lanes: they recompute the main
// sweep for the columns next to the wave's 62 interior lanes and store nothing,
// so a wave needs no other wave and no barrier.  Traffic per lattice-point
// update: 5 reads (a0, a, b, a_hs, b_hs) + 4 writes = 36 B instead of 56 B.
// Every value is produced by the same point_update() in the same order as in
// the two-launch form, so results are bit-identical to it.
// ---------------------------------------------------------------------------
struct FusedArgs {
  const float *a0, *a_c, *b_c, *a_h, *b_h;   // a0; main lattice (t); half-step lattice (t+dt/2)
  float *a_n, *b_n, *a_hn, *b_hn;            // main lattice (t+dt); half-step lattice (t+3dt/2)
  float E_dc, E_omega, B, dt, dPhi, PhiYmin, nu, nu2, nu_tilde, bdt, c0, c1, c2, c3;
  int N, pitch, m_offset, n_chunk, jj0;
  int j1_first, j1_last;                     // columns the main sweep writes
  int j2_first, j2_last;                     // columns the half-step sweep writes
};

// value of lane-1 / lane+1 (lane 0 / lane 63 get 0; they never use it)
__device__ __forceinline__ float from_left_lane(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right_lane(float v)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// own[0..3] -> e[0..5] = columns j-1 .. j+4 using the adjacent lanes
__device__ __forceinline__ void widen(const float (&own)[4], float (&e)[6])
{
  e[0] = from_left_lane(own[3]);
  e[1] = own[0]; e[2] = own[1]; e[3] = own[2]; e[4] = own[3];
  e[5] = from_right_lane(own[0]);
}

#ifndef SLB_ADDR
#define SLB_ADDR 0      // how the double step addresses memory: 0 plain pointers; 1 a scalar row pointer (forced through
                        // v_readfirstlane) + a 32-bit lane offset: 2 spilled registers instead of 10, same speed
#endif
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}

// DIAG: timing-only build of the point update (same memory accesses and lane exchanges,
// two additions instead of the arithmetic); never used for results.
template <bool DIAG, class Params>
__device__ __forceinline__ void point_update_sel(
    const Params &p, int n, float fn, float mu_part, float mu1_part,
    float a0v, float ac, float bc, float Ap, float Am, float Bp, float Bm,
    float dA_lo, float dB_lo, float &a_new, float &b_new)
{
  if constexpr (DIAG) {
    a_new = (a0v + ac) + (Ap + dA_lo) + mu_part * fn;
    b_new = (bc + Bp) + (Am + Bm) + (dB_lo + mu1_part);
  } else {
    point_update(p, n, fn, mu_part, mu1_part, a0v, ac, bc, Ap, Am, Bp, Bm, dA_lo, dB_lo, a_new, b_new);
  }
}

// ---------------------------------------------------------------------------
// The same point update with fewer instructions, bit for bit the same results.
//
// div2_shared: the two quotients of a point have the same denominator xi = nu^2 + mu_t1^2.  The
// compiler's IEEE division is, per quotient: scale numerator and denominator (v_div_scale x 2),
// reciprocal (v_rcp, quarter rate), two fused steps to refine it, the quotient and two
// correction steps, v_div_fmas, v_div_fixup.  v_div_scale leaves a denominator alone unless it
// is denormal, at least 2^126, 2^96 times smaller than the numerator, or small together with a
// tiny numerator -- none of which happens for 0.5 <= xi <= 2^100 (the launcher checks nu^2 and
// the largest mu_t1 a run can form, and takes the plain division otherwise) -- so the refined
// reciprocal is formed once, from xi itself, and serves both quotients.  A numerator is still
// scaled by the hardware (v_div_scale: 2^64 up when the quotient would be tiny), and
// v_div_fmas is told so (its flag is "the numerator was changed"), then v_div_fixup for
// zeros, infinities and NaN as usual: 19 vector instructions and one reciprocal for both
// quotients instead of 22 and two.  slb_selftest_division compares it with `/` on the GPU.
//
// Row-dependent factors: the reference multiplies the n-1 terms by 0 at n < 2 (n = 0 for the
// other array) and by 2 at n = 1.  Those rows are the first two of 2^7 and more, and n is the
// same for the whole wave: the caller patches the two differences under a branch that only
// rows 0 and 1 take, so the other rows spend no instruction on it (1.0f * x is x itself).
//
// dta0: dt * a0 is the same product in every stage a row passes through; it is formed when
// the row is loaded and handed on instead of a0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void div2_shared(float n1, float n2, float d, float &q1, float &q2)
{
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float e  = __builtin_fmaf(-d, y0, 1.0f);
  const float y  = __builtin_fmaf(e, y0, y0);
  auto quotient = [&](float n) {
    bool hw_flag;
    const float ns = __builtin_amdgcn_div_scalef(n, d, true, &hw_flag);
    const bool scaled = ns != n;            // (the hardware's flag also fires for numerators beyond 2^96 x d,
                                            //  for which it would have scaled the denominator instead)
    const float qa = ns * y;
    const float ra = __builtin_fmaf(-d, qa, ns);
    const float qb = __builtin_fmaf(ra, y, qa);
    const float rb = __builtin_fmaf(-d, qb, ns);
    const float qc = __builtin_amdgcn_div_fmasf(rb, y, qb, scaled);
    return __builtin_amdgcn_div_fixupf(qc, d, n);
  };
  q1 = quotient(n1);
  q2 = quotient(n2);
}

// dta0v: dt * a0;  loA: chi * [n >= 1] (A[n-1][m+1] - A[n-1][m-1]);  loB: [n >= 2] (B[n-1][m+1] - B[n-1][m-1])
template <bool DIV2, class Params>
__device__ __forceinline__ void point_update_fast(
    const Params &p, float fn, float mu_part, float mu1_part,
    float dta0v, float ac, float bc, float Ap, float Am, float Bp, float Bm,
    float loA, float loB, float &a_new, float &b_new)
{
  const float mu_t  = fn * mu_part;
  const float mu_t1 = fn * mu1_part;
  const float sb = (Bp - Bm) - loB;
  const float g  = ((dta0v + ac * p.nu_tilde) - bc * mu_t) + p.bdt * sb;
  const float sa = (loA - Ap) + Am;
  const float h  = (bc * p.nu_tilde + ac * mu_t) + p.bdt * sa;
  const float xi = p.nu2 + mu_t1 * mu_t1;
  const float na = g * p.nu - h * mu_t1, nb = g * mu_t1 + h * p.nu;
  if constexpr (DIV2) {
    div2_shared(na, nb, xi, a_new, b_new);
  } else {
    a_new = na / xi;
    b_new = nb / xi;
  }
}

// Two adjacent columns at a time, spelled with two-element vectors so that every operation is
// one packed instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: separately rounded per
// element, like their scalar forms) whatever the optimiser would have paired on its own -- in
// particular the fused steps of the division, which it never pairs.
typedef float slb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ slb_f2 fma2(slb_f2 a, slb_f2 b, slb_f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void div2_shared_pair(slb_f2 n1, slb_f2 n2, slb_f2 d, slb_f2 &q1, slb_f2 &q2)
{
  const slb_f2 y0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const slb_f2 one = {1.0f, 1.0f};
  const slb_f2 e = fma2(-d, y0, one);
  const slb_f2 y = fma2(e, y0, y0);
  auto quotient = [&](slb_f2 n) {
    bool hw_flag;
    const slb_f2 ns = {__builtin_amdgcn_div_scalef(n.x, d.x, true, &hw_flag), __builtin_amdgcn_div_scalef(n.y, d.y, true, &hw_flag)};
    const bool scaled_x = ns.x != n.x, scaled_y = ns.y != n.y;
    const slb_f2 qa = ns * y;
    const slb_f2 ra = fma2(-d, qa, ns);
    const slb_f2 qb = fma2(ra, y, qa);
    const slb_f2 rb = fma2(-d, qb, ns);
    const float cx = __builtin_amdgcn_div_fmasf(rb.x, y.x, qb.x, scaled_x);
    const float cy = __builtin_amdgcn_div_fmasf(rb.y, y.y, qb.y, scaled_y);
    const slb_f2 q = {__builtin_amdgcn_div_fixupf(cx, d.x, n.x), __builtin_amdgcn_div_fixupf(cy, d.y, n.y)};
    return q;
  };
  q1 = quotient(n1);
  q2 = quotient(n2);
}

template <bool DIV2, class Params>
__device__ __forceinline__ void point_update_pair(
    const Params &p, float fn, slb_f2 mu_part, slb_f2 mu1_part,
    slb_f2 dta0v, slb_f2 ac, slb_f2 bc, slb_f2 Ap, slb_f2 Am, slb_f2 Bp, slb_f2 Bm,
    slb_f2 loA, slb_f2 loB, slb_f2 &a_new, slb_f2 &b_new)
{
  const slb_f2 mu_t  = fn * mu_part;
  const slb_f2 mu_t1 = fn * mu1_part;
  const slb_f2 sb = (Bp - Bm) - loB;
  const slb_f2 g  = ((dta0v + ac * p.nu_tilde) - bc * mu_t) + p.bdt * sb;
  const slb_f2 sa = (loA - Ap) + Am;
  const slb_f2 h  = (bc * p.nu_tilde + ac * mu_t) + p.bdt * sa;
  const slb_f2 xi = p.nu2 + mu_t1 * mu_t1;
  const slb_f2 na = g * p.nu - h * mu_t1, nb = g * mu_t1 + h * p.nu;
  if constexpr (DIV2) {
    div2_shared_pair(na, nb, xi, a_new, b_new);
  } else {
    a_new = na / xi;
    b_new = nb / xi;
  }
}

template <bool NT, int MINW, bool DIAG = false, bool NTS = NT>
__global__ void __launch_bounds__(256, MINW)
fused_step_kernel(const FusedArgs p)
{
  constexpr int V = 4;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int jj   = p.jj0 + (wave * 62 + lane - 1) * V;          // first own column (may be < 0)
  const int j_hi = p.j1_last > p.j2_last ? p.j1_last : p.j2_last;
  if (p.jj0 + wave * 62 * V > j_hi) return;                     // whole wave past the end
  const bool interior = lane >= 1 && lane <= 62;
  // columns used for addressing stay inside the row; lanes that were clamped never store
  const int jcl = jj < 0 ? 0 : (jj > p.pitch - V ? p.pitch - V : jj);
  const unsigned j = (unsigned)jcl;

  const int n0 = blockIdx.y * p.n_chunk;
  const int n1 = n0 + p.n_chunk < p.N ? n0 + p.n_chunk : p.N;
  const int r_lo  = n0 > 0 ? n0 - 1 : 0;        // first main-sweep row this workgroup forms
  const int r_end = n1;                         // last row needed: n1 (row N comes from memory)

  // B*phi_y per column; the field terms (reference: src/boltzmann_c_solver.c:363-364)
  // are re-formed from it each row, which costs three operations and saves registers
  float bphi[V];
  unsigned wmask = 0;                           // bit k: main sweep stores column k; bit 4+k: half sweep;
                                                // bit 8+k: the main sweep produces column k
  bool need_old = false;
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int c = jj + k;
    bphi[k] = p.B * (p.PhiYmin + p.dPhi * (float)(p.m_offset + c - 1));
    const bool ok1 = c >= p.j1_first && c <= p.j1_last;
    if (ok1) wmask |= 1u << (8 + k);
    if (interior && ok1) wmask |= 1u << k;
    if (interior && c >= p.j2_first && c <= p.j2_last) wmask |= 1u << (4 + k);
    // a column the half-step sweep reads but the main sweep does not write keeps its old value
    if (!ok1 && c >= p.j2_first - 1 && c <= p.j2_last + 1 && c >= 0 && c < p.pitch) need_old = true;
  }
  if (jj != jcl) need_old = false;
  const bool w1_all = (wmask & 0xfu) == 0xfu, w2_all = (wmask & 0xf0u) == 0xf0u;
  const float s0 = p.E_dc + p.E_omega * p.c0, s1 = p.E_dc + p.E_omega * p.c1;
  const float s2 = p.E_dc + p.E_omega * p.c2, s3 = p.E_dc + p.E_omega * p.c3;

  const size_t P = (size_t)p.pitch;

  // sliding state (rows relative to the main-sweep row r of the current iteration)
  float hA_p[V], hB_p[V];        // half-step lattice, row r-1, own columns
  float hA_c[V], hB_c[V];        //                    row r
  float a0_p[V];                 // a0 row r-1
  float d1A_pp[V], d1B_pp[V];    // main-sweep output, row r-2: X[m+1]-X[m-1]
  float d1A_p[V], d1B_p[V];      //                    row r-1

  auto load_h = [&](int r, float (&ownA)[V], float (&ownB)[V]) {
    const float4 va = ld4<NT>(p.a_h + (size_t)r * P, j), vb = ld4<NT>(p.b_h + (size_t)r * P, j);
    ownA[0] = va.x; ownA[1] = va.y; ownA[2] = va.z; ownA[3] = va.w;
    ownB[0] = vb.x; ownB[1] = vb.y; ownB[2] = vb.z; ownB[3] = vb.w;
  };

#pragma unroll
  for (int k = 0; k < V; k++) {
    hA_p[k] = hB_p[k] = 0.0f; a0_p[k] = 0.0f;
    d1A_pp[k] = d1B_pp[k] = d1A_p[k] = d1B_p[k] = 0.0f;
  }
  if (r_lo >= 1) load_h(r_lo - 1, hA_p, hB_p);
  load_h(r_lo, hA_c, hB_c);

  // the five input rows of main-sweep row r: a0, a, b at row r; half-step lattice at row r+1
  struct RowIn { float a0[V], ac[V], bc[V], hA[V], hB[V]; };
  auto load_row = [&](int r, RowIn &in) {
    const size_t row = (size_t)r * P;
    const float4 v0 = ld4<NT>(p.a0 + row, j), v1 = ld4<NT>(p.a_c + row, j), v2 = ld4<NT>(p.b_c + row, j);
    const float4 va = ld4<NT>(p.a_h + row + P, j), vb = ld4<NT>(p.b_h + row + P, j);
    in.a0[0] = v0.x; in.a0[1] = v0.y; in.a0[2] = v0.z; in.a0[3] = v0.w;
    in.ac[0] = v1.x; in.ac[1] = v1.y; in.ac[2] = v1.z; in.ac[3] = v1.w;
    in.bc[0] = v2.x; in.bc[1] = v2.y; in.bc[2] = v2.z; in.bc[3] = v2.w;
    in.hA[0] = va.x; in.hA[1] = va.y; in.hA[2] = va.z; in.hA[3] = va.w;
    in.hB[0] = vb.x; in.hB[1] = vb.y; in.hB[2] = vb.z; in.hB[3] = vb.w;
  };
  RowIn cur = {};

  for (int r = r_lo; r <= r_end; r++) {
    float o1A[V], o1B[V];          // main-sweep output row r, own columns
    const size_t row = (size_t)r * P;
    if (r < p.N) {
      // ---- main sweep, row r ----
      load_row(r, cur);
      float ea[6], eb[6], la[6], lb[6];
      widen(cur.hA, ea); widen(cur.hB, eb);       // row r+1 with its m+-1 neighbours
      widen(hA_p, la);   widen(hB_p, lb);         // row r-1 likewise
      const float fr = (float)r;
#pragma unroll
      for (int k = 0; k < V; k++) {
        const float mu_a = (s0 + bphi[k]) * p.dt / 2.0f, mu_b = (s1 + bphi[k]) * p.dt / 2.0f;
        point_update_sel<DIAG>(p, r, fr, mu_a, mu_b, cur.a0[k], cur.ac[k], cur.bc[k],
                     ea[k + 2], ea[k], eb[k + 2], eb[k], la[k + 2] - la[k], lb[k + 2] - lb[k],
                     o1A[k], o1B[k]);
      }
      if (r >= n0 && r < n1) {
        if (w1_all) {
          st4<NTS>(p.a_n + row, j, o1A);
          if (r > 0) st4<NTS>(p.b_n + row, j, o1B);
        } else {
#pragma unroll
          for (int k = 0; k < V; k++) {
            if (wmask & (1u << k)) {
              p.a_n[row + j + k] = o1A[k];
              if (r > 0) p.b_n[row + j + k] = o1B[k];
            }
          }
        }
      }
      if (need_old) {
        // columns the main sweep does not write keep their present contents.  (Row 0 of b
        // is not written either, but no update ever reads it: the n-1 term is dropped for n<2.)
        const float4 oa = *reinterpret_cast<const float4 *>(p.a_n + row + j);
        const float4 ob = *reinterpret_cast<const float4 *>(p.b_n + row + j);
        const float olda[V] = {oa.x, oa.y, oa.z, oa.w}, oldb[V] = {ob.x, ob.y, ob.z, ob.w};
#pragma unroll
        for (int k = 0; k < V; k++) {
          if (!(wmask & (1u << (8 + k)))) { o1A[k] = olda[k]; o1B[k] = oldb[k]; }
        }
      }
    } else {
      // row N is never written by the main sweep: its present contents are the stencil
      const float4 oa = *reinterpret_cast<const float4 *>(p.a_n + row + j);
      const float4 ob = *reinterpret_cast<const float4 *>(p.b_n + row + j);
      o1A[0] = oa.x; o1A[1] = oa.y; o1A[2] = oa.z; o1A[3] = oa.w;
      o1B[0] = ob.x; o1B[1] = ob.y; o1B[2] = ob.z; o1B[3] = ob.w;
    }

    // ---- half-step sweep, row r-1, from main rows r (here) and r-2 (differences kept) ----
    float e1a[6], e1b[6];
    widen(o1A, e1a); widen(o1B, e1b);
    const int n = r - 1;
    if (n >= n0) {
      float o2A[V], o2B[V];
      const float fn = (float)n;
      const size_t rown = (size_t)n * P;
#pragma unroll
      for (int k = 0; k < V; k++) {
        const float mu_a = (s2 + bphi[k]) * p.dt / 2.0f, mu_b = (s3 + bphi[k]) * p.dt / 2.0f;
        point_update_sel<DIAG>(p, n, fn, mu_a, mu_b, a0_p[k], hA_p[k], hB_p[k],
                     e1a[k + 2], e1a[k], e1b[k + 2], e1b[k], d1A_pp[k], d1B_pp[k], o2A[k], o2B[k]);
      }
      if (w2_all) {
        st4<NTS>(p.a_hn + rown, j, o2A);
        if (n > 0) st4<NTS>(p.b_hn + rown, j, o2B);
      } else {
#pragma unroll
        for (int k = 0; k < V; k++) {
          if (wmask & (1u << (4 + k))) {
            p.a_hn[rown + j + k] = o2A[k];
            if (n > 0) p.b_hn[rown + j + k] = o2B[k];
          }
        }
      }
    }
    // ---- slide the window one row up ----
#pragma unroll
    for (int k = 0; k < V; k++) {
      d1A_pp[k] = d1A_p[k];               d1B_pp[k] = d1B_p[k];
      d1A_p[k]  = e1a[k + 2] - e1a[k];    d1B_p[k]  = e1b[k + 2] - e1b[k];
      hA_p[k] = hA_c[k];   hB_p[k] = hB_c[k];
      hA_c[k] = cur.hA[k]; hB_c[k] = cur.hB[k];
      a0_p[k] = cur.a0[k];
    }
  }
}

// ---------------------------------------------------------------------------
// double step: TWO loop iterations (four sweeps) in one pass over the lattice.
//
// Stage A = main sweep at t, B = half-step sweep at t, C = main sweep at t+dt,
// D = half-step sweep at t+dt.  Each stage reads the previous stage's output at
// rows n+-1, columns m+-1 (reference: src/boltzmann_c_solver.c:370-372, 400-402),
// so a lane walking n upward runs A at row r, B at r-1, C at r-2, D at r-3 and
// hands rows from stage to stage in registers; columns m+-1 come from the
// adjacent lanes.  Every stage costs a column of validity per side of a wave (the first
// for want of a neighbour lane's input, the others at each hand-off): four columns, which
// one halo lane on each side of a wave covers.
// Only C's and D's outputs (the lattices at t+2dt and t+5dt/2) are stored; the
// lattices in between never reach memory: 5 reads + 4 writes = 36 B per TWO
// lattice-point updates.
//
// Cells no sweep writes (row N, boundary columns) differ from buffer to buffer in
// the reference's ping-pong; each stage takes them from the buffer that would
// hold its output there: stage A from p1a/p1b (the reference's a/b[next]), stage B
// from p2a/p2b (a/b[next_hs]), stage C from its own output arrays.  The outputs
// must not alias the inputs (halo lanes of other waves still read them), so the
// caller passes spare arrays whose never-written cells equal those of a/b[current]
// and a/b[current_hs], and exchanges them with those afterwards.
// ---------------------------------------------------------------------------
struct Fused2Args {
  const float *a0, *a_c, *b_c, *a_h, *b_h;   // a0; main lattice (t); half-step lattice (t+dt/2)
  const float *p1a, *p1b, *p2a, *p2b;        // never-written cells of the in-between lattices
  float *a_o, *b_o, *a_ho, *b_ho;            // main lattice (t+2dt); half-step lattice (t+5dt/2)
  float *rows1, *rows2;                      // ROWS: [3][pitch] each: a rows 0, 1 and b row 1 of stage A's / C's output
  const double *a0_row, *a0_col;             // A0F: a0[n][j] = (float)(a0_row[n] * a0_col[j]), see a0_factors_check_kernel
  float E_dc, E_omega, B, dt, dPhi, PhiYmin, nu, nu2, nu_tilde, bdt;
  float c[8];                                // cosines: stage A 0,1  B 2,3  C 4,5  D 6,7
  int N, pitch, m_offset, jj0;
  int jf[4], jl[4];                          // columns stage A, B, C, D produce
  // Launch schedule.  Workgroups are dispatched in the order of their index; phase p takes
  // indices wg_begin[p] .. wg_begin[p+1]-1 and deals them out over the column-strip groups
  // g0[p] .. g0[p]+ng[p]-1 (one group = the strips of one workgroup), fastest, and the row
  // chunks of length len[p] that rows 0..N-1 are cut into.  Indices a phase was padded with
  // (so that the next phase starts on a multiple of 8) do nothing.
  int n_phase, wg_begin[4], g0[3], ng[3], len[3];
};

// ROWS: the three rows av() reads (a rows 0 and 1, b row 1) of BOTH new main lattices are
// also written to small side buffers, so the harmonic sums of the in-between time step can
// be formed although that lattice itself is never stored.
// H: halo lanes per side of a wave (1 suffices; 2, what the launcher picks, puts the 60 interior
// lanes' stores -- 960 bytes per row -- on 64-byte boundaries; 4 puts 896 bytes on 128-byte line
// boundaries at the price of forming 8 of 64 columns twice)
// NTS: non-temporal stores.  The interior lanes of a wave store pieces (992 bytes at H = 1)
// that do not line up with 128-byte lines, so every boundary between two waves leaves a line
// in two pieces; marked non-temporal the pieces are written out 2-10 % faster (measured on
// MI355X with this kernel and, without arithmetic, with scripts/membench.hip).  NTL: the same
// mark on the loads -- slower, the halo lanes' re-reads then miss.
// A0F: the equilibrium coefficients are not read from memory but formed from their two factors,
// a0[n][j] = (float)(a0_row[n] * a0_col[j]) in double arithmetic (one v_mul_f64 and one
// v_cvt_f32_f64 per value): 4 of the 36 bytes per cell and launch less.  The caller has checked
// cell by cell that this gives the bits of the stored a0 (slb_a0_factors_check).  1: a row's
// values are formed once and handed from stage to stage like loaded ones; 2: every stage forms
// the values of its own row (12 registers fewer, three times the double-precision work).
// OPT (bits): 1 = point_update_fast (row-dependent factors patched under a wave-uniform branch,
// dt * a0 carried instead of a0), 2 = also div2_shared.  Same bits as OPT = 0.
template <int MINW, bool DIAG = false, bool ROWS = false, int H = 1, bool NTS = false, bool NTL = false, int A0F = 0, int OPT = 0>
__global__ void __launch_bounds__(256, MINW)
fused2_step_kernel(const Fused2Args p)
{
  constexpr int V = 4, INNER = 64 - 2 * H;
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Neighbouring
  // column strips overlap by their halo lanes, so strips that are adjacent in memory are
  // given to workgroups of the same XCD: the shared lines are then served by that XCD's L2.
  // (Placement only changes where re-reads are served from, never what is computed.)
  int id = blockIdx.x, ph = 0;
  if (p.n_phase > 1 && id >= p.wg_begin[1]) ph = 1;
  if (p.n_phase > 2 && id >= p.wg_begin[2]) ph = 2;
  id -= p.wg_begin[ph];
  const int ng = p.ng[ph];
  const int chunk = id / ng;
  int bx = id - chunk * ng;
  const int n0 = chunk * p.len[ph];
  if (n0 >= p.N) return;                                        // padding of the phase
  const int per_xcd = ng >> 3;
  if (bx < (per_xcd << 3)) bx = (bx & 7) * per_xcd + (bx >> 3);
  bx += p.g0[ph];
  const int wave = (bx * blockDim.x + threadIdx.x) >> 6;
  const int jj   = p.jj0 + (wave * INNER + lane - H) * V;       // first own column (may be < 0)
  int j_hi = p.jl[0];
#pragma unroll
  for (int q = 1; q < 4; q++) j_hi = p.jl[q] > j_hi ? p.jl[q] : j_hi;
  if (p.jj0 + wave * INNER * V > j_hi) return;                  // whole wave past the end
  const bool interior = lane >= H && lane < 64 - H;
  const int jcl = jj < 0 ? 0 : (jj > p.pitch - V ? p.pitch - V : jj);
  const unsigned j = (unsigned)jcl;

  const int n1 = n0 + p.len[ph] < p.N ? n0 + p.len[ph] : p.N;
  // first row each stage has to form so that D can form rows n0..n1-1
  const int c_lo = n0 > 1 ? n0 - 1 : 0, b_lo = n0 > 2 ? n0 - 2 : 0, a_lo = n0 > 3 ? n0 - 3 : 0;
  const int r_end = n1 + 2;                      // D forms row n1-1 when A stands at n1+2

  // per-column flags.  bits 0-3: stage A produces column k; 4-7: B; 8-11: C;
  // 12-15: C stores column k; 16-19: D stores column k
  float bphi[V];
  unsigned flags = 0;
  bool need1 = false, need2 = false, need3 = false;
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int c = jj + k;
    bphi[k] = p.B * (p.PhiYmin + p.dPhi * (float)(p.m_offset + c - 1));
    const bool in_row = c >= 0 && c < p.pitch;
    const bool okA = c >= p.jf[0] && c <= p.jl[0], okB = c >= p.jf[1] && c <= p.jl[1];
    const bool okC = c >= p.jf[2] && c <= p.jl[2], okD = c >= p.jf[3] && c <= p.jl[3];
    if (okA) flags |= 1u << k;
    if (okB) flags |= 1u << (4 + k);
    if (okC) flags |= 1u << (8 + k);
    if (okC && interior) flags |= 1u << (12 + k);
    if (okD && interior) flags |= 1u << (16 + k);
    // a column the next stage reads at m+-1 that this stage does not produce keeps its old value
    if (!okA && in_row && c >= p.jf[1] - 1 && c <= p.jl[1] + 1) need1 = true;
    if (!okB && in_row && c >= p.jf[2] - 1 && c <= p.jl[2] + 1) need2 = true;
    if (!okC && in_row && c >= p.jf[3] - 1 && c <= p.jl[3] + 1) need3 = true;
  }
  if (jj != jcl) need1 = need2 = need3 = false;
  const bool wC_all = ((flags >> 12) & 0xfu) == 0xfu, wD_all = ((flags >> 16) & 0xfu) == 0xfu;
  float s[8];
#pragma unroll
  for (int q = 0; q < 8; q++) s[q] = p.E_dc + p.E_omega * p.c[q];

  const size_t P = (size_t)p.pitch;

  const unsigned jb = j * 4u;
  (void)jb;
  auto at_of = [&](const float *base, size_t row) -> const char * {
#if SLB_ADDR == 1
    return reinterpret_cast<const char *>(uniform_ptr(base + row)) + jb;
#else
    return reinterpret_cast<const char *>(base + row + j);
#endif
  };
  auto load4 = [&](const float *base, size_t row, float (&v)[V]) {
    const char *at = at_of(base, row);
    if constexpr (NTL) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 q = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(at));
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      const float4 q = *reinterpret_cast<const float4 *>(at);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
  };
  auto store4 = [&](float *base, size_t row, const float (&v)[V]) {
    float4 *at = reinterpret_cast<float4 *>(const_cast<char *>(at_of(base, row)));
    if constexpr (NTS) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 q = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(at));
    } else {
      *at = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store1 = [&](float *base, size_t row, int k, float x) {
    reinterpret_cast<float *>(const_cast<char *>(at_of(base, row)))[k] = x;
  };
  // replace the columns a stage does not produce by the present contents of `pa`/`pb`
  auto keep_old = [&](const float *pa, const float *pb, size_t row, unsigned okbits,
                      float (&oa)[V], float (&ob)[V]) {
    float xa[V], xb[V];
    load4(pa, row, xa); load4(pb, row, xb);
#pragma unroll
    for (int k = 0; k < V; k++)
      if (!(okbits & (1u << k))) { oa[k] = xa[k]; ob[k] = xb[k]; }
  };
  // one stage at row n for the four own columns
  auto stage = [&](int n, float sa, float sb, const float (&a0v)[V], const float (&ca)[V], const float (&cb)[V],
                   const float (&upa)[6], const float (&upb)[6], const float (&dla)[V], const float (&dlb)[V],
                   float (&oa)[V], float (&ob)[V]) {
    const float fn = (float)n;
    if constexpr (OPT == 0 || DIAG) {
#pragma unroll
      for (int k = 0; k < V; k++) {
        const float mu_a = (sa + bphi[k]) * p.dt / 2.0f, mu_b = (sb + bphi[k]) * p.dt / 2.0f;
        point_update_sel<DIAG>(p, n, fn, mu_a, mu_b, a0v[k], ca[k], cb[k],
                               upa[k + 2], upa[k], upb[k + 2], upb[k], dla[k], dlb[k], oa[k], ob[k]);
      }
    } else {
      float la[V], lb[V];
#pragma unroll
      for (int k = 0; k < V; k++) { la[k] = dla[k]; lb[k] = dlb[k]; }
      if (n < 2) {                               // the same for the whole wave; rows 0 and 1 only
        asm volatile("" ::);                     // (keeps this a branch: as selects it would cost every row)
#pragma unroll
        for (int k = 0; k < V; k++) { lb[k] = 0.0f; la[k] = n == 0 ? 0.0f : 2.0f * la[k]; }
      }
      if constexpr ((OPT & 4) != 0) {
#pragma unroll
        for (int k = 0; k < V; k += 2) {
          const slb_f2 bp = {bphi[k], bphi[k + 1]};
          const slb_f2 mu_a = (sa + bp) * p.dt / 2.0f, mu_b = (sb + bp) * p.dt / 2.0f;
          slb_f2 qa, qb;
          point_update_pair<(OPT & 2) != 0>(p, fn, mu_a, mu_b, slb_f2{a0v[k], a0v[k + 1]}, slb_f2{ca[k], ca[k + 1]},
                                            slb_f2{cb[k], cb[k + 1]}, slb_f2{upa[k + 2], upa[k + 3]}, slb_f2{upa[k], upa[k + 1]},
                                            slb_f2{upb[k + 2], upb[k + 3]}, slb_f2{upb[k], upb[k + 1]},
                                            slb_f2{la[k], la[k + 1]}, slb_f2{lb[k], lb[k + 1]}, qa, qb);
          oa[k] = qa.x; oa[k + 1] = qa.y; ob[k] = qb.x; ob[k + 1] = qb.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < V; k++) {
          const float mu_a = (sa + bphi[k]) * p.dt / 2.0f, mu_b = (sb + bphi[k]) * p.dt / 2.0f;
          point_update_fast<(OPT & 2) != 0>(p, fn, mu_a, mu_b, a0v[k], ca[k], cb[k],
                                            upa[k + 2], upa[k], upb[k + 2], upb[k], la[k], lb[k], oa[k], ob[k]);
        }
      }
    }
  };
  auto diffs = [&](const float (&own)[V], float (&d)[V]) {
    float e[6];
    widen(own, e);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] = e[k + 2] - e[k];
  };

  // sliding state, named by the row they hold relative to stage A's row r
  float hA_p[V] = {}, hB_p[V] = {}, hA_c[V], hB_c[V];           // half-step lattice rows r-1, r
  float a0_1[V] = {}, a0_2[V] = {}, a0_3[V] = {};               // a0 rows r-1, r-2, r-3
  float o1A_p[V] = {}, o1B_p[V] = {}, o1A_pp[V] = {}, o1B_pp[V] = {};   // stage A output rows r-1, r-2
  float o2A_p[V] = {}, o2B_p[V] = {}, o2A_pp[V] = {}, o2B_pp[V] = {};   // stage B output rows r-2, r-3
  float d3A_p[V] = {}, d3B_p[V] = {}, d3A_pp[V] = {}, d3B_pp[V] = {};   // stage C output rows r-3, r-4: X[m+1]-X[m-1]

  if (a_lo >= 1) { load4(p.a_h, (size_t)(a_lo - 1) * P, hA_p); load4(p.b_h, (size_t)(a_lo - 1) * P, hB_p); }
  load4(p.a_h, (size_t)a_lo * P, hA_c); load4(p.b_h, (size_t)a_lo * P, hB_c);

  // A0F: this lane's column factors
  double a0g[V] = {};
  if constexpr (A0F != 0) {
    const double2 g01 = *reinterpret_cast<const double2 *>(p.a0_col + j);
    const double2 g23 = *reinterpret_cast<const double2 *>(p.a0_col + j + 2);
    a0g[0] = g01.x; a0g[1] = g01.y; a0g[2] = g23.x; a0g[3] = g23.y;
  }
  auto a0_of_row = [&](int n, float (&v)[V]) {
    if constexpr (A0F == 4) {                    // timing probe only: a0 neither read nor formed
#pragma unroll
      for (int k = 0; k < V; k++) v[k] = bphi[k];
      return;
    }
    const double c = p.a0_row[n];                // wave-uniform
#pragma unroll
    for (int k = 0; k < V; k++) {
      if constexpr (A0F == 3) v[k] = (float)a0g[k] * (float)c;      // timing probe only: not the reference's bits
      else v[k] = (float)(a0g[k] * c);
    }
  };

  struct RowIn { float a0[V], ac[V], bc[V], hA[V], hB[V]; };
  auto load_row = [&](int r, RowIn &in) {
    const size_t row = (size_t)r * P;
    if constexpr (A0F == 0) load4(p.a0, row, in.a0);
    load4(p.a_c, row, in.ac); load4(p.b_c, row, in.bc);
    load4(p.a_h, row + P, in.hA); load4(p.b_h, row + P, in.hB);
    if constexpr (A0F != 0) a0_of_row(r, in.a0);
    if constexpr (OPT != 0 && !DIAG) {
#pragma unroll
      for (int k = 0; k < V; k++) in.a0[k] = p.dt * in.a0[k];     // handed on as dt * a0
    }
  };
  RowIn cur = {};

  for (int r = a_lo; r <= r_end; r++) {

    // ---- stage A: main sweep at t, row r ----
    float o1A[V] = {}, o1B[V] = {};
    if (r < p.N) {
      load_row(r, cur);
      float ea[6], eb[6], dla[V], dlb[V];
      widen(cur.hA, ea); widen(cur.hB, eb);
      diffs(hA_p, dla);  diffs(hB_p, dlb);
      stage(r, s[0], s[1], cur.a0, cur.ac, cur.bc, ea, eb, dla, dlb, o1A, o1B);
      if constexpr (ROWS) {
        if (r <= 1 && n0 == 0 && interior) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            if (flags & (1u << k)) {
              store1(p.rows1, (size_t)r * P, k, o1A[k]);
              if (r == 1) store1(p.rows1, 2 * P, k, o1B[k]);
            }
          }
        }
      }
      if (need1) keep_old(p.p1a, p.p1b, (size_t)r * P, flags, o1A, o1B);
    } else if (r == p.N) {
      load4(p.p1a, (size_t)r * P, o1A); load4(p.p1b, (size_t)r * P, o1B);
      if constexpr (ROWS) {
        // a single harmonic: row 1 is the row no sweep writes, and av() reads it all the same
        if (r == 1 && n0 == 0 && interior) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            if (flags & (1u << k)) { store1(p.rows1, P, k, o1A[k]); store1(p.rows1, 2 * P, k, o1B[k]); }
          }
        }
      }
    }

    // ---- stage B: half-step sweep at t, row r-1 ----
    const int rb = r - 1;
    float o2A[V] = {}, o2B[V] = {};
    if (rb >= b_lo && rb < p.N) {
      float ea[6], eb[6], dla[V], dlb[V];
      widen(o1A, ea); widen(o1B, eb);
      diffs(o1A_pp, dla); diffs(o1B_pp, dlb);
      if constexpr (A0F == 2) a0_of_row(rb, a0_1);
      stage(rb, s[2], s[3], a0_1, hA_p, hB_p, ea, eb, dla, dlb, o2A, o2B);
      if (need2) keep_old(p.p2a, p.p2b, (size_t)rb * P, flags >> 4, o2A, o2B);
    } else if (rb == p.N) {
      load4(p.p2a, (size_t)rb * P, o2A); load4(p.p2b, (size_t)rb * P, o2B);
    }

    // ---- stage C: main sweep at t+dt, row r-2 (stored) ----
    const int rc = r - 2;
    float o3A[V] = {}, o3B[V] = {};
    if (rc >= c_lo && rc < p.N) {
      float ea[6], eb[6], dla[V], dlb[V];
      widen(o2A, ea); widen(o2B, eb);
      diffs(o2A_pp, dla); diffs(o2B_pp, dlb);
      if constexpr (A0F == 2) a0_of_row(rc, a0_2);
      stage(rc, s[4], s[5], a0_2, o1A_pp, o1B_pp, ea, eb, dla, dlb, o3A, o3B);
      const size_t row = (size_t)rc * P;
      if (rc >= n0 && rc < n1) {
        if (wC_all) {
          store4(p.a_o, row, o3A);
          if (rc > 0) store4(p.b_o, row, o3B);
        } else {
#pragma unroll
          for (int k = 0; k < V; k++) {
            if (flags & (1u << (12 + k))) {
              store1(p.a_o, row, k, o3A[k]);
              if (rc > 0) store1(p.b_o, row, k, o3B[k]);
            }
          }
        }
        if constexpr (ROWS) {
          if (rc <= 1) {
#pragma unroll
            for (int k = 0; k < V; k++) {
              if (flags & (1u << (12 + k))) {
                store1(p.rows2, row, k, o3A[k]);
                if (rc == 1) store1(p.rows2, 2 * P, k, o3B[k]);
              }
            }
          }
        }
      }
      if (need3) keep_old(p.a_o, p.b_o, row, flags >> 8, o3A, o3B);
    } else if (rc == p.N) {
      load4(p.a_o, (size_t)rc * P, o3A); load4(p.b_o, (size_t)rc * P, o3B);
      if constexpr (ROWS) {
        if (rc == 1 && n0 == 0) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            if (flags & (1u << (12 + k))) { store1(p.rows2, P, k, o3A[k]); store1(p.rows2, 2 * P, k, o3B[k]); }
          }
        }
      }
    }

    // ---- stage D: half-step sweep at t+dt, row r-3 (stored) ----
    const int rd = r - 3;
    float e3a[6], e3b[6];
    widen(o3A, e3a); widen(o3B, e3b);
    if (rd >= n0 && rd < n1) {
      float o4A[V], o4B[V];
      if constexpr (A0F == 2) a0_of_row(rd, a0_3);
      stage(rd, s[6], s[7], a0_3, o2A_pp, o2B_pp, e3a, e3b, d3A_pp, d3B_pp, o4A, o4B);
      const size_t row = (size_t)rd * P;
      if (wD_all) {
        store4(p.a_ho, row, o4A);
        if (rd > 0) store4(p.b_ho, row, o4B);
      } else {
#pragma unroll
        for (int k = 0; k < V; k++) {
          if (flags & (1u << (16 + k))) {
            store1(p.a_ho, row, k, o4A[k]);
            if (rd > 0) store1(p.b_ho, row, k, o4B[k]);
          }
        }
      }
    }

    // ---- slide every window one row up ----
#pragma unroll
    for (int k = 0; k < V; k++) {
      d3A_pp[k] = d3A_p[k]; d3B_pp[k] = d3B_p[k];
      d3A_p[k] = e3a[k + 2] - e3a[k]; d3B_p[k] = e3b[k + 2] - e3b[k];
      o2A_pp[k] = o2A_p[k]; o2B_pp[k] = o2B_p[k]; o2A_p[k] = o2A[k]; o2B_p[k] = o2B[k];
      o1A_pp[k] = o1A_p[k]; o1B_pp[k] = o1B_p[k]; o1A_p[k] = o1A[k]; o1B_p[k] = o1B[k];
      if constexpr (A0F != 2) { a0_3[k] = a0_2[k]; a0_2[k] = a0_1[k]; a0_1[k] = cur.a0[k]; }
      hA_p[k] = hA_c[k]; hB_p[k] = hB_c[k]; hA_c[k] = cur.hA[k]; hB_c[k] = cur.hB[k];
    }
  }
}

// Does (float)(a0_row[n] * a0_col[j]) -- the product and its narrowing as the double-step kernel
// forms them -- give the stored a0[n][j], bit for bit?  bad[n] counts the cells of row n that
// differ, for rows row0 .. row0+nrows-1 and columns 0 .. ncols-1.
__global__ void __launch_bounds__(256)
a0_factors_check_kernel(const float *a0, int pitch, int row0, int nrows, int ncols,
                        const double *a0_row, const double *a0_col, int *bad)
{
  const int n = row0 + blockIdx.y;
  if (n >= row0 + nrows) return;
  const double c = a0_row[n];
  int differ = 0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < ncols; j += gridDim.x * blockDim.x) {
    const float formed = (float)(a0_col[j] * c);
    differ += __float_as_uint(formed) != __float_as_uint(a0[(size_t)n * pitch + j]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) differ += __shfl_down(differ, off, 64);
  if ((threadIdx.x & 63) == 0 && differ) atomicAdd(&bad[n], differ);
}

// div2_shared against the compiler's IEEE division: thread i forms two numerators and a
// denominator from a counter-based hash (every exponent, denormals, zeros, infinities and NaN
// among the numerators; denominators 0.5 .. 2^100 -- the range the launcher admits: below it the
// hardware scales numerator AND denominator of a tiny numerator whose quotient is not tiny, which
// a shared unscaled reciprocal cannot follow -- half of them of the form nu^2 + x^2 near 1)
// and counts the quotients whose bits differ (any NaN equals any NaN).
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void __launch_bounds__(256)
division_selftest_kernel(unsigned long long count, uint32_t seed, unsigned long long *differ, float *first_bad)
{
  unsigned long long bad = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t h1 = mix32((uint32_t)i * 3u + seed), h2 = mix32((uint32_t)i * 3u + 1u + seed),
                   h3 = mix32((uint32_t)i * 3u + 2u + seed) ^ (uint32_t)(i >> 32);
    const float n1 = __uint_as_float(h1), n2 = __uint_as_float(h2);
    float d;
    if (h3 & 1u) {
      const float x = __uint_as_float(0x30000000u + (h3 >> 1) % 0x10000000u);        // 2^-31 .. 2^1
      d = 1.0010002f + x * x;
    } else {
      d = __uint_as_float(((126u + (h3 >> 8) % 101u) << 23) | ((h3 >> 1) & 0x7fffffu)); // 0.5 .. 2^100
    }
    float q1, q2;
    div2_shared(n1, n2, d, q1, q2);
    const float r1 = n1 / d, r2 = n2 / d;
    const bool ok1 = __float_as_uint(q1) == __float_as_uint(r1) || (q1 != q1 && r1 != r1);
    const bool ok2 = __float_as_uint(q2) == __float_as_uint(r2) || (q2 != q2 && r2 != r2);
    if (!(ok1 && ok2)) {
      if (bad == 0 && atomicAdd(differ, 0ull) == 0) { first_bad[0] = ok1 ? n2 : n1; first_bad[1] = d; first_bad[2] = ok1 ? q2 : q1; first_bad[3] = ok1 ? r2 : r1; }
      bad++;
    }
  }
  if (bad) atomicAdd(differ, bad);
}

// ---------------------------------------------------------------------------
// S loop iterations (2S sweeps) in one pass: the general form of the two
// kernels above.  Stage q (0 = main sweep at t, 1 = half-step sweep at t,
// 2 = main sweep at t+dt, ...) runs at row r-q when stage 0 stands at row r.
// Its centre is the output of stage q-2 two iterations ago, its stencil the
// output of stage q-1 of this iteration (row n+1) and of two iterations ago
// (row n-1); stages 0 and 1 take the input lattices instead.  Every stage keeps
// its last two output rows in registers; only the last two stages store.
// Every stage costs one column of validity per side of a wave (the first for want of a
// neighbour lane's input, the others at each hand-off), so H = ceil(2S / V) halo lanes per
// side recompute what the neighbouring wave owns.
// pers_a/pers_b[q]: the array whose never-written cells (row N, boundary
// columns) stage q's output inherits -- where the reference's ping-pong would
// have put that lattice; for stage 2S-2 that is its own output array.
// ---------------------------------------------------------------------------
template <int S>
struct MultiArgs {
  static constexpr int Q = 2 * S;
  const float *a0, *a_c, *b_c, *a_h, *b_h;
  const float *pers_a[Q - 1], *pers_b[Q - 1];
  float *a_o, *b_o, *a_ho, *b_ho;            // outputs of stage Q-2 (main) and Q-1 (half-step)
  float *rows1, *rows2;                      // ROWS: [3][pitch] each: a rows 0, 1 and b row 1 of stage 0's / stage Q-2's output
  float E_dc, E_omega, B, dt, dPhi, PhiYmin, nu, nu2, nu_tilde, bdt;
  float c[2 * Q];                            // cosines, two per stage
  int N, pitch, m_offset, n_chunk, jj0;
  int jf[Q], jl[Q];                          // columns each stage produces
};

// V = columns per lane (4: 16-byte accesses; 2: half the registers per lane, twice the waves;
// 1: a quarter of the instructions per pass of the row loop, for lattices too narrow to fill the device)
template <int V>
__device__ __forceinline__ void widen_v(const float (&own)[V], float (&e)[V + 2])
{
  e[0] = from_left_lane(own[V - 1]);
#pragma unroll
  for (int k = 0; k < V; k++) e[k + 1] = own[k];
  e[V + 1] = from_right_lane(own[0]);
}

// ROWS (as in the double step): the three rows av() reads of the first and of the last new main
// lattice also go to small side buffers.
template <int S, int V, int MINW, bool DIAG = false, bool ROWS = false>
__global__ void __launch_bounds__(256, MINW)
multi_step_kernel(const MultiArgs<S> p)
{
  static_assert(V == 1 || V == 2 || V == 4, "one, two or four columns per lane");
  constexpr int Q = 2 * S, H = (Q + V - 1) / V, INNER = 64 - 2 * H;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int jj   = p.jj0 + (wave * INNER + lane - H) * V;       // first own column (may be < 0)
  int j_hi = p.jl[0];
#pragma unroll
  for (int q = 1; q < Q; q++) j_hi = p.jl[q] > j_hi ? p.jl[q] : j_hi;
  if (p.jj0 + wave * INNER * V > j_hi) return;                  // whole wave past the end
  const bool interior = lane >= H && lane < 64 - H;
  const int jcl = jj < 0 ? 0 : (jj > p.pitch - V ? p.pitch - V : jj);
  const unsigned j = (unsigned)jcl;

  const int n0 = blockIdx.y * p.n_chunk;
  const int n1 = n0 + p.n_chunk < p.N ? n0 + p.n_chunk : p.N;
  const int a_lo  = n0 > Q - 1 ? n0 - (Q - 1) : 0;              // first row of stage 0
  const int r_end = n1 + Q - 2;                                 // last stage forms row n1-1 then

  float bphi[V];
  unsigned ok = 0;          // bit q*V+k: stage q produces column k (stages 0..Q-2)
  unsigned wr = 0;          // bit k: stage Q-2 stores column k; bit V+k: stage Q-1 stores it
  unsigned need = 0;        // bit q: this lane must fetch never-written cells for stage q
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int c = jj + k;
    bphi[k] = p.B * (p.PhiYmin + p.dPhi * (float)(p.m_offset + c - 1));
    const bool in_row = c >= 0 && c < p.pitch;
#pragma unroll
    for (int q = 0; q < Q - 1; q++) {
      const bool okq = c >= p.jf[q] && c <= p.jl[q];
      if (okq) ok |= 1u << (q * V + k);
      // a column the next stage reads at m+-1 that this stage does not produce keeps its old value
      if (!okq && in_row && c >= p.jf[q + 1] - 1 && c <= p.jl[q + 1] + 1) need |= 1u << q;
    }
    if (interior && c >= p.jf[Q - 2] && c <= p.jl[Q - 2]) wr |= 1u << k;
    if (interior && c >= p.jf[Q - 1] && c <= p.jl[Q - 1]) wr |= 1u << (V + k);
  }
  if (jj != jcl) need = 0;
  float sc[2 * Q];
#pragma unroll
  for (int q = 0; q < 2 * Q; q++) sc[q] = p.E_dc + p.E_omega * p.c[q];

  const size_t P = (size_t)p.pitch;
  auto load4 = [&](const float *base, size_t row, float (&v)[V]) {
    if constexpr (V == 4) {
      const float4 x = *reinterpret_cast<const float4 *>(base + row + j);
      v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else if constexpr (V == 2) {
      const float2 x = *reinterpret_cast<const float2 *>(base + row + j);
      v[0] = x.x; v[1] = x.y;
    } else {
      v[0] = base[row + j];
    }
  };
  auto store_all = [&](float *base, size_t row, const float (&v)[V]) {
    if constexpr (V == 4)      *reinterpret_cast<float4 *>(base + row + j) = make_float4(v[0], v[1], v[2], v[3]);
    else if constexpr (V == 2) *reinterpret_cast<float2 *>(base + row + j) = make_float2(v[0], v[1]);
    else                       base[row + j] = v[0];
  };
  constexpr unsigned ALL = (1u << V) - 1u;
  auto diffs = [&](const float (&own)[V], float (&d)[V]) {
    float e[V + 2];
    widen_v<V>(own, e);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] = e[k + 2] - e[k];
  };

  // histories: index 0 = one iteration ago, 1 = two iterations ago
  float hA[2][V] = {}, hB[2][V] = {};               // input half-step lattice rows r, r-1
  float oA[Q - 1][2][V] = {}, oB[Q - 1][2][V] = {}; // stage outputs
  float a0h[Q - 1][V] = {};                         // a0 rows r-1 .. r-(Q-1)

  if (a_lo >= 1) { load4(p.a_h, (size_t)(a_lo - 1) * P, hA[1]); load4(p.b_h, (size_t)(a_lo - 1) * P, hB[1]); }
  load4(p.a_h, (size_t)a_lo * P, hA[0]); load4(p.b_h, (size_t)a_lo * P, hB[0]);

  for (int r = a_lo; r <= r_end; r++) {
    float nA[Q][V] = {}, nB[Q][V] = {};             // this iteration's stage outputs
    float a0r[V] = {}, inA[V] = {}, inB[V] = {}, upA0[V] = {}, upB0[V] = {};
    if (r < p.N) {
      const size_t row = (size_t)r * P;
      load4(p.a0, row, a0r); load4(p.a_c, row, inA); load4(p.b_c, row, inB);
      load4(p.a_h, row + P, upA0); load4(p.b_h, row + P, upB0);
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int n = r - q;
      const int lo_q = n0 > Q - 1 - q ? n0 - (Q - 1 - q) : 0;
      if (n >= lo_q && n < p.N) {
        float ea[V + 2], eb[V + 2], dla[V], dlb[V];
        // stencil: row n+1 (this iteration) with its m+-1 neighbours, row n-1 as differences
        if (q == 0) { widen_v<V>(upA0, ea); widen_v<V>(upB0, eb); diffs(hA[1], dla); diffs(hB[1], dlb); }
        else        { widen_v<V>(nA[q - 1], ea); widen_v<V>(nB[q - 1], eb); diffs(oA[q - 1][1], dla); diffs(oB[q - 1][1], dlb); }
        const float fn = (float)n;
#pragma unroll
        for (int k = 0; k < V; k++) {
          const float ca = q == 0 ? inA[k] : (q == 1 ? hA[1][k] : oA[q - 2][1][k]);
          const float cb = q == 0 ? inB[k] : (q == 1 ? hB[1][k] : oB[q - 2][1][k]);
          const float a0v = q == 0 ? a0r[k] : a0h[q - 1][k];
          const float mu_a = (sc[2 * q] + bphi[k]) * p.dt / 2.0f, mu_b = (sc[2 * q + 1] + bphi[k]) * p.dt / 2.0f;
          point_update_sel<DIAG>(p, n, fn, mu_a, mu_b, a0v, ca, cb,
                                 ea[k + 2], ea[k], eb[k + 2], eb[k], dla[k], dlb[k], nA[q][k], nB[q][k]);
        }
        const size_t row = (size_t)n * P;
        if (q >= Q - 2 && n >= n0 && n < n1) {
          float *da = q == Q - 2 ? p.a_o : p.a_ho, *db = q == Q - 2 ? p.b_o : p.b_ho;
          const unsigned w = q == Q - 2 ? (wr & ALL) : (wr >> V);
          if (w == ALL) {
            store_all(da, row, nA[q]);
            if (n > 0) store_all(db, row, nB[q]);
          } else {
#pragma unroll
            for (int k = 0; k < V; k++) {
              if (w & (1u << k)) {
                da[row + j + k] = nA[q][k];
                if (n > 0) db[row + j + k] = nB[q][k];
              }
            }
          }
        }
        if constexpr (ROWS) {
          // stage 0's rows come from the chunk that starts at row 0 (no chunk stores them), the
          // last main lattice's from the chunk that stores them
          if ((q == 0 || q == Q - 2) && n <= 1 && interior && (q == 0 ? n0 == 0 : (n >= n0 && n < n1))) {
            float *rows = q == 0 && Q > 2 ? p.rows1 : p.rows2;
#pragma unroll
            for (int k = 0; k < V; k++) {
              if (ok & (1u << (q * V + k))) {
                rows[row + j + k] = nA[q][k];
                if (n == 1) rows[2 * P + j + k] = nB[q][k];
              }
            }
          }
        }
        if (q < Q - 1) {
          if (need & (1u << q)) {
            float xa[V], xb[V];
            load4(p.pers_a[q], row, xa); load4(p.pers_b[q], row, xb);
#pragma unroll
            for (int k = 0; k < V; k++)
              if (!(ok & (1u << (q * V + k)))) { nA[q][k] = xa[k]; nB[q][k] = xb[k]; }
          }
        }
      } else if (q < Q - 1 && n == p.N) {
        // row N is never written: the stage's "output" there is what its buffer holds
        load4(p.pers_a[q], (size_t)n * P, nA[q]); load4(p.pers_b[q], (size_t)n * P, nB[q]);
        if constexpr (ROWS) {
          // a single harmonic: row 1 is the row no sweep writes, and av() reads it all the same
          if ((q == 0 || q == Q - 2) && n == 1 && n0 == 0 && interior) {
            float *rows = q == 0 && Q > 2 ? p.rows1 : p.rows2;
#pragma unroll
            for (int k = 0; k < V; k++) {
              if (ok & (1u << (q * V + k))) { rows[P + j + k] = nA[q][k]; rows[2 * P + j + k] = nB[q][k]; }
            }
          }
        }
      }
    }
    // ---- slide every history one row up ----
#pragma unroll
    for (int k = 0; k < V; k++) {
#pragma unroll
      for (int q = 0; q < Q - 1; q++) {
        oA[q][1][k] = oA[q][0][k]; oB[q][1][k] = oB[q][0][k];
        oA[q][0][k] = nA[q][k];    oB[q][0][k] = nB[q][k];
      }
#pragma unroll
      for (int q = Q - 2; q >= 1; q--) a0h[q][k] = a0h[q - 1][k];
      a0h[0][k] = a0r[k];
      hA[1][k] = hA[0][k]; hB[1][k] = hB[0][k]; hA[0][k] = upA0[k]; hB[0][k] = upB0[k];
    }
  }
}

// ---------------------------------------------------------------------------
// The general S-steps-per-launch kernel with everything the double step has learnt: four
// columns per lane, H halo lanes per side (H*4 >= 2S columns), the launch schedule of
// fused2_step_kernel (whole column strips first, the rest cut into chunks of rows, neighbouring
// strips on one XCD), non-temporal stores, the two-columns-wide point update with the shared
// division (DIV2), and optionally (PF) the loads of stage 0's next row issued one pass of the
// row loop ahead, so that a wave covers its own memory latency -- which matters at S = 3, where
// registers allow two waves per SIMD only.  Same stages and hand-offs as multi_step_kernel,
// same bits.
// ---------------------------------------------------------------------------
template <int S>
struct PipeArgs {
  static constexpr int Q = 2 * S;
  const float *a0, *a_c, *b_c, *a_h, *b_h;
  const float *pers_a[Q - 1], *pers_b[Q - 1];
  float *a_o, *b_o, *a_ho, *b_ho;
  float E_dc, E_omega, B, dt, dPhi, PhiYmin, nu, nu2, nu_tilde, bdt;
  float c[2 * Q];
  int N, pitch, m_offset, jj0;
  int jf[Q], jl[Q];
  int n_phase, wg_begin[4], g0[3], ng[3], len[3];      // as in Fused2Args
  int edge_last;                                       // > 0: phase 0 holds two strip groups, the first and this one
};

// RING: the stage outputs wait in LDS instead of registers.  A row a stage forms in pass r of
// the row loop is used at once (as the next stage's row n+1) and again two passes later (as that
// stage's row n-1 and as the centre row of the stage after it); in between it is only kept.
// Each wave owns a ring of two slots per stage and array in LDS -- 64 lanes x 16 bytes per slot,
// written and read with 16-byte LDS accesses, no barrier since no other wave ever sees it --
// and a pass reads the slot it is about to overwrite.  That takes 16 (2S-1) registers per lane
// out of the kernel (80 at S = 3, where they do not fit otherwise) and the moves that slide
// the histories with them; it costs 4 (2S-1) KB of LDS per wave.
template <int S, int MINW, int H, bool NTS, bool DIV2, bool PF, bool RING = false>
__global__ void __launch_bounds__(256, MINW)
pipe_step_kernel(const PipeArgs<S> p)
{
  constexpr int V = 4, Q = 2 * S, INNER = 64 - 2 * H;
  static_assert(H * V >= Q, "halo lanes must cover one column per stage");
  const int lane = threadIdx.x & 63;
  int id = blockIdx.x, ph = 0;
  if (p.n_phase > 1 && id >= p.wg_begin[1]) ph = 1;
  if (p.n_phase > 2 && id >= p.wg_begin[2]) ph = 2;
  id -= p.wg_begin[ph];
  const int ng = p.ng[ph];
  const int chunk = id / ng;
  int bx = id - chunk * ng;
  const int n0 = chunk * p.len[ph];
  if (n0 >= p.N) return;                                        // padding of the phase
  const int per_xcd = ng >> 3;
  if (bx < (per_xcd << 3)) bx = (bx & 7) * per_xcd + (bx >> 3);
  bx += p.g0[ph];
  const int wave = (bx * blockDim.x + threadIdx.x) >> 6;
  const int jj   = p.jj0 + (wave * INNER + lane - H) * V;       // first own column (may be < 0)
  int j_hi = p.jl[0];
#pragma unroll
  for (int q = 1; q < Q; q++) j_hi = p.jl[q] > j_hi ? p.jl[q] : j_hi;
  if (p.jj0 + wave * INNER * V > j_hi) return;                  // whole wave past the end
  const bool interior = lane >= H && lane < 64 - H;
  const int jcl = jj < 0 ? 0 : (jj > p.pitch - V ? p.pitch - V : jj);
  const unsigned j = (unsigned)jcl;

  const int n1 = n0 + p.len[ph] < p.N ? n0 + p.len[ph] : p.N;
  const int a_lo  = n0 > Q - 1 ? n0 - (Q - 1) : 0;              // first row of stage 0
  const int r_end = n1 + Q - 2;                                 // last stage forms row n1-1 then

  float bphi[V];
  unsigned ok = 0;          // bit q*V+k: stage q produces column k (stages 0..Q-2)
  unsigned wr = 0;          // bit k: stage Q-2 stores column k; bit V+k: stage Q-1 stores it
  unsigned need = 0;        // bit q: this lane must fetch never-written cells for stage q
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int c = jj + k;
    bphi[k] = p.B * (p.PhiYmin + p.dPhi * (float)(p.m_offset + c - 1));
    const bool in_row = c >= 0 && c < p.pitch;
#pragma unroll
    for (int q = 0; q < Q - 1; q++) {
      const bool okq = c >= p.jf[q] && c <= p.jl[q];
      if (okq) ok |= 1u << (q * V + k);
      if (!okq && in_row && c >= p.jf[q + 1] - 1 && c <= p.jl[q + 1] + 1) need |= 1u << q;
    }
    if (interior && c >= p.jf[Q - 2] && c <= p.jl[Q - 2]) wr |= 1u << k;
    if (interior && c >= p.jf[Q - 1] && c <= p.jl[Q - 1]) wr |= 1u << (V + k);
  }
  if (jj != jcl) need = 0;
  float sc[2 * Q];
#pragma unroll
  for (int q = 0; q < 2 * Q; q++) sc[q] = p.E_dc + p.E_omega * p.c[q];

  const size_t P = (size_t)p.pitch;
  auto load4 = [&](const float *base, size_t row, float (&v)[V]) {
    const float4 x = *reinterpret_cast<const float4 *>(base + row + j);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  };
  auto store4 = [&](float *base, size_t row, const float (&v)[V]) {
    float4 *at = reinterpret_cast<float4 *>(base + row + j);
    if constexpr (NTS) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 q = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(at));
    } else {
      *at = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  constexpr unsigned ALL = (1u << V) - 1u;
  auto diffs = [&](const float (&own)[V], float (&d)[V]) {
    float e[V + 2];
    widen(own, e);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] = e[k + 2] - e[k];
  };

  // histories: index 0 = one pass ago, 1 = two passes ago
  float hA[2][V] = {}, hB[2][V] = {};               // input half-step lattice rows r, r-1
  constexpr int QH = RING ? 1 : Q - 1;
  float oA[QH][2][V] = {}, oB[QH][2][V] = {};       // stage outputs (not RING)
  float a0h[Q - 1][V] = {};                         // dt * a0, rows r-1 .. r-(Q-1)
  // RING: slot (parity of the pass, stage, array) of this wave, one float4 per lane
  constexpr int RING_SLOTS = RING ? 4 * (Q - 1) : 1;
  __shared__ float4 ring_mem[RING ? 4 * RING_SLOTS * 64 : 1];
  float4 *ring = ring_mem + ((threadIdx.x >> 6) * RING_SLOTS) * 64 + lane;
  if constexpr (RING) {
#pragma unroll
    for (int i = 0; i < RING_SLOTS; i++) ring[i * 64] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  auto ring_put = [&](int par, int q, int arr, const float (&v)[V]) {
    ring[((par * (Q - 1) + q) * 2 + arr) * 64] = make_float4(v[0], v[1], v[2], v[3]);
  };
  auto ring_get = [&](int par, int q, int arr, float (&v)[V]) {
    const float4 x = ring[((par * (Q - 1) + q) * 2 + arr) * 64];
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  };

  if (a_lo >= 1) { load4(p.a_h, (size_t)(a_lo - 1) * P, hA[1]); load4(p.b_h, (size_t)(a_lo - 1) * P, hB[1]); }
  load4(p.a_h, (size_t)a_lo * P, hA[0]); load4(p.b_h, (size_t)a_lo * P, hB[0]);

  struct RowIn { float a0[V], ac[V], bc[V], upA[V], upB[V]; };
  auto load_row = [&](int r, RowIn &in) {
    const size_t row = (size_t)r * P;
    load4(p.a0, row, in.a0); load4(p.a_c, row, in.ac); load4(p.b_c, row, in.bc);
    load4(p.a_h, row + P, in.upA); load4(p.b_h, row + P, in.upB);
  };
  RowIn cur = {}, nxt = {};
  if constexpr (PF) { if (a_lo < p.N) load_row(a_lo, cur); }

  for (int r = a_lo; r <= r_end; r++) {
    float nA[Q][V] = {}, nB[Q][V] = {};             // this pass's stage outputs
    const int par = (r - a_lo) & 1;
    // RING: the rows the stages left two passes ago (stage q-1's for stage q's row n-1, stage
    // q-2's for its centre), fetched just before the first stage that needs them
    float oldA[Q][V] = {}, oldB[Q][V] = {};
    if constexpr (PF) {
      if (r + 1 < p.N && r + 1 <= r_end) load_row(r + 1, nxt);
    } else {
      if (r < p.N) load_row(r, cur);
    }
    float a0r[V];
#pragma unroll
    for (int k = 0; k < V; k++) a0r[k] = p.dt * cur.a0[k];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int n = r - q;
      const int lo_q = n0 > Q - 1 - q ? n0 - (Q - 1 - q) : 0;
      if constexpr (RING) {
        // what this stage left two passes ago is wanted by the NEXT stage (and the one after it):
        // asked for a stage ahead, so that the LDS latency passes behind this stage's arithmetic
        if (q <= Q - 2) { ring_get(par, q, 0, oldA[q]); ring_get(par, q, 1, oldB[q]); }
      }
      if (n >= lo_q && n < p.N) {
        float ea[V + 2], eb[V + 2], la[V], lb[V];
        // stencil: row n+1 (this pass) with its m+-1 neighbours, row n-1 as differences
        if (q == 0) { widen(cur.upA, ea); widen(cur.upB, eb); diffs(hA[1], la); diffs(hB[1], lb); }
        else if constexpr (RING) { widen(nA[q - 1], ea); widen(nB[q - 1], eb); diffs(oldA[q >= 1 ? q - 1 : 0], la); diffs(oldB[q >= 1 ? q - 1 : 0], lb); }
        else        { widen(nA[q - 1], ea); widen(nB[q - 1], eb); diffs(oA[q >= 1 ? q - 1 : 0][1], la); diffs(oB[q >= 1 ? q - 1 : 0][1], lb); }
        if (n < 2) {                               // the same for the whole wave; rows 0 and 1 only
          asm volatile("" ::);
#pragma unroll
          for (int k = 0; k < V; k++) { lb[k] = 0.0f; la[k] = n == 0 ? 0.0f : 2.0f * la[k]; }
        }
        const float fn = (float)n;
#pragma unroll
        for (int k = 0; k < V; k += 2) {
          slb_f2 ca, cb;
          if (q == 0)      { ca = slb_f2{cur.ac[k], cur.ac[k + 1]}; cb = slb_f2{cur.bc[k], cur.bc[k + 1]}; }
          else if (q == 1) { ca = slb_f2{hA[1][k], hA[1][k + 1]};   cb = slb_f2{hB[1][k], hB[1][k + 1]}; }
          else if constexpr (RING) {
            ca = slb_f2{oldA[q >= 2 ? q - 2 : 0][k], oldA[q >= 2 ? q - 2 : 0][k + 1]};
            cb = slb_f2{oldB[q >= 2 ? q - 2 : 0][k], oldB[q >= 2 ? q - 2 : 0][k + 1]};
          } else {
            ca = slb_f2{oA[q >= 2 && !RING ? q - 2 : 0][1][k], oA[q >= 2 && !RING ? q - 2 : 0][1][k + 1]};
            cb = slb_f2{oB[q >= 2 && !RING ? q - 2 : 0][1][k], oB[q >= 2 && !RING ? q - 2 : 0][1][k + 1]};
          }
          const slb_f2 a0v = q == 0 ? slb_f2{a0r[k], a0r[k + 1]} : slb_f2{a0h[q >= 1 ? q - 1 : 0][k], a0h[q >= 1 ? q - 1 : 0][k + 1]};
          const slb_f2 bp = {bphi[k], bphi[k + 1]};
          const slb_f2 mu_a = (sc[2 * q] + bp) * p.dt / 2.0f, mu_b = (sc[2 * q + 1] + bp) * p.dt / 2.0f;
          slb_f2 qa, qb;
          point_update_pair<DIV2>(p, fn, mu_a, mu_b, a0v, ca, cb, slb_f2{ea[k + 2], ea[k + 3]}, slb_f2{ea[k], ea[k + 1]},
                                  slb_f2{eb[k + 2], eb[k + 3]}, slb_f2{eb[k], eb[k + 1]},
                                  slb_f2{la[k], la[k + 1]}, slb_f2{lb[k], lb[k + 1]}, qa, qb);
          nA[q][k] = qa.x; nA[q][k + 1] = qa.y; nB[q][k] = qb.x; nB[q][k + 1] = qb.y;
        }
        const size_t row = (size_t)n * P;
        if (q >= Q - 2 && n >= n0 && n < n1) {
          float *da = q == Q - 2 ? p.a_o : p.a_ho, *db = q == Q - 2 ? p.b_o : p.b_ho;
          const unsigned w = q == Q - 2 ? (wr & ALL) : (wr >> V);
          if (w == ALL) {
            store4(da, row, nA[q]);
            if (n > 0) store4(db, row, nB[q]);
          } else {
#pragma unroll
            for (int k = 0; k < V; k++) {
              if (w & (1u << k)) {
                da[row + j + k] = nA[q][k];
                if (n > 0) db[row + j + k] = nB[q][k];
              }
            }
          }
        }
        if (q < Q - 1) {
          if (need & (1u << q)) {
            float xa[V], xb[V];
            load4(p.pers_a[q], row, xa); load4(p.pers_b[q], row, xb);
#pragma unroll
            for (int k = 0; k < V; k++)
              if (!(ok & (1u << (q * V + k)))) { nA[q][k] = xa[k]; nB[q][k] = xb[k]; }
          }
          if constexpr (RING) { ring_put(par, q, 0, nA[q]); ring_put(par, q, 1, nB[q]); }
        }
      } else if (q < Q - 1 && n == p.N) {
        // row N is never written: the stage's "output" there is what its buffer holds
        load4(p.pers_a[q], (size_t)n * P, nA[q]); load4(p.pers_b[q], (size_t)n * P, nB[q]);
      }
    }
    // ---- slide every history one row up ----
#pragma unroll
    for (int k = 0; k < V; k++) {
      if constexpr (!RING) {
#pragma unroll
        for (int q = 0; q < Q - 1; q++) {
          oA[q][1][k] = oA[q][0][k]; oB[q][1][k] = oB[q][0][k];
          oA[q][0][k] = nA[q][k];    oB[q][0][k] = nB[q][k];
        }
      }
#pragma unroll
      for (int q = Q - 2; q >= 1; q--) a0h[q][k] = a0h[q - 1][k];
      a0h[0][k] = a0r[k];
      hA[1][k] = hA[0][k]; hB[1][k] = hB[0][k]; hA[0][k] = cur.upA[k]; hB[0][k] = cur.upB[k];
    }
    if constexpr (PF) cur = nxt;
  }
}

// ---------------------------------------------------------------------------
// ring_step_kernel: pipe_step_kernel's RING form with the row loop written twice.
//
// What held the ring kernel back at three steps per launch was not its arithmetic but that a
// wave stood still for the whole memory latency at the top of every pass: the loop body held
// loads under conditions (never-written cells for the lanes at the lattice's edges, row N at the
// end of a strip, single-column stores at a ragged right edge), so every wait in it had to be a
// wait for ALL outstanding accesses -- the previous pass's stores included -- and loads issued a
// pass ahead could not stay in flight.  Here a wave first asks whether any of its lanes ever
// needs those cases (only the waves at the two ends of the lattice do); if none does, every pass up to
// the one that meets row N runs a body whose only branches are on the row number (rows 0 and 1) and
// hold no loads: the five input rows of pass r+1 are requested at the top of pass r (the row loop is
// unrolled by two, so the two register sets just alternate and the ring slots have fixed
// addresses), stores of halo lanes are dropped by the buffer bounds check (their offset lies
// beyond the row's end) instead of by a branch, and the only wait for memory is the one that lets
// the four stores of the pass before stay outstanding.  The last passes of a strip (row N is fetched,
// stages run out of rows) and all passes of the edge waves run the general body (the RING body of
// pipe_step_kernel).
// Same stages, same hand-offs, same point update: same bits.
//
// ROWS: as in the double step, rows 0 and 1 of a and row 1 of b of every new main lattice
// (stages 0, 2, .., 2S-2) also go to small side buffers for av(); only general passes form them.
// ---------------------------------------------------------------------------
template <int S>
struct RingArgs {
  PipeArgs<S> p;
  float *rows[S];                                      // ROWS: [3][pitch] each
  unsigned long long *stamps;                          // timing builds (X == 4): per wave start, end, HW_ID, XCC_ID
};

typedef unsigned int slb_u4 __attribute__((vector_size(16)));
typedef float slb_f4 __attribute__((vector_size(16)));

template <int S, int MINW, bool ROWS = false, bool DIV2 = true, int X = 0>
__global__ void __launch_bounds__(256, MINW)
ring_step_kernel(const RingArgs<S> ra)
{
  const PipeArgs<S> &p = ra.p;
  constexpr int V = 4, Q = 2 * S, H = 2, INNER = 64 - 2 * H;
  static_assert(H * V >= Q, "halo lanes must cover one column per stage");
  const int lane = threadIdx.x & 63;
  int id = blockIdx.x, ph = 0;
  if (p.n_phase > 1 && id >= p.wg_begin[1]) ph = 1;
  if (p.n_phase > 2 && id >= p.wg_begin[2]) ph = 2;
  id -= p.wg_begin[ph];
  const int ng = p.ng[ph];
  const int chunk = id / ng;
  int bx = id - chunk * ng;
  const int n0 = chunk * p.len[ph];
  if (n0 >= p.N) return;                                        // padding of the phase
  const int per_xcd = ng >> 3;
  if (bx < (per_xcd << 3)) bx = (bx & 7) * per_xcd + (bx >> 3);
  bx += p.g0[ph];
  if (ph == 0 && p.edge_last > 0 && bx > 0) bx = p.edge_last;
  const int wave = __builtin_amdgcn_readfirstlane((bx * blockDim.x + threadIdx.x) >> 6);
  const int jj   = p.jj0 + (wave * INNER + lane - H) * V;       // first own column (may be < 0)
  int j_hi = p.jl[0];
#pragma unroll
  for (int q = 1; q < Q; q++) j_hi = p.jl[q] > j_hi ? p.jl[q] : j_hi;
  if (p.jj0 + wave * INNER * V > j_hi) return;                  // whole wave past the end
  unsigned long long t_start = 0;
  if constexpr (X == 4) t_start = __builtin_amdgcn_s_memrealtime();
  const bool interior = lane >= H && lane < 64 - H;
  const int jcl = jj < 0 ? 0 : (jj > p.pitch - V ? p.pitch - V : jj);
  const unsigned j = (unsigned)jcl;

  const int n1 = n0 + p.len[ph] < p.N ? n0 + p.len[ph] : p.N;
  const int a_lo  = n0 > Q - 1 ? n0 - (Q - 1) : 0;              // first row of stage 0
  const int r_end = n1 + Q - 2;                                 // last stage forms row n1-1 then

  float bphi[V];
  unsigned ok = 0;          // bit q*V+k: stage q produces column k (stages 0..Q-2)
  unsigned wr = 0;          // bit k: stage Q-2 stores column k; bit V+k: stage Q-1 stores it
  unsigned need = 0;        // bit q: this lane must fetch never-written cells for stage q
#pragma unroll
  for (int k = 0; k < V; k++) {
    const int c = jj + k;
    bphi[k] = p.B * (p.PhiYmin + p.dPhi * (float)(p.m_offset + c - 1));
    const bool in_row = c >= 0 && c < p.pitch;
#pragma unroll
    for (int q = 0; q < Q - 1; q++) {
      const bool okq = c >= p.jf[q] && c <= p.jl[q];
      if (okq) ok |= 1u << (q * V + k);
      if (!okq && in_row && c >= p.jf[q + 1] - 1 && c <= p.jl[q + 1] + 1) need |= 1u << q;
    }
    if (interior && c >= p.jf[Q - 2] && c <= p.jl[Q - 2]) wr |= 1u << k;
    if (interior && c >= p.jf[Q - 1] && c <= p.jl[Q - 1]) wr |= 1u << (V + k);
  }
  if (jj != jcl) need = 0;
  float sc[2 * Q];
#pragma unroll
  for (int q = 0; q < 2 * Q; q++) sc[q] = p.E_dc + p.E_omega * p.c[q];

  const size_t P = (size_t)p.pitch;
  auto load4 = [&](const float *base, size_t row, float (&v)[V]) {
    const float4 x = *reinterpret_cast<const float4 *>(base + row + j);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  };
  auto store4 = [&](float *base, size_t row, const float (&v)[V]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 q = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(base + row + j));
  };
  // branch-free accesses of the steady passes: a descriptor per row (base of the row, its length
  // in bytes), the lane's byte offset in the row; an offset beyond the row's end drops a store
  constexpr int kRsrcFlags = 0x00020000;                       // raw 32-bit data, no swizzle
  const int row_bytes = p.pitch * 4;
  const int off_ld = (int)(j * 4u);
  const int off_st = (wr == 0xffu) ? off_ld : (int)0x80000000u;
  auto bload4 = [&](const float *base, size_t row, float (&v)[V]) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + row), 0, row_bytes, kRsrcFlags);
    // (the whole vector is re-typed at once: __builtin_bit_cast of a single vector element reads element 0)
    const slb_f4 x = __builtin_bit_cast(slb_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off_ld, 0, 0));
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  };
  auto bstore4 = [&](float *base, size_t row, const float (&v)[V], bool live) {
    // (a row that is not to be stored gets a descriptor of length zero: every lane's store is dropped)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + row), 0, live ? row_bytes : 0, kRsrcFlags);
    const slb_f4 x = {v[0], v[1], v[2], v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(slb_u4, x), rs, off_st, 0, 2);    // 2: non-temporal
  };
  constexpr unsigned ALL = (1u << V) - 1u;
  auto diffs = [&](const float (&own)[V], float (&d)[V]) {
    float e[V + 2];
    widen(own, e);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] = e[k + 2] - e[k];
  };

  float hA[2][V] = {}, hB[2][V] = {};               // input half-step lattice rows r, r-1
  float a0h[Q - 1][V] = {};                         // dt * a0, rows r-1 .. r-(Q-1)
  // slot (parity of the pass, stage, array) of this wave's ring, one float4 per lane
  constexpr int RING_SLOTS = 4 * (Q - 1);
  __shared__ float4 ring_mem[4 * RING_SLOTS * 64];
  float4 *ring = ring_mem + ((threadIdx.x >> 6) * RING_SLOTS) * 64 + lane;
#pragma unroll
  for (int i = 0; i < RING_SLOTS; i++) ring[i * 64] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  auto ring_put = [&](int par, int q, int arr, const float (&v)[V]) {
    ring[((par * (Q - 1) + q) * 2 + arr) * 64] = make_float4(v[0], v[1], v[2], v[3]);
  };
  auto ring_get = [&](int par, int q, int arr, float (&v)[V]) {
    const float4 x = ring[((par * (Q - 1) + q) * 2 + arr) * 64];
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  };

  if (a_lo >= 1) { load4(p.a_h, (size_t)(a_lo - 1) * P, hA[1]); load4(p.b_h, (size_t)(a_lo - 1) * P, hB[1]); }
  load4(p.a_h, (size_t)a_lo * P, hA[0]); load4(p.b_h, (size_t)a_lo * P, hB[0]);

  struct RowIn { float a0[V], ac[V], bc[V], upA[V], upB[V]; };
  auto load_row = [&](int r, RowIn &in) {
    const size_t row = (size_t)r * P;
    load4(p.a0, row, in.a0); load4(p.a_c, row, in.ac); load4(p.b_c, row, in.bc);
    load4(p.a_h, row + P, in.upA); load4(p.b_h, row + P, in.upB);
  };
  auto bload_row = [&](int r, RowIn &in) {
    const size_t row = (size_t)r * P;
    bload4(p.a0, row, in.a0); bload4(p.a_c, row, in.ac); bload4(p.b_c, row, in.bc);
    bload4(p.a_h, row + P, in.upA); bload4(p.b_h, row + P, in.upB);
  };

  // one pass of the row loop: stage q forms row r-q.  ST: the branch-free form (every stage
  // active and past row 1, both last stages store whole float4s or nothing, no never-written
  // cells, PAR = r & 1 known when compiling, the rows of pass r+1 requested into `nxt`).
  auto pass = [&](auto st_tag, auto par_tag, int r, RowIn &cur, RowIn &nxt) {
    constexpr bool ST = decltype(st_tag)::value;
    const int par = ST ? (int)decltype(par_tag)::value : (r & 1);
    float nA[Q][V] = {}, nB[Q][V] = {};             // this pass's stage outputs
    float oldA[Q][V] = {}, oldB[Q][V] = {};         // what the stages left two passes ago
    if constexpr (ST && X != 1) bload_row(r + 1, nxt);
    else if constexpr (ST) bload_row(r, cur);
    else if (r < p.N) load_row(r, cur);
    float a0r[V];
#pragma unroll
    for (int k = 0; k < V; k++) a0r[k] = p.dt * cur.a0[k];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int n = r - q;
      const int lo_q = n0 > Q - 1 - q ? n0 - (Q - 1 - q) : 0;
      // what this stage left two passes ago is wanted by the NEXT stage (and the one after it):
      // asked for a stage ahead, so that the LDS latency passes behind this stage's arithmetic
      if (q <= Q - 2) { ring_get(par, q, 0, oldA[q]); ring_get(par, q, 1, oldB[q]); }
      if (ST || (n >= lo_q && n < p.N)) {
        float ea[V + 2], eb[V + 2], la[V], lb[V];
        // stencil: row n+1 (this pass) with its m+-1 neighbours, row n-1 as differences
        if (q == 0) { widen(cur.upA, ea); widen(cur.upB, eb); diffs(hA[1], la); diffs(hB[1], lb); }
        else        { widen(nA[q >= 1 ? q - 1 : 0], ea); widen(nB[q >= 1 ? q - 1 : 0], eb);
                      diffs(oldA[q >= 1 ? q - 1 : 0], la); diffs(oldB[q >= 1 ? q - 1 : 0], lb); }
        if (n < 2) {                               // the same for the whole wave; rows 0 and 1 only (and rows < 0 of a strip's first passes)
          asm volatile("" ::);
#pragma unroll
          for (int k = 0; k < V; k++) { lb[k] = 0.0f; la[k] = n == 0 ? 0.0f : 2.0f * la[k]; }
        }
        const float fn = (float)n;
#pragma unroll
        for (int k = 0; k < V; k += 2) {
          slb_f2 ca, cb;
          if (q == 0)      { ca = slb_f2{cur.ac[k], cur.ac[k + 1]}; cb = slb_f2{cur.bc[k], cur.bc[k + 1]}; }
          else if (q == 1) { ca = slb_f2{hA[1][k], hA[1][k + 1]};   cb = slb_f2{hB[1][k], hB[1][k + 1]}; }
          else {
            ca = slb_f2{oldA[q >= 2 ? q - 2 : 0][k], oldA[q >= 2 ? q - 2 : 0][k + 1]};
            cb = slb_f2{oldB[q >= 2 ? q - 2 : 0][k], oldB[q >= 2 ? q - 2 : 0][k + 1]};
          }
          const slb_f2 a0v = q == 0 ? slb_f2{a0r[k], a0r[k + 1]} : slb_f2{a0h[q >= 1 ? q - 1 : 0][k], a0h[q >= 1 ? q - 1 : 0][k + 1]};
          // (the field terms are the same in every pass: in the branch-free body the optimiser carries
          // them across the loop in 8 S registers -- 9 % fewer vector instructions per pass, and the
          // kernel's time follows its vector instructions, not its bytes)
          const slb_f2 bp = {bphi[k], bphi[k + 1]};
          const slb_f2 mu_a = (sc[2 * q] + bp) * p.dt / 2.0f, mu_b = (sc[2 * q + 1] + bp) * p.dt / 2.0f;
          slb_f2 qa, qb;
          point_update_pair<DIV2>(p, fn, mu_a, mu_b, a0v, ca, cb, slb_f2{ea[k + 2], ea[k + 3]}, slb_f2{ea[k], ea[k + 1]},
                                  slb_f2{eb[k + 2], eb[k + 3]}, slb_f2{eb[k], eb[k + 1]},
                                  slb_f2{la[k], la[k + 1]}, slb_f2{lb[k], lb[k + 1]}, qa, qb);
          nA[q][k] = qa.x; nA[q][k + 1] = qa.y; nB[q][k] = qb.x; nB[q][k + 1] = qb.y;
        }
        const size_t row = (size_t)(n > 0 ? n : 0) * P;
        if constexpr (ST) {
          // (one stage at a time: left to itself the scheduler interleaves the stages of the branch-free
          // body for parallelism it has no registers for)
          __builtin_amdgcn_sched_barrier(0);
          if (q >= Q - 2 && X != 2) {
            const bool mine = n >= n0 && n < n1;       // (a strip's first passes form rows before its own, or before row 0)
            bstore4(q == Q - 2 ? p.a_o : p.a_ho, row, nA[q], mine);
            bstore4(q == Q - 2 ? p.b_o : p.b_ho, row, nB[q], mine && n > 0);
          }
          if (q < Q - 1) { ring_put(par, q, 0, nA[q]); ring_put(par, q, 1, nB[q]); }
        } else {
          if (q >= Q - 2 && n >= n0 && n < n1) {
            float *da = q == Q - 2 ? p.a_o : p.a_ho, *db = q == Q - 2 ? p.b_o : p.b_ho;
            const unsigned w = q == Q - 2 ? (wr & ALL) : (wr >> V);
            if (w == ALL) {
              store4(da, row, nA[q]);
              if (n > 0) store4(db, row, nB[q]);
            } else {
#pragma unroll
              for (int k = 0; k < V; k++) {
                if (w & (1u << k)) {
                  da[row + j + k] = nA[q][k];
                  if (n > 0) db[row + j + k] = nB[q][k];
                }
              }
            }
          }
          if (q < Q - 1) {
            if (need & (1u << q)) {
              float xa[V], xb[V];
              load4(p.pers_a[q], row, xa); load4(p.pers_b[q], row, xb);
#pragma unroll
              for (int k = 0; k < V; k++)
                if (!(ok & (1u << (q * V + k)))) { nA[q][k] = xa[k]; nB[q][k] = xb[k]; }
            }
            ring_put(par, q, 0, nA[q]); ring_put(par, q, 1, nB[q]);
          }
        }
        if constexpr (ROWS) {
          // a main lattice's rows 0 and 1 come from the chunk that starts at row 0 (a branch the whole wave
          // takes or not, in the first passes of a strip only)
          if ((q & 1) == 0 && n >= 0 && n <= 1 && n0 == 0) {
            if (interior) {
              float *rows = ra.rows[q >> 1];
#pragma unroll
              for (int k = 0; k < V; k++) {
                if ((q == Q - 2 ? (wr >> k) & 1u : (ok >> (q * V + k)) & 1u)) {
                  rows[row + j + k] = nA[q][k];
                  if (n == 1) rows[2 * P + j + k] = nB[q][k];
                }
              }
            }
          }
        }
      } else if (q < Q - 1 && n == p.N) {
        // row N is never written: the stage's "output" there is what its buffer holds
        load4(p.pers_a[q], (size_t)n * P, nA[q]); load4(p.pers_b[q], (size_t)n * P, nB[q]);
        if constexpr (ROWS) {
          // a single harmonic: row 1 is the row no sweep writes, and av() reads it all the same
          if ((q & 1) == 0 && n == 1 && n0 == 0 && interior) {
            float *rows = ra.rows[q >> 1];
#pragma unroll
            for (int k = 0; k < V; k++) {
              if ((ok >> (q * V + k)) & 1u) { rows[P + j + k] = nA[q][k]; rows[2 * P + j + k] = nB[q][k]; }
            }
          }
        }
      }
    }
    // ---- slide the histories that live in registers one row up ----
#pragma unroll
    for (int k = 0; k < V; k++) {
#pragma unroll
      for (int q = Q - 2; q >= 1; q--) a0h[q][k] = a0h[q - 1][k];
      a0h[0][k] = a0r[k];
      hA[1][k] = hA[0][k]; hB[1][k] = hB[0][k]; hA[0][k] = cur.upA[k]; hB[0][k] = cur.upB[k];
    }
  };

  // may this wave run the branch-free passes at all?  every lane: no never-written cells to fetch,
  // and the two last stages store all four of its columns or none
  const bool lane_plain = need == 0 && (wr == 0xffu || wr == 0u);
  const bool wave_plain = __builtin_amdgcn_ballot_w64(lane_plain) == ~0ull;
  // passes r in [s0, s1): from the strip's first pass (a pair of passes starts at an even row) to the last
  // one in which row r+1 is still a row the sweeps read (so that the request a pass ahead stays inside the
  // arrays).  In a strip's first passes the later stages run on rows before the strip's own, or before row 0
  // (their row index clamped for addressing): what they form is stored nowhere (a descriptor of length
  // zero) and read by no stage at a row that counts -- stage q+1 reads stage q's row n-1 only at its own
  // rows n >= 2 and stage q+2 reads row n as its centre, both of which stage q formed when it was already
  // at rows of its own.
  int s0 = a_lo + (a_lo & 1);
  const int s1 = n1 + Q - 2 < p.N - 1 ? n1 + Q - 2 : p.N - 1;

  RowIn x = {}, y = {};
  typedef std::integral_constant<bool, false> general_t;
  typedef std::integral_constant<bool, true> steady_t;
  typedef std::integral_constant<int, 0> even_t;
  typedef std::integral_constant<int, 1> odd_t;
  int r = a_lo;
  while (r <= r_end) {
    if (wave_plain && r >= s0 && r + 2 <= s1) {
      bload_row(r, x);
      // (waited for here, so that inside the loop the wait for a pass's rows is the same on the way
      // in as on the way round: "all but the nine accesses issued since", which leaves the stores out)
      asm volatile("" :: "v"(x.a0[0]), "v"(x.ac[0]), "v"(x.bc[0]), "v"(x.upA[0]), "v"(x.upB[0]));
      do {
        pass(steady_t(), even_t(), r, x, y);
        if constexpr (X == 1) pass(steady_t(), odd_t(), r + 1, x, y);
        else pass(steady_t(), odd_t(), r + 1, y, x);
        r += 2;
      } while (r + 2 <= s1);
    } else {
      pass(general_t(), even_t(), r, x, y);
      r++;
    }
  }
  if constexpr (X == 4) {
    if (ra.stamps && lane == 0) {
      unsigned long long *o = ra.stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
      o[0] = t_start; o[1] = __builtin_amdgcn_s_memrealtime();
      o[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_ID
      o[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // XCC_ID
    }
  }
}

// ---------------------------------------------------------------------------
// halo movers: one column of rows 0..nrows-1 of two arrays <-> contiguous buffer
// ---------------------------------------------------------------------------
// buf[(q*ncols + c)*nrows + r] <-> array q (0 = a, 1 = b), row r, column col + c
__global__ void halo_pack_kernel(const float *a, const float *b, int pitch, int nrows, int col, int ncols, float *buf)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * ncols * nrows) return;
  const int r = i % nrows, qc = i / nrows, c = qc % ncols;
  const float *src = qc < ncols ? a : b;
  buf[i] = src[(size_t)r * pitch + col + c];
}
__global__ void halo_unpack_kernel(float *a, float *b, int pitch, int nrows, int col, int ncols, const float *buf)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * ncols * nrows) return;
  const int r = i % nrows, qc = i / nrows, c = qc % ncols;
  float *dst = qc < ncols ? a : b;
  dst[(size_t)r * pitch + col + c] = buf[i];
}

// several (array pair, column run, buffer) segments in one launch: segment k covers
// elements start[k] .. start[k+1]-1 of the launch, laid out in its buffer like
// halo_pack_kernel's
struct HaloSegs {
  float *a[4], *b[4], *buf[4];
  int col[4], ncols[4], start[5];
  int nsegs, pitch, nrows;
};
template <bool PACK>
__global__ void halo_segs_kernel(const HaloSegs h)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h.start[h.nsegs]) return;
  int k = 0;
  while (k + 1 < h.nsegs && i >= h.start[k + 1]) k++;
  const int e = i - h.start[k];
  const int r = e % h.nrows, qc = e / h.nrows, c = qc % h.ncols[k];
  float *cell = (qc < h.ncols[k] ? h.a[k] : h.b[k]) + (size_t)r * h.pitch + h.col[k] + c;
  if (PACK) h.buf[k][e] = *cell; else *cell = h.buf[k][e];
}

// ---------------------------------------------------------------------------
// reductions (reference: src/boltzmann_c_solver.c:413-437)
// ---------------------------------------------------------------------------
struct AvArgs {
  const float *a, *b;
  int pitch, m_offset, j_first, j_last;
  float dPhi, PhiYmin;
};

// term q of column j: 0: b[1][j]*dPhi   1: a[0][j]*phi_y*dPhi   2: a[1][j]*dPhi
__device__ __forceinline__ float av_term(const AvArgs &p, int q, int j)
{
  if (q == 0) return p.b[(size_t)p.pitch + j] * p.dPhi;
  if (q == 2) return p.a[(size_t)p.pitch + j] * p.dPhi;
  const float phi = p.PhiYmin + p.dPhi * (float)(p.m_offset + j - 1);
  return p.a[j] * phi * p.dPhi;
}

// up to eight lattices' sums in one launch: workgroup 3*set + q forms sum q of lattice `set`
constexpr int kAvMaxSets = 8;
struct AvSets {
  AvArgs geom;                       // a, b unused
  const float *a[kAvMaxSets], *b[kAvMaxSets];
  float *tot[kAvMaxSets];
  const float *init[kAvMaxSets];     // where each lattice's three running sums start; null: at zero
};
__device__ __forceinline__ AvArgs av_set_args(const AvSets &s, int set)
{
  AvArgs p = s.geom;
#pragma unroll
  for (int k = 0; k < kAvMaxSets; k++) if (k == set) { p.a = s.a[k]; p.b = s.b[k]; }
  return p;
}
__device__ __forceinline__ float *av_set_tot(const AvSets &s, int set)
{
  float *t = s.tot[0];
#pragma unroll
  for (int k = 1; k < kAvMaxSets; k++) if (k == set) t = s.tot[k];
  return t;
}

__device__ __forceinline__ float av_set_init(const AvSets &s, int set, int q)
{
  const float *t = s.init[0];
#pragma unroll
  for (int k = 1; k < kAvMaxSets; k++) if (k == set) t = s.init[k];
  return t ? t[q] : 0.0f;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// tree mode, stage 1: per-block partial sums in double -> ws[block*3 + q]
__global__ void __launch_bounds__(256) av_partial_kernel(const AvArgs p, double *ws)
{
  __shared__ double red[3][4];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int j = p.j_first + blockIdx.x * blockDim.x + threadIdx.x; j <= p.j_last;
       j += gridDim.x * blockDim.x) {
    s0 += (double)av_term(p, 0, j);
    s1 += (double)av_term(p, 1, j);
    s2 += (double)av_term(p, 2, j);
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    ws[blockIdx.x * 3 + q] = ((red[q][0] + red[q][1]) + red[q][2]) + red[q][3];
  }
}

// tree mode, stage 2: fixed-order sum of the block partials -> sums[0..2]
__global__ void __launch_bounds__(64) av_combine_kernel(const double *ws, int nblocks, double *sums)
{
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblocks; i += 64) {
    s[0] += ws[i * 3 + 0]; s[1] += ws[i * 3 + 1]; s[2] += ws[i * 3 + 2];
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const double t = wave_sum(s[q]);
    if (threadIdx.x == 0) sums[q] = t;
  }
}

// the running-average update of av_data[0..5]; cos_wt / sin_wt are
// cos((double)(omega*t)), sin(...) evaluated by the host's libm so that the
// values are the ones the CPU solver uses
__device__ __forceinline__ void av_apply(float *av_data, float v_dr, float v_y, float mmx,
                                         double cos_wt, double sin_wt, float dt)
{
  const int av_count = (int)(av_data[0] + 1.0f);
  av_data[1] += (v_dr - av_data[1]) / (float)av_count;
  av_data[2] += (v_y  - av_data[2]) / (float)av_count;
  av_data[3] += (mmx  - av_data[3]) / (float)av_count;
  av_data[4] = (float)((double)av_data[4] + cos_wt * (double)v_dr * (double)dt);
  av_data[5] = (float)((double)av_data[5] + sin_wt * (double)v_dr * (double)dt);
  av_data[0] += 1.0f;
}

__global__ void av_update_kernel(const double *sums, float *av_data, double cos_wt, double sin_wt, float dt)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    av_apply(av_data, (float)sums[0], (float)sums[1], (float)sums[2], cos_wt, sin_wt, dt);
}

// exact mode: workgroup q (one wave; q = 0,1,2) forms sum q by adding the terms
// one at a time in float in column order, which is the reference's order of
// operations.  The 64 lanes form 64 x T terms in parallel and park them in LDS
// in column order (fetching the next 64 x T meanwhile); lane 0 then adds them
// one after the other, four per 16-byte LDS read.  The chain of dependent
// additions is the cost (about 8 cycles per column on gfx950), which is why the
// three sums run as three workgroups: waves of one workgroup can end up on the
// same SIMD and then take turns.  A one-lane kernel applies the update.
__global__ void __launch_bounds__(64) av_exact_kernel(const AvSets sets)
{
  constexpr int T = 16, B = 64 * T;                  // columns per group
  __shared__ __attribute__((aligned(16))) float park[B];
  const int set = blockIdx.x / 3;
  const int q   = blockIdx.x % 3;
  const AvArgs p = av_set_args(sets, set);
  float *tot = av_set_tot(sets, set);
  const int lane = threadIdx.x;
  float acc = av_set_init(sets, set, q);
  float term[T];
  int base = p.j_first;
#pragma unroll
  for (int t = 0; t < T; t++) {
    const int j = base + t * 64 + lane;
    term[t] = j <= p.j_last ? av_term(p, q, j) : 0.0f;
  }
  while (base <= p.j_last) {
    __syncthreads();                                 // lane 0 has finished with the last group
#pragma unroll
    for (int t = 0; t < T; t++) park[t * 64 + lane] = term[t];
    __syncthreads();                                 // all 64 lanes' terms are in place
    const int nb = base + B;
#pragma unroll
    for (int t = 0; t < T; t++) {                    // next group's loads fly during the chain
      const int j = nb + t * 64 + lane;
      term[t] = j <= p.j_last ? av_term(p, q, j) : 0.0f;
    }
    const int valid = p.j_last - base + 1;           // columns left, this group included
    if (lane == 0) {
      if (valid >= B) {
        const float4 *v4 = reinterpret_cast<const float4 *>(park);
#pragma unroll 8
        for (int i = 0; i < B / 4; i++) {
          const float4 v = v4[i];
          acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w;
        }
      } else {
        for (int i = 0; i < valid; i++) acc = acc + park[i];
      }
    }
    base = nb;
  }
  if (lane == 0) tot[q] = acc;
}

// ---------------------------------------------------------------------------
// A float sum in index order, evaluated in parallel, with the bits of the
// sequential loop `for (i) acc = acc + x[i]`.
//
// While the running sum stays inside one binade, |acc| in [2^e, 2^(e+1)), every
// addition rounds to the same grid of spacing u = 2^(e-23): with acc = A*u (A an
// integer) and h = x/u, fl(acc + x) = (A + rint(h))*u as long as h is not exactly
// halfway between two integers and the exact sum stays inside the binade.  Integer
// addition is associative, so a workgroup takes a chunk of terms at once: a few
// consecutive terms per thread, r = rint(h), a prefix sum over the workgroup, and a
// check of every prefix against the binade.  The first term that is a tie, leaves the binade, is too large
// or not finite is added by a real float addition instead (any term can be; the
// test is conservative), after which the exponent is read again and the rest of the
// chunk is redone.  When such a pass gets only a few terms further, one thread adds
// the next few hundred terms one by one, so the worst case costs little more than
// the plain loop.
// ---------------------------------------------------------------------------
constexpr int kScanBurst = 256;                 // terms one thread adds after a pass that got nowhere

// W waves per workgroup, T consecutive terms per thread: a chunk is 64*W*T terms
template <int W, int T>
struct ScanShared {
  int   wave_total[W];
  int   first_bad[W];
  int   prev_sum[W];
  int   last_sum;
  float acc_out;
  __attribute__((aligned(16))) float park[64 * W * T];
};

__device__ __forceinline__ float grid_value(int S, int biased_exp)
{
  // S*2^(e-23) for 2^23 <= |S| < 2^24: sign, the accumulator's exponent field, S's low 23 bits
  const unsigned sign = S < 0 ? 0x80000000u : 0u;
  const unsigned mag  = (unsigned)(S < 0 ? -S : S);
  return __uint_as_float(sign | ((unsigned)biased_exp << 23) | (mag & 0x7fffffu));
}

// inclusive prefix sum over the 64 lanes of a wave: shifts inside rows of 16 lanes, then
// the last lane of a row broadcast to the rows after it
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
  return v;
}

template <int W, int T, class F>
__device__ float seq_sum_block(F term, int n, ScanShared<W, T> &sh, int *stats = nullptr, float start = 0.0f)
{
  constexpr int kScanChunk = 64 * W * T;
  // |rint(h)| < 2^18 for every term of a chunk keeps all prefix sums inside an int
  static_assert((long long)kScanChunk << 18 <= 1ll << 30, "prefix sums must fit an int");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int e0 = tid * T;                                  // my first term's index in the chunk
  int n_passes = 0, n_bursts = 0, burst = kScanBurst;
  float acc = start;
  float x[T], x_next[T];
#pragma unroll
  for (int t = 0; t < T; t++) x_next[t] = e0 + t < n ? term(e0 + t) : 0.0f;
  for (int base = 0; base < n; base += kScanChunk) {
    const int count = n - base < kScanChunk ? n - base : kScanChunk;
#pragma unroll
    for (int t = 0; t < T; t++) {
      x[t] = x_next[t];
      sh.park[e0 + t] = x[t];
      const int jn = base + kScanChunk + e0 + t;           // the next chunk's loads fly during the scans
      x_next[t] = jn < n ? term(jn) : 0.0f;
    }
    int pos = 0;
    while (pos < count) {
      n_passes++;
      const unsigned ab = __float_as_uint(acc);
      const int  be = (int)((ab >> 23) & 0xffu);
      const bool normal = be != 0 && be != 0xff;            // else zero, denormal, inf or nan: no grid
      // which of my terms are still to be added: bits lo..hi-1
      const int lo = pos - e0 < 0 ? 0 : (pos - e0 > T ? T : pos - e0);
      const int hi = count - e0 < 0 ? 0 : (count - e0 > T ? T : count - e0);
      const unsigned mine = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
      unsigned badmask = 0;
      int ps[T];                                            // inclusive prefix of rint(h) over my terms
      int A0 = 0;
      if (normal) {
        const int m = (int)((ab & 0x7fffffu) | 0x800000u);
        A0 = (ab >> 31) ? -m : m;
        // h = x / 2^(e-23) with e = be - 127: a power-of-two scaling, exact in float unless
        // it overflows (caught as "too large") or underflows (then rint(h) = 0 either way)
        const int up = 150 - be;
        int run = 0;
#pragma unroll
        for (int t = 0; t < T; t++) {
          const float h  = __builtin_ldexpf(x[t], up);
          const float rr = __builtin_rintf(h);
          const bool big = !(__builtin_fabsf(h) < 262144.0f);          // also nan, inf: goes the slow way
          const bool tie = __builtin_fabsf(h - rr) == 0.5f;            // rounding would depend on the sum's parity
          badmask |= (big || tie) ? 1u << t : 0u;
          run += (((mine >> t) & 1u) && !big) ? (int)rr : 0;
          ps[t] = run;
        }
      } else {
        badmask = ~0u;
#pragma unroll
        for (int t = 0; t < T; t++) ps[t] = 0;
      }
      const int s = wave_inclusive_scan(ps[T - 1]);
      if (lane == 63) sh.wave_total[w] = s;
      __syncthreads();
      int before = A0 + s - ps[T - 1];
#pragma unroll
      for (int k = 0; k < W; k++) {
        const int v = sh.wave_total[k];
        before += k < w ? v : 0;
      }
      // the sum after each of my terms, in grid units; inside the binade means 2^23 < |S| < 2^24
      if (normal) {
#pragma unroll
        for (int t = 0; t < T; t++) {
          const int S = before + ps[t];
          const unsigned mag = (unsigned)(S < 0 ? -S : S);
          badmask |= (mag - 0x800001u >= 0x7fffffu) ? 1u << t : 0u;
        }
      }
      badmask &= mine;
      const int t_bad = badmask ? __builtin_ctz(badmask) : T;
      const unsigned long long vote = __ballot(badmask != 0);
      const int first = vote ? (int)__ffsll((long long)vote) - 1 : 64;
      // one lane per wave reports: the first lane with a bad term, or lane 0 that there is none
      if (lane == (vote ? first : 0)) {
        int prev = before;                                  // the sum before my first bad term
#pragma unroll
        for (int t = 1; t < T; t++) prev = t == t_bad ? before + ps[t - 1] : prev;
        sh.first_bad[w] = vote ? e0 + t_bad : 0x7fffffff;
        sh.prev_sum[w]  = prev;
      }
      if (e0 <= count - 1 && count - 1 < e0 + T) {
        int last = 0;
#pragma unroll
        for (int t = 0; t < T; t++) last = e0 + t == count - 1 ? before + ps[t] : last;
        sh.last_sum = last;
      }
      __syncthreads();
      int jb = 0x7fffffff, pv = 0;
#pragma unroll
      for (int k = W - 1; k >= 0; k--) {
        const int f = sh.first_bad[k], q = sh.prev_sum[k];
        if (f != 0x7fffffff) { jb = f; pv = q; }
      }
      if (jb == 0x7fffffff) {
        acc = grid_value(sh.last_sum, be);
        pos = count;
      } else {
        const float prev = normal ? grid_value(pv, be) : acc;
        acc = prev + sh.park[jb];                           // the real thing for this one term
        const int advanced = jb + 1 - pos;
        pos = jb + 1;
        if (advanced >= kScanBurst / 2) burst = kScanBurst;
        else if (pos < count) {
          // the scan did not pay for itself (the sum is changing binade every few terms,
          // or ties and specials are frequent): one thread adds the next terms one by one,
          // twice as many after every scan in a row that got nowhere
          const int stop = pos + burst < count ? pos + burst : count;
          if (tid == 0) {
            float a = acc;
            int i = pos;
            for (; i < stop && (i & 3); i++) a = a + sh.park[i];
            const float4 *v4 = reinterpret_cast<const float4 *>(sh.park);
#pragma unroll 8
            for (; i + 4 <= stop; i += 4) {
              const float4 v = v4[i >> 2];
              a = a + v.x; a = a + v.y; a = a + v.z; a = a + v.w;
            }
            for (; i < stop; i++) a = a + sh.park[i];
            sh.acc_out = a;
          }
          __syncthreads();
          acc = sh.acc_out;
          pos = stop;
          burst = burst < kScanChunk ? burst * 2 : kScanChunk;
          n_bursts++;
        }
      }
      // no barrier at the end of a pass: what the next pass writes before its first barrier
      // (wave_total) was last read before this pass's second barrier; everything else is
      // written after the next pass's first barrier, and acc_out after the one above
    }
    __syncthreads();                                        // park[] is rewritten next
  }
  if (stats && tid == 0) { stats[0] = n_passes; stats[1] = n_bursts; }
  return acc;
}

// exact mode, parallel form
constexpr int kAvScanWaves = 8, kAvScanPerLane = 8;     // fastest of the shapes tried at 2^20 columns
__global__ void __launch_bounds__(64 * kAvScanWaves) av_exact_scan_kernel(const AvSets sets)
{
  __shared__ ScanShared<kAvScanWaves, kAvScanPerLane> sh;
  const int set = blockIdx.x / 3, q = blockIdx.x % 3;
  const AvArgs p = av_set_args(sets, set);
  const float acc = seq_sum_block<kAvScanWaves, kAvScanPerLane>(
      [&](int i) { return av_term(p, q, p.j_first + i); }, p.j_last - p.j_first + 1, sh, nullptr,
      av_set_init(sets, set, q));
  if (threadIdx.x == 0) av_set_tot(sets, set)[q] = acc;
}

template <int W, int T>
__global__ void __launch_bounds__(64 * W) seq_sum_kernel(const float *x, int n, float *out, int *stats)
{
  __shared__ ScanShared<W, T> sh;
  const float acc = seq_sum_block<W, T>([&](int i) { return x[i]; }, n, sh, stats);
  if (threadIdx.x == 0) out[0] = acc;
}

__global__ void av_update_exact_kernel(const float *tot, float *av_data, double cos_wt, double sin_wt, float dt)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) av_apply(av_data, tot[0], tot[1], tot[2], cos_wt, sin_wt, dt);
}

// display=8 frame: out[p][c] = max(0, sum_n a[n][m] cos(n phi_x[p]) + b[n][m] sin(n phi_x[p])),
// m = col_first + c (reference: src/boltzmann_solver.c:495-504).  The sum keeps the
// reference's order and precisions: a float running value, each term formed and
// added in double, rounded back to float after every harmonic.  cs[p][n] holds
// (cos, sin) of n*phi_x[p] as the host's libm returns them, so the device adds
// nothing of its own to the result.  A lane owns one column and P phi_x values;
// the table entries are the same for the whole workgroup (scalar loads).
// SINES = false leaves the b terms out: the equilibrium distribution's sum over a0
// (reference: src/boltzmann_solver.c:322).
template <int P, bool SINES>
__global__ void __launch_bounds__(256) frame_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                    int pitch, int nrows, int col_first, int ncols,
                                                    const double2 *__restrict__ cs, int nphi,
                                                    float *__restrict__ out)
{
  const int c  = blockIdx.x * 256 + threadIdx.x;
  const int p0 = blockIdx.y * P;
  if (c >= ncols) return;
  const float *pa = a + col_first + c;
  const float *pb = SINES ? b + col_first + c : nullptr;
  const double2 *row[P];
#pragma unroll
  for (int p = 0; p < P; p++) {
    const int pp = p0 + p < nphi ? p0 + p : nphi - 1;     // the last tile repeats a row; stores are guarded
    row[p] = cs + (size_t)pp * nrows;
  }
  float v[P];
#pragma unroll
  for (int p = 0; p < P; p++) v[p] = 0.0f;
  for (int n = 0; n < nrows; n++) {
    const double ad = (double)pa[(size_t)n * pitch];
    const double bd = SINES ? (double)pb[(size_t)n * pitch] : 0.0;
#pragma unroll
    for (int p = 0; p < P; p++) {
      const double2 k = row[p][n];
      if (SINES) v[p] = (float)((double)v[p] + (ad * k.x + bd * k.y));
      else       v[p] = (float)((double)v[p] + ad * k.x);
    }
  }
#pragma unroll
  for (int p = 0; p < P; p++)
    if (p0 + p < nphi) out[(size_t)(p0 + p) * ncols + c] = v[p] < 0 ? 0.0f : v[p];
}

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
template <int MINW, bool DIAG = false, bool ROWS = false, int H = 2, bool NTS = true, bool NTL = false, int A0F = 0, int OPT = 0>
int launch_fused2(Fused2Args &args, int block, int n_chunk, int tail_chunk)
{
  int lo = args.jf[0], hi = args.jl[0];
  for (int q = 1; q < 4; q++) { lo = args.jf[q] < lo ? args.jf[q] : lo; hi = args.jl[q] > hi ? args.jl[q] : hi; }
  args.jj0 = H == 1 ? lo / 4 * 4 : lo / 32 * 32;         // wide halos: strips start on line boundaries
  const int per_wave = (64 - 2 * H) * 4;
  const int waves = (hi + 1 - args.jj0 + per_wave - 1) / per_wave;
  const int wpb = block / 64;
  const int groups = (waves + wpb - 1) / wpb;
  auto chunks_of = [&](int len) { return (args.N + len - 1) / len; };
  // (tuning variant 39: as if the occupancy query had failed -- every strip is cut into chunks)
  const int slots = n_chunk > 0 || g_variant == 39 ? 0 : resident_workgroups(fused2_step_kernel<MINW, DIAG, ROWS, H, NTS, NTL, A0F, OPT>, block);
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
                           int n_chunk, int slots, bool edge_first = false, int fewer_rounds = 0, int rest_chunk = 0)
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
  const bool edges = edge_first && slots > 0 && groups >= slots + 2;
  const int inner = edges ? groups - 2 : groups;          // strip groups other than the two at the ends
  int whole = slots > 0 && inner >= slots ? inner / slots * slots : 0;
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
  else                s.p = make_pipe_args<S>(g, a0, in, pers_a, pers_b, out, cosines, 2, block, n_chunk, slots, true);
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
  // timing builds: 92 = the rows of a pass requested in that pass, 93 = no stores, 95 = time stamps per wave
  if (v == 92 && g->steps == 3) return launch_ring<3, 2, false, true, 1>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
  if (v == 93 && g->steps == 3) return launch_ring<3, 2, false, true, 2>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
  if (v == 95 && g->steps == 3) return launch_ring<3, 2, false, true, 4>(g, a0, in, pers_a, pers_b, out, cosines, g_n_chunk);
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
  if (g->steps == 3 && block == 256 && (v == 0 || v == 2 || rows)) {
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
