// Streaming micro-benchmark: what HBM rate does the sweep's access pattern allow?
// Each lane moves float4s; R read streams and W write streams; either a flat
// grid-stride sweep over every stream, or the sweep kernels' pattern (a wave
// owns 1 KiB of a row and walks down rows that are a pitch apart).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Ptrs { float4 *r[8]; float4 *w[8]; };

template <int R, int W>
__global__ void flat_kernel(Ptrs p, size_t n4)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; k++) { float4 v = p.r[k][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
    for (int k = 0; k < W; k++) p.w[k][i] = acc;
    if (W == 0 && acc.x == 12345.678f) p.r[0][0] = acc;
  }
}

// rows x cols4 (float4 columns); a lane owns one float4 column and walks `chunk` rows
template <int R, int W>
__global__ void rowwalk_kernel(Ptrs p, int rows, int cols4, size_t pitch4, int chunk)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols4) return;
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int r = r0; r < r1; r++) {
    const size_t i = (size_t)r * pitch4 + c;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; k++) { float4 v = p.r[k][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
    for (int k = 0; k < W; k++) p.w[k][i] = acc;
    if (W == 0 && acc.x == 12345.678f) p.r[0][0] = acc;
  }
}

// The double step's pattern: a wave owns INNER float4 columns (INNER = 62: the 64 lanes load
// 1 KiB that starts 16 bytes before the wave's columns, lanes 1..62 store -- pieces of 992
// bytes that do not line up with 128-byte lines; INNER = 60, 56: two, four halo lanes per side,
// stores of 960 bytes on 64-byte and of 896 bytes on 128-byte boundaries; INNER = 64: every lane
// loads and stores its own column, 1 KiB pieces on line boundaries) and walks `chunk` rows; dynamic LDS caps the
// workgroups per compute unit the way the real kernel's registers do.
// SM > 0: the strips do not overlap (every load on line boundaries) but SM lanes at each end of
// a wave do not store, which leaves the stores in pieces that do not line up -- separates what
// the loads' from what the stores' alignment costs.
// SM < 0: as SM = 0 with non-temporal stores (SM = -1), loads (-2) or both (-3); printed as -2, -4, -6 lanes "more not storing"
template <int R, int W, int INNER, int SM = 0>
__global__ void __launch_bounds__(256) strip_kernel(Ptrs p, int rows, int cols4, size_t pitch4, int chunk)
{
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int H = (64 - INNER) / 2;                   // halo lanes per side
  int c = wave * INNER + lane - H;
  if (wave * INNER >= cols4) return;
  constexpr int SMP = SM > 0 ? SM : 0;
  constexpr bool NTS = SM == -1 || SM == -3, NTL = SM == -2 || SM == -3;
  const bool store = lane >= H + SMP && lane < 64 - H - SMP && c >= 0 && c < cols4;
  c = c < 0 ? 0 : (c >= cols4 ? cols4 - 1 : c);
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int r = r0; r < r1; r++) {
    const size_t i = (size_t)r * pitch4 + c;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; k++) {
      float4 v;
      if constexpr (NTL) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 q = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&p.r[k][i]));
        v = make_float4(q.x, q.y, q.z, q.w);
      } else v = p.r[k][i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (store) {
#pragma unroll
      for (int k = 0; k < W; k++) {
        if constexpr (NTS) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 q = {acc.x, acc.y, acc.z, acc.w};
          __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(&p.w[k][i]));
        } else p.w[k][i] = acc;
      }
    }
    if (acc.x == 12345.678f) lds[threadIdx.x] = acc.y;
  }
}

// The same walk with the halo lanes at the two ends of a WORKGROUP (HW lanes per side of its
// 256) instead of at the ends of every wave, and NBAR workgroup barriers per row: what a
// kernel whose waves hand their edge columns to each other through LDS would see.
template <int R, int W, int HW, int NBAR>
__global__ void __launch_bounds__(256) wgstrip_kernel(Ptrs p, int rows, int cols4, size_t pitch4, int chunk)
{
  extern __shared__ float lds[];
  constexpr int INNER = 256 - 2 * HW;
  int c = blockIdx.x * INNER + (int)threadIdx.x - HW;
  const bool store = (int)threadIdx.x >= HW && (int)threadIdx.x < 256 - HW && c >= 0 && c < cols4;
  c = c < 0 ? 0 : (c >= cols4 ? cols4 - 1 : c);
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int r = r0; r < r1; r++) {
    const size_t i = (size_t)r * pitch4 + c;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; k++) { float4 v = p.r[k][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
    for (int b = 0; b < NBAR; b++) {
      if ((threadIdx.x & 63) == 0) lds[(threadIdx.x >> 6) + 4 * (b & 1)] = acc.x;
      __syncthreads();
      acc.y += lds[((threadIdx.x >> 6) ^ 1) + 4 * (b & 1)];
    }
    if (store) {
#pragma unroll
      for (int k = 0; k < W; k++) p.w[k][i] = acc;
    }
    if (acc.x == 12345.678f) lds[threadIdx.x] = acc.y;
  }
}

// 62-lane strips whose stores are put together into whole 128-byte lines: a lane whose line
// (8 lanes) lies inside its own wave's 62 stores directly; the three lines of a workgroup that
// straddle two waves are assembled in LDS and stored by eight lanes after a barrier.
template <int R, int W>
__global__ void __launch_bounds__(256) linestrip_kernel(Ptrs p, int rows, int cols4, size_t pitch4, int chunk)
{
  extern __shared__ float4 lds4[];                 // [2][W][256]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wl = w * 62 + lane - 1;                // lane's place among the workgroup's 248 columns
  int c = blockIdx.x * 248 + wl;
  const bool interior = lane >= 1 && lane < 63;
  const int line0 = (wl >> 3) << 3;
  const bool whole = interior && line0 >= w * 62 && line0 + 7 <= w * 62 + 61;
  const bool wg_ok = blockIdx.x * 248 + 248 <= cols4;
  const bool direct = wg_ok ? whole : (interior && c < cols4);
  const bool stage = wg_ok && interior && !whole;
  const bool gather = wg_ok && w < 3 && lane < 8;
  const int gl = (7 + 8 * w) * 8 + lane;           // the straddling line this wave puts together
  c = c < 0 ? 0 : (c >= cols4 ? cols4 - 1 : c);
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int r = r0; r < r1; r++) {
    const size_t i = (size_t)r * pitch4 + c;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; k++) { float4 v = p.r[k][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float4 *buf = lds4 + (r & 1) * W * 256;
    if (direct) {
#pragma unroll
      for (int k = 0; k < W; k++) p.w[k][i] = acc;
    }
    if (stage) {
#pragma unroll
      for (int k = 0; k < W; k++) buf[k * 256 + wl] = acc;
    }
    if (wg_ok) __syncthreads();
    if (gather) {
      const size_t gi = (size_t)r * pitch4 + blockIdx.x * 248 + gl;
#pragma unroll
      for (int k = 0; k < W; k++) p.w[k][gi] = buf[k * 256 + gl];
    }
    if (acc.x == 12345.678f) p.w[0][0] = acc;
  }
}

template <int R, int W>
void run_linestrips(Ptrs p, int rows, int cols4, size_t pitch4)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)(R + W) * rows * cols4 * 16.0;
  float ms;
  const size_t smem = (size_t)(160 * 1024 / 3) / 1024 * 1024 - 1024;
  CK(hipFuncSetAttribute((const void *)linestrip_kernel<R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  for (int chunk : {32, 56, 128}) {
    dim3 grid((cols4 + 247) / 248, (rows + chunk - 1) / chunk);
    hipLaunchKernelGGL((linestrip_kernel<R, W>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; it++)
      hipLaunchKernelGGL((linestrip_kernel<R, W>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("R=%d W=%d strips of 62 lanes, stores put together into whole lines, chunk=%4d   %8.1f GB/s\n", R, W, chunk,
           bytes / (ms / 10 * 1e-3) / 1e9);
  }
}

// 60-lane strips (the double step's present geometry) with the stores' cache-policy bits spelt
// out: HINT 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc1 nt, 7 sc0 nt
template <int HINT>
__device__ __forceinline__ void store_hint(float4 *ptr, float4 v)
{
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 q = {v.x, v.y, v.z, v.w};
  if constexpr (HINT == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(ptr), "v"(q) : "memory");
  if constexpr (HINT == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(ptr), "v"(q) : "memory");
}

template <int R, int W, int HINT>
__global__ void __launch_bounds__(256) hint_kernel(Ptrs p, int rows, int cols4, size_t pitch4, int chunk)
{
  extern __shared__ float lds[];
  constexpr int INNER = 60, H = 2;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int c = wave * INNER + lane - H;
  if (wave * INNER >= cols4) return;
  const bool store = lane >= H && lane < 64 - H && c >= 0 && c < cols4;
  c = c < 0 ? 0 : (c >= cols4 ? cols4 - 1 : c);
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int r = r0; r < r1; r++) {
    const size_t i = (size_t)r * pitch4 + c;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < R; k++) { float4 v = p.r[k][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (store) {
#pragma unroll
      for (int k = 0; k < W; k++) store_hint<HINT>(&p.w[k][i], acc);
    }
    if (acc.x == 12345.678f) lds[threadIdx.x] = acc.y;
  }
}

template <int R, int W, int HINT>
void run_hint(Ptrs p, int rows, int cols4, size_t pitch4)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)(R + W) * rows * cols4 * 16.0;
  float ms;
  const size_t smem = (size_t)(160 * 1024 / 3) / 1024 * 1024 - 1024;
  CK(hipFuncSetAttribute((const void *)hint_kernel<R, W, HINT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  for (int chunk : {56, 129}) {
    const int waves = (cols4 + 59) / 60;
    dim3 grid((waves + 3) / 4, (rows + chunk - 1) / chunk);
    hipLaunchKernelGGL((hint_kernel<R, W, HINT>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; it++)
      hipLaunchKernelGGL((hint_kernel<R, W, HINT>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("R=%d W=%d strips of 60 lanes, store policy %d, chunk=%4d   %8.1f GB/s\n", R, W, HINT, chunk,
           bytes / (ms / 10 * 1e-3) / 1e9);
  }
}

template <int R, int W, int HW, int NBAR>
void run_wgstrips(Ptrs p, int rows, int cols4, size_t pitch4)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)(R + W) * rows * cols4 * 16.0;
  float ms;
  constexpr int INNER = 256 - 2 * HW;
  const size_t smem = (size_t)(160 * 1024 / 3) / 1024 * 1024 - 1024;
  CK(hipFuncSetAttribute((const void *)wgstrip_kernel<R, W, HW, NBAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  for (int chunk : {32, 56, 128}) {
    dim3 grid((cols4 + INNER - 1) / INNER, (rows + chunk - 1) / chunk);
    hipLaunchKernelGGL((wgstrip_kernel<R, W, HW, NBAR>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; it++)
      hipLaunchKernelGGL((wgstrip_kernel<R, W, HW, NBAR>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("R=%d W=%d workgroup strips, %d halo lanes per side, %d barriers per row, chunk=%4d   %8.1f GB/s\n", R, W, HW, NBAR,
           chunk, bytes / (ms / 10 * 1e-3) / 1e9);
  }
}

template <int R, int W, int INNER, int SM = 0>
void run_strips(Ptrs p, int rows, int cols4, size_t pitch4)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)(R + W) * rows * cols4 * 16.0;
  float ms;
  for (int per_cu : {3}) {
    const size_t smem = (size_t)(160 * 1024 / per_cu) / 1024 * 1024 - 1024;     // workgroups per compute unit
    CK(hipFuncSetAttribute((const void *)strip_kernel<R, W, INNER, SM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int chunk : {32, 56, 128}) {
      const int waves = (cols4 + INNER - 1) / INNER;
      dim3 grid((waves + 3) / 4, (rows + chunk - 1) / chunk);
      hipLaunchKernelGGL((strip_kernel<R, W, INNER, SM>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
      CK(hipEventRecord(e0));
      for (int it = 0; it < 10; it++)
        hipLaunchKernelGGL((strip_kernel<R, W, INNER, SM>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("R=%d W=%d strips of %d lanes (%d more not storing), %d workgroups per CU, chunk=%4d   %8.1f GB/s\n", R, W, INNER, 2 * SM, per_cu, chunk,
             bytes / (ms / 10 * 1e-3) / 1e9);
    }
  }
}

template <int R, int W>
void run(Ptrs p, int rows, int cols4, size_t pitch4)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t n4 = (size_t)rows * pitch4;
  const double bytes = (double)(R + W) * rows * cols4 * 16.0;
  float ms;
  // flat
  for (int grid : {2048, 8192, 65536}) {
    hipLaunchKernelGGL((flat_kernel<R, W>), dim3(grid), dim3(256), 0, 0, p, n4);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL((flat_kernel<R, W>), dim3(grid), dim3(256), 0, 0, p, n4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("R=%d W=%d flat   grid=%6d            %8.1f GB/s\n", R, W, grid, (double)(R + W) * n4 * 16.0 / (ms / 10 * 1e-3) / 1e9);
  }
  for (int chunk : {8, 16, 32, 128}) {
    dim3 grid((cols4 + 255) / 256, (rows + chunk - 1) / chunk);
    hipLaunchKernelGGL((rowwalk_kernel<R, W>), grid, dim3(256), 0, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; it++) hipLaunchKernelGGL((rowwalk_kernel<R, W>), grid, dim3(256), 0, 0, p, rows, cols4, pitch4, chunk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("R=%d W=%d rowwalk chunk=%4d           %8.1f GB/s\n", R, W, chunk, bytes / (ms / 10 * 1e-3) / 1e9);
  }
}

int main(int argc, char **)
{
  const int rows = 129, cols4 = (1 << 20) / 4;
  const size_t pitch4 = cols4 + 8;
  const size_t n4 = (size_t)rows * pitch4;
  Ptrs p;
  for (int k = 0; k < 8; k++) { CK(hipMalloc(&p.r[k], n4 * 16)); CK(hipMemset(p.r[k], 0, n4 * 16)); }
  for (int k = 0; k < 8; k++) { CK(hipMalloc(&p.w[k], n4 * 16)); CK(hipMemset(p.w[k], 0, n4 * 16)); }
  CK(hipDeviceSynchronize());
  if (argc > 5) {
    run_hint<5, 4, 0>(p, rows, cols4, pitch4); run_hint<5, 4, 1>(p, rows, cols4, pitch4);
    run_hint<5, 4, 2>(p, rows, cols4, pitch4); run_hint<5, 4, 3>(p, rows, cols4, pitch4);
    run_hint<5, 4, 4>(p, rows, cols4, pitch4); run_hint<5, 4, 5>(p, rows, cols4, pitch4);
    run_hint<5, 4, 6>(p, rows, cols4, pitch4); run_hint<5, 4, 7>(p, rows, cols4, pitch4);
    run_hint<5, 4, 0>(p, rows, cols4, pitch4); run_hint<5, 4, 1>(p, rows, cols4, pitch4);
    return 0;
  }
  if (argc > 4) {
    run_strips<5, 4, 62>(p, rows, cols4, pitch4);
    run_strips<5, 4, 62, -1>(p, rows, cols4, pitch4);
    run_strips<5, 4, 62, -2>(p, rows, cols4, pitch4);
    run_strips<5, 4, 62, -3>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64, -1>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64, -3>(p, rows, cols4, pitch4);
    return 0;
  }
  if (argc > 3) {
    run_strips<5, 4, 62>(p, rows, cols4, pitch4);
    run_linestrips<5, 4>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64>(p, rows, cols4, pitch4);
    return 0;
  }
  if (argc > 2) {                       // loads and stores apart
    run_strips<5, 0, 62>(p, rows, cols4, pitch4);
    run_strips<5, 0, 64>(p, rows, cols4, pitch4);
    run_strips<1, 4, 62>(p, rows, cols4, pitch4);
    run_strips<1, 4, 64>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64, 1>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64, 4>(p, rows, cols4, pitch4);
    return 0;
  }
  if (argc > 1) {                       // strips only: the double step's pattern, lined up or not
    run_strips<5, 4, 62>(p, rows, cols4, pitch4);
    run_strips<5, 4, 60>(p, rows, cols4, pitch4);
    run_strips<5, 4, 56>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 1, 0>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 4, 0>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 8, 0>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 1, 1>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 1, 4>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 4, 4>(p, rows, cols4, pitch4);
    run_wgstrips<5, 4, 8, 4>(p, rows, cols4, pitch4);
    return 0;
  }
  run<1, 1>(p, rows, cols4, pitch4);
  run<1, 0>(p, rows, cols4, pitch4);
  run<5, 0>(p, rows, cols4, pitch4);
  run<5, 2>(p, rows, cols4, pitch4);
  run<5, 4>(p, rows, cols4, pitch4);
  run<2, 2>(p, rows, cols4, pitch4);
  return 0;
}
