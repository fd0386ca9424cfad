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
template <int R, int W, int INNER>
__global__ void __launch_bounds__(256) strip_kernel(Ptrs p, int rows, int cols4, size_t pitch4, int chunk)
{
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int H = (64 - INNER) / 2;                   // halo lanes per side
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
      for (int k = 0; k < W; k++) p.w[k][i] = acc;
    }
    if (acc.x == 12345.678f) lds[threadIdx.x] = acc.y;
  }
}

template <int R, int W, int INNER>
void run_strips(Ptrs p, int rows, int cols4, size_t pitch4)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)(R + W) * rows * cols4 * 16.0;
  float ms;
  for (int per_cu : {3}) {
    const size_t smem = (size_t)(160 * 1024 / per_cu) / 1024 * 1024 - 1024;     // workgroups per compute unit
    CK(hipFuncSetAttribute((const void *)strip_kernel<R, W, INNER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int chunk : {32, 56, 128}) {
      const int waves = (cols4 + INNER - 1) / INNER;
      dim3 grid((waves + 3) / 4, (rows + chunk - 1) / chunk);
      hipLaunchKernelGGL((strip_kernel<R, W, INNER>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
      CK(hipEventRecord(e0));
      for (int it = 0; it < 10; it++)
        hipLaunchKernelGGL((strip_kernel<R, W, INNER>), grid, dim3(256), smem, 0, p, rows, cols4, pitch4, chunk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("R=%d W=%d strips of %d lanes, %d workgroups per CU, chunk=%4d   %8.1f GB/s\n", R, W, INNER, per_cu, chunk,
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
  if (argc > 1) {                       // strips only: the double step's pattern, lined up or not
    run_strips<5, 4, 62>(p, rows, cols4, pitch4);
    run_strips<5, 4, 60>(p, rows, cols4, pitch4);
    run_strips<5, 4, 56>(p, rows, cols4, pitch4);
    run_strips<5, 4, 64>(p, rows, cols4, pitch4);
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
