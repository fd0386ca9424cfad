// How fast is a chain of dependent fp32 additions on one wave of an MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(const float *x, float *out, int n)
{
  float acc = 0.0f;
  const float4 *v4 = reinterpret_cast<const float4 *>(x);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll 8
    for (int i = 0; i < n / 4; i++) { float4 v = v4[i & 255]; acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w; }
    out[threadIdx.x >> 6] = acc;
  }
}
__global__ void chain_regs(float *out, int n, float a, float b)
{
  float acc = 0.0f;
  for (int i = 0; i < n / 8; i++) {
    acc = acc + a; acc = acc + b; acc = acc + a; acc = acc + b; acc = acc + a; acc = acc + b; acc = acc + a; acc = acc + b;
    asm volatile("" : "+v"(acc));
  }
  if (threadIdx.x == 0) out[0] = acc;
}
__global__ void spin(float *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f; }
int main()
{
  float *x, *out, *big; const int n = 1 << 20; const size_t nb = 1ull << 28;
  hipMalloc(&x, 4096); hipMemset(x, 0, 4096); hipMalloc(&out, 64); hipMalloc(&big, nb * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  for (int rep = 0; rep < 2; rep++) {
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, 0, big, nb);   // raise the clock
    hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, x, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("chain (loads from L1/L2): %.3f ms for 2^20 adds = %.2f ns/add\n", ms, ms * 1e6 / n);
    for (int w = 2; w <= 4; w++) {
      hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(1), dim3(64 * w), 0, 0, x, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); printf("  %d waves of one workgroup, a chain each: %.3f ms\n", w, ms);
    }
    hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(3), dim3(64), 0, 0, x, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("  3 workgroups of one wave, a chain each: %.3f ms\n", ms);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, 0, big, nb);
    hipEventRecord(e0); hipLaunchKernelGGL(chain_regs, dim3(1), dim3(64), 0, 0, out, n, 1e-9f, 2e-9f); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("chain (registers only):   %.3f ms for 2^20 adds = %.2f ns/add\n", ms, ms * 1e6 / n);
  }
  return 0;
}
