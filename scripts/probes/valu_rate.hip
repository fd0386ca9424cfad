// Issue cost of the vector instructions the step kernels are made of, on one SIMD of gfx950, with 1, 2 and 4
// waves per SIMD: cycles (s_memtime) per instruction per SIMD.  Each test is a loop of 64 independent copies of
// one instruction.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int T>
__global__ void __launch_bounds__(256) k(unsigned long long *out, int iters, float seed)
{
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = 1.0001f, b1 = 0.9999f, c0 = 0.5f, c1 = 0.25f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b0, b1}, pc = {c0, c1};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if constexpr (T == 0) {        // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));)
    } else if constexpr (T == 1) { // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
    } else if constexpr (T == 2) { // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                        "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
    } else if constexpr (T == 3) { // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));)
    } else if constexpr (T == 4) { // v_mul_f32 / v_add_f32 alternating
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                        "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));)
    } else if constexpr (T == 5) { // v_mov_b32
      REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n"
                        "v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (T == 6) { // v_mov_b32 dpp wave_shr
      REP8(asm volatile("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %4, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %6, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (T == 7) { // v_rcp_f32
      REP8(asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_rcp_f32 %2, %6\n v_rcp_f32 %3, %7\n"
                        "v_rcp_f32 %4, %0\n v_rcp_f32 %5, %1\n v_rcp_f32 %6, %2\n v_rcp_f32 %7, %3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (T == 8) { // v_div_scale_f32 (writes vcc)
      REP8(asm volatile("v_div_scale_f32 %0, vcc, %4, %5, %4\n v_div_scale_f32 %1, vcc, %5, %6, %5\n v_div_scale_f32 %2, vcc, %6, %7, %6\n v_div_scale_f32 %3, vcc, %7, %4, %7\n"
                        "v_div_scale_f32 %0, vcc, %4, %5, %4\n v_div_scale_f32 %1, vcc, %5, %6, %5\n v_div_scale_f32 %2, vcc, %6, %7, %6\n v_div_scale_f32 %3, vcc, %7, %4, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");)
    } else if constexpr (T == 9) { // v_div_fmas_f32 (reads vcc)
      REP8(asm volatile("v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n"
                        "v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));)
    } else if constexpr (T == 10) { // v_div_fixup_f32
      REP8(asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n"
                        "v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));)
    } else if constexpr (T == 11) { // v_cmp_neq_f32 -> vcc
      REP8(asm volatile("v_cmp_neq_f32 vcc, %0, %1\n v_cmp_neq_f32 vcc, %1, %2\n v_cmp_neq_f32 vcc, %2, %3\n v_cmp_neq_f32 vcc, %3, %4\n"
                        "v_cmp_neq_f32 vcc, %4, %5\n v_cmp_neq_f32 vcc, %5, %6\n v_cmp_neq_f32 vcc, %6, %7\n v_cmp_neq_f32 vcc, %7, %0\n"
                        :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");)
    } else if constexpr (T == 12) { // dependent chain v_fma_f32
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b0), "v"(c0));)
    } else if constexpr (T == 13) { // dependent chain v_pk_fma_f32
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(p0) : "v"(pb), "v"(pc));)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x % 64 == 0) {
    out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0;
    out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = (unsigned long long)s;
  }
}

template <int T>
void run(const char *name, unsigned long long *d_out)
{
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps;      // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[4096 * 2];
    hipMemcpy(h, d_out, sizeof(unsigned long long) * blocks * 4 * 2, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < blocks * 4; i++) mean += (double)h[2 * i]; mean /= blocks * 4;
    const double per_wave = mean / (iters * 64.0);          // memtime ticks per instruction, one wave
    // s_memtime runs at a fixed 100 MHz on this part: convert with the wall time instead
    const double instr_per_simd = (double)iters * 64.0 * wps;
    const double ns_per_instr_simd = ms * 1e6 / instr_per_simd;
    printf("%-22s waves/SIMD %d: %7.3f ms  %.3f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz)  [memtime ticks per instr per wave %.4f]\n",
           name, wps, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4, per_wave);
  }
}

int main()
{
  unsigned long long *d; hipMalloc(&d, sizeof(unsigned long long) * 4096 * 2);
  run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d);
  run<4>("v_mul/v_add_f32", d); run<5>("v_mov_b32", d); run<6>("v_mov_b32 dpp", d); run<7>("v_rcp_f32", d);
  run<8>("v_div_scale_f32", d); run<9>("v_div_fmas_f32", d); run<10>("v_div_fixup_f32", d); run<11>("v_cmp_neq_f32", d);
  run<12>("v_fma_f32 chain", d); run<13>("v_pk_fma_f32 chain", d);
  return 0;
}
