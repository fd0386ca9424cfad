// What v_div_scale_f32 / v_div_fmas_f32 do on this GPU, by trying them: prints, for a fixed denominator
// and a ladder of numerators, the scaled numerator and denominator with their flags, and the
// factor v_div_fmas_f32 applies when its flag input is set.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void probe(const float *n, int cnt, float d, float *out)
{
  const int i = threadIdx.x;
  if (i >= cnt) return;
  bool fn, fd;
  const float ns = __builtin_amdgcn_div_scalef(n[i], d, true, &fn);    // numerator scaled
  const float ds = __builtin_amdgcn_div_scalef(n[i], d, false, &fd);   // denominator scaled
  out[i * 8 + 0] = ns; out[i * 8 + 1] = fn ? 1.f : 0.f;
  out[i * 8 + 2] = ds; out[i * 8 + 3] = fd ? 1.f : 0.f;
  out[i * 8 + 4] = __builtin_amdgcn_div_fmasf(0.0f, 0.0f, n[i], true);
  out[i * 8 + 5] = __builtin_amdgcn_div_fmasf(0.0f, 0.0f, n[i], false);
  out[i * 8 + 6] = n[i] / d;
  out[i * 8 + 7] = 0;
}
int main()
{
  const int cnt = 40;
  float hn[cnt], ho[cnt * 8], *dn, *dout;
  int k = 0;
  for (int e = -149; e <= 127 && k < cnt - 4; e += 8) hn[k++] = ldexpf(1.25f, e);
  hn[k++] = 0.0f; hn[k++] = -0.0f; hn[k++] = ldexpf(1.0f, -126); hn[k++] = ldexpf(1.0f, -103);
  hipMalloc(&dn, sizeof(hn)); hipMalloc(&dout, sizeof(ho));
  hipMemcpy(dn, hn, sizeof(hn), hipMemcpyHostToDevice);
  for (float d : {1.5f, 1.0f, 1024.0f}) {
    probe<<<1, 64>>>(dn, k, d, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    printf("d = %g\n", d);
    for (int i = 0; i < k; i++) {
      int en, es, ed, ef;
      frexpf(hn[i], &en); frexpf(ho[i * 8], &es); frexpf(ho[i * 8 + 2], &ed); frexpf(ho[i * 8 + 4], &ef);
      printf("  n=%-14g (2^%4d)  num_scaled 2^%+4d flag %g | den_scaled=%-12g (2^%+3d) flag %g | fmas(vcc=1) 2^%+4d  fmas(vcc=0)=%g  n/d=%g\n",
             hn[i], en - 1, es - en, ho[i * 8 + 1], ho[i * 8 + 2], ed - 1, ho[i * 8 + 3], ef - en, ho[i * 8 + 5], ho[i * 8 + 6]);
    }
  }
  return 0;
}
