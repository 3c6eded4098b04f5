// one_newton_root_test.hip -- does the root of the pair geometry need TWO Newton steps on y = d2 * rsq(d2)?
// every float that is 0 or in [2^-96, FLT_MAX): one step / two steps (shipped) against sqrtf.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/one_newton_root_test.hip -o tools/one_newton_root_test
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned long long *cnt) {
  const unsigned base = (blockIdx.x * 256u + threadIdx.x) * 16u;
  unsigned bad1 = 0, bad2 = 0, bad1b = 0, seen = 0;
  for (unsigned i = 0; i < 16u; i++) {
    const unsigned bits = base + i;
    if (!(bits == 0u || (bits >= 0x0F800000u && bits < 0x7F800000u))) continue;
    const float x = __uint_as_float(bits), ref = sqrtf(x);
    const float s = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(x), 0.0f, 0x1.fffffep127f), h = 0.5f * s;
    float y = x * s;
    float e = __builtin_fmaf(-y, y, x);
    const float y1 = __builtin_fmaf(e, h, y);
    e = __builtin_fmaf(-y1, y1, x);
    const float y2 = __builtin_fmaf(e, h, y1);
    // variant: the first step with the residual against a half-corrected h (Goldschmidt-style): h' = h (1.5 - ...)? no:
    // one step but on the v_sqrt_f32 seed instead of x * rsq
    const float z = __builtin_amdgcn_sqrtf(x);
    const float z1 = __builtin_fmaf(__builtin_fmaf(-z, z, x), h, z);
    seen++;
    bad1 += __float_as_uint(y1) != __float_as_uint(ref);
    bad2 += __float_as_uint(y2) != __float_as_uint(ref);
    bad1b += __float_as_uint(z1) != __float_as_uint(ref);
  }
  if (seen) atomicAdd(cnt + 0, (unsigned long long)seen);
  if (bad1) atomicAdd(cnt + 1, (unsigned long long)bad1);
  if (bad2) atomicAdd(cnt + 2, (unsigned long long)bad2);
  if (bad1b) atomicAdd(cnt + 3, (unsigned long long)bad1b);
}
int main() {
  unsigned long long *d, h[4];
  (void)hipMalloc(&d, sizeof h);
  (void)hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k, dim3(1u << 20), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%llu floats: root from x*rsq(x) + ONE Newton step differs from sqrtf in %llu, + TWO steps (shipped) in %llu; "
         "v_sqrt_f32 + one step (h from rsq) in %llu\n", h[0], h[1], h[2], h[3]);
  return 0;
}
