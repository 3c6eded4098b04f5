// dev: is one Newton step on v_sqrt_f32 with the reciprocal from v_rcp_f32 correctly rounded for
// EVERY float in [2^-96, FLT_MAX]?  (candidate replacement of the 8-instruction +-1 ulp fix-up)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ __launch_bounds__(256) void k(unsigned long long *cnt) {
  const unsigned base = (blockIdx.x * 256u + threadIdx.x) * 16u;
  unsigned badA = 0, badB = 0, badC = 0, badD = 0, badE = 0, seen = 0;
  for (unsigned i = 0; i < 16u; i++) {
    const unsigned bits = base + i;
    if (!(bits >= 0x0F800000u && bits < 0x7F800000u)) continue;
    const float x = __uint_as_float(bits);
    const float ref = sqrtf(x);
    seen++;
    const float y = __builtin_amdgcn_sqrtf(x);
    const float r0 = __builtin_amdgcn_rcpf(y);
    const float e = __builtin_fmaf(-y, y, x);
    const float yA = __builtin_fmaf(e * 0.5f, r0, y);
    const float yB = __builtin_fmaf(e, 0.5f * r0, y);
    // variant C: reciprocal refined by one Newton step first
    const float er = __builtin_fmaf(-y, r0, 1.0f);
    const float r1 = __builtin_fmaf(er, r0, r0);
    const float yC = __builtin_fmaf(e * 0.5f, r1, y);
    // variant D: TWO Newton steps with h = r0 / 2 (5 instructions after v_sqrt/v_rcp instead of the 8 of the
    // +-1 ulp fix-up); variant E: the same with the refined reciprocal
    const float h0 = 0.5f * r0;
    const float d1 = __builtin_fmaf(e, h0, y);
    const float yD = __builtin_fmaf(__builtin_fmaf(-d1, d1, x), h0, d1);
    const float h1 = 0.5f * r1;
    const float e1 = __builtin_fmaf(e, h1, y);
    const float yE = __builtin_fmaf(__builtin_fmaf(-e1, e1, x), h1, e1);
    badD += __float_as_uint(yD) != __float_as_uint(ref);
    badE += __float_as_uint(yE) != __float_as_uint(ref);
    badA += __float_as_uint(yA) != __float_as_uint(ref);
    badB += __float_as_uint(yB) != __float_as_uint(ref);
    badC += __float_as_uint(yC) != __float_as_uint(ref);
  }
  if (seen) atomicAdd(cnt + 0, (unsigned long long)seen);
  if (badA) atomicAdd(cnt + 1, (unsigned long long)badA);
  if (badB) atomicAdd(cnt + 2, (unsigned long long)badB);
  if (badC) atomicAdd(cnt + 3, (unsigned long long)badC);
  if (badD) atomicAdd(cnt + 4, (unsigned long long)badD);
  if (badE) atomicAdd(cnt + 5, (unsigned long long)badE);
}
int main() {
  unsigned long long *d, h[6];
  hipMalloc(&d, sizeof h);
  hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k, dim3(1u << 20), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("checked %llu floats: mismatches A(e/2*r0) %llu  B(e*(r0/2)) %llu  C(refined r) %llu  D(two steps, r0/2) %llu  "
         "E(two steps, refined r/2) %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  return 0;
}
