// one_correction_test.hip -- would the fast exact quotients survive with ONE residual correction instead of two?
//   two (what ships, hipcc's own sequence):  q = a*r; t = fma(-d,q,a); q = fma(t,r,q); t = fma(-d,q,a); q = fma(t,r,q)
//   one:                                      q = a*r; t = fma(-d,q,a); q = fma(t,r,q)
// with r the reciprocal the kernels use (a Newton step on v_rsq_f32's or v_rcp_f32's result).  Each quotient would be
// 3 instructions instead of 5, four quotients per candidate pair.  Exhaustive over the mantissas, as
// tools/rsq_form_test.hip: (A) the unit vector, d = sqrtf(d2) for every d2 in [1, 4) x every numerator mantissa
// (2^47); (B) pbDiv2Fast's general denominator, every d in [1, 2) x every numerator mantissa (2^46).  For the
// mismatching (d, a) it also records whether the SECOND residual t2 = fma(-d, q1, a) has the sign that would give them
// away, and how many distinct denominators are involved (a per-denominator test could be hoisted out of the pair).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/one_correction_test.hip -o tools/one_correction_test
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define DEV __device__ __forceinline__

DEV float quot2(float a, float d, float r) {
  float q = a * r;
  float t = __builtin_fmaf(-d, q, a);
  q = __builtin_fmaf(t, r, q);
  t = __builtin_fmaf(-d, q, a);
  return __builtin_fmaf(t, r, q);
}
DEV float quot1(float a, float d, float r) {
  const float q = a * r;
  const float t = __builtin_fmaf(-d, q, a);
  return __builtin_fmaf(t, r, q);
}

// second-order Newton step r = c (1 + e + e^2), e = 1 - d c: the correctly rounded reciprocal (Markstein: with it ONE
// correction gives the correctly rounded quotient unless d's mantissa is all ones) at the price of one more fma
DEV float recip2(float d, float c) {
  const float e = __builtin_fmaf(-d, c, 1.0f);
  return __builtin_fmaf(__builtin_fmaf(e, e, e), c, c);
}

// cnt: [0] mismatches of quot1, [1] mismatches of quot2 (must be 0), [2] denominators with at least one quot1 mismatch,
//      [3] quot1 mismatches whose second residual is nonzero (always? then no cheaper test exists than computing it)
template <bool ROOT>
__global__ __launch_bounds__(256) void k_test(unsigned d0, unsigned long long *cnt) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned di = d0 + (t >> 3), chunk = t & 7u;
  float d, r, rr;
  if (ROOT) {
    const float d2 = __uint_as_float(0x3F800000u + di);  // [1, 4)
    d = sqrtf(d2);
    const float s = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(d2), 0.0f, 0x1.fffffep127f);
    const float er = __builtin_fmaf(-d, s, 1.0f);
    r = __builtin_fmaf(er, s, s);
    if (er == 0x1p-24f) {  // the all-ones roots: the shipped code takes the v_rcp_f32 form there
      const float c = __builtin_amdgcn_rcpf(d);
      r = __builtin_fmaf(__builtin_fmaf(-d, c, 1.0f), c, c);
    }
    rr = recip2(d, s);
  } else {
    d = __uint_as_float(0x3F800000u + di);  // [1, 2)
    const float c = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, c, 1.0f), c, c);
    rr = recip2(d, c);
  }
  unsigned bad1 = 0, bad2 = 0, badT = 0, badR = 0;
  const unsigned a0 = 0x3F800000u + (chunk << 20);
  for (unsigned i = 0; i < (1u << 20); i++) {
    const float a = __uint_as_float(a0 + i);
    const unsigned ref = __float_as_uint(a / d);
    const float q1 = quot1(a, d, r);
    if (__float_as_uint(q1) != ref) {
      bad1++;
      badT += __builtin_fmaf(-d, q1, a) != 0.0f;
    }
    bad2 += __float_as_uint(quot2(a, d, r)) != ref;
    badR += __float_as_uint(quot1(a, d, rr)) != ref;
  }
  if (badR) {
    atomicAdd(cnt + 4, (unsigned long long)badR);
    atomicAdd(cnt + 5, 1ull);
    if (chunk == 0 || true) {
      // which denominators? (few are expected: print the first handful)
      if (atomicAdd(cnt + 6, 1ull) < 12ull) printf("    one correction + second-order reciprocal fails for d = %a (%u numerators of this chunk)\n", d, badR);
    }
  }
  if (bad1) {
    atomicAdd(cnt + 0, (unsigned long long)bad1);
    atomicAdd(cnt + 3, (unsigned long long)badT);
    atomicAdd(cnt + 2, 1ull);  // (per chunk: an upper bound on the denominators involved, x <= 8)
  }
  if (bad2) atomicAdd(cnt + 1, (unsigned long long)bad2);
}

int main(int argc, char **argv) {
  const unsigned slices = argc > 1 ? (unsigned)atoi(argv[1]) : 64u;
  unsigned long long *d, h[7];
  (void)hipMalloc(&d, sizeof h);
  for (int part = 0; part < 2; part++) {
    (void)hipMemset(d, 0, sizeof h);
    const unsigned total = part == 0 ? (1u << 24) : (1u << 23), per = total / 64u;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (unsigned sl = 0; sl < slices && sl < 64u; sl++) {
      if (part == 0) hipLaunchKernelGGL(k_test<true>, dim3(per * 8u / 256u), dim3(256), 0, 0, sl * per, d);
      else hipLaunchKernelGGL(k_test<false>, dim3(per * 8u / 256u), dim3(256), 0, 0, sl * per, d);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%s: %.4g (denominator, numerator) mantissa pairs in %.1f s: ONE correction differs from IEEE division in %llu "
           "(in at most %llu denominator chunks; second residual nonzero in %llu of them); TWO corrections (shipped) in %llu; ONE correction with the second-order reciprocal in %llu (%llu chunks)\n",
           part == 0 ? "A unit vector, d = sqrtf(d2), d2 in [1,4)" : "B general denominator in [1,2)",
           (double)slices * per * 8388608.0, ms * 1e-3, h[0], h[2], h[3], h[1], h[4], h[5]);
    fflush(stdout);
  }
  return 0;
}
