// Exhaustive check of the "one v_rsq_f32" form of the pair geometry (pbDistUnitFast, pb_device.hpp):
//   s    = v_rsq_f32(d2), clamped to FLT_MAX           (1/sqrt(d2), 1 ulp)
//   dist = two Newton steps on y = d2*s with h = s/2   -> must equal sqrtf(d2) for EVERY float of the domain
//   r    = one Newton step on s against dist           (reciprocal of dist without a v_rcp_f32)
//   q    = a*r, two residual corrections               -> must equal a/dist for EVERY (d2, a)
// Part 1: every float in [2^-96, FLT_MAX).  Part 2: every d2 in [1, 4) (all 2^24 mantissa x exponent-parity
// cases) against every numerator mantissa a in [1, 2) (2^23): 2^47 divisions, compared with the compiler's
// IEEE a/dist.  Scaling d2 by 4^k and a by 2^m scales every intermediate exactly (no denormals inside the
// kernel's domain), so the mantissa cases are all there is; part 3 samples
// random exponents all the same.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rsq_form_test.hip -o /tmp/rsq_form_test && /tmp/rsq_form_test [slices]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define DEV __device__ __forceinline__

DEV float rsqClamped(float d2) {
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(d2), 0.0f, 0x1.fffffep127f);
}
DEV float sqrtRsq(float d2, float s) {
  float y = d2 * s;
  const float h = 0.5f * s;
  float e = __builtin_fmaf(-y, y, d2);
  y = __builtin_fmaf(e, h, y);
  e = __builtin_fmaf(-y, y, d2);
  return __builtin_fmaf(e, h, y);
}
DEV float sqrtRsq3(float d2, float s) {
  float y = d2 * s;
  const float h = 0.5f * s;
  for (int i = 0; i < 3; i++) y = __builtin_fmaf(__builtin_fmaf(-y, y, d2), h, y);
  return y;
}
DEV float quot(float a, float d, float r) {
  float q = a * r;
  float t = __builtin_fmaf(-d, q, a);
  q = __builtin_fmaf(t, r, q);
  t = __builtin_fmaf(-d, q, a);
  return __builtin_fmaf(t, r, q);
}

__global__ __launch_bounds__(256) void k_sqrt(unsigned long long *cnt) {
  const unsigned base = (blockIdx.x * 256u + threadIdx.x) * 16u;
  unsigned bad2 = 0, bad3 = 0, seen = 0;
  for (unsigned i = 0; i < 16u; i++) {
    const unsigned bits = base + i;
    if (!(bits == 0u || (bits >= 0x0F800000u && bits < 0x7F800000u))) continue;
    const float x = __uint_as_float(bits);
    const float ref = sqrtf(x);
    const float s = rsqClamped(x);
    seen++;
    bad2 += __float_as_uint(sqrtRsq(x, s)) != __float_as_uint(ref);
    bad3 += __float_as_uint(sqrtRsq3(x, s)) != __float_as_uint(ref);
  }
  if (seen) atomicAdd(cnt + 0, (unsigned long long)seen);
  if (bad2) atomicAdd(cnt + 1, (unsigned long long)bad2);
  if (bad3) atomicAdd(cnt + 2, (unsigned long long)bad3);
}

// d2 index range [d0, d0 + nd) of the 2^24 values in [1, 4); 8 threads per d2, 2^20 numerators each
__global__ __launch_bounds__(256) void k_div(unsigned d0, unsigned long long *cnt) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned di = d0 + (t >> 3), chunk = t & 7u;
  const float d2 = __uint_as_float(0x3F800000u + di);  // [1, 4)
  const float dist = sqrtf(d2);
  const float s = rsqClamped(d2);
  const float r1 = __builtin_fmaf(__builtin_fmaf(-dist, s, 1.0f), s, s);            // one Newton step from s
  // second-order step instead: r = s (1 + e + e^2), e = 1 - dist s  (error e^3: r is the correctly rounded 1/dist)
  const float e1 = __builtin_fmaf(-dist, s, 1.0f);
  const float r2 = __builtin_fmaf(__builtin_fmaf(e1, e1, e1), s, s);
  const float c0 = __builtin_amdgcn_rcpf(dist);
  const float rc = __builtin_fmaf(__builtin_fmaf(-dist, c0, 1.0f), c0, c0);         // what ships today
  unsigned bad1 = 0, bad2 = 0, badc = 0, rdiff = (__float_as_uint(r1) != __float_as_uint(rc));
  const unsigned a0 = 0x3F800000u + (chunk << 20);
  for (unsigned i = 0; i < (1u << 20); i++) {
    const float a = __uint_as_float(a0 + i);
    const unsigned ref = __float_as_uint(a / dist);
    bad1 += __float_as_uint(quot(a, dist, r1)) != ref;
    bad2 += __float_as_uint(quot(a, dist, r2)) != ref;
    badc += __float_as_uint(quot(a, dist, rc)) != ref;
  }
  if (bad1) atomicAdd(cnt + 0, (unsigned long long)bad1);
  if (bad2) atomicAdd(cnt + 1, (unsigned long long)bad2);
  if (badc) atomicAdd(cnt + 2, (unsigned long long)badc);
  if (rdiff && chunk == 0) atomicAdd(cnt + 3, 1ull);
  if (chunk == 0 && __float_as_uint(r2) != __float_as_uint(rc)) atomicAdd(cnt + 4, 1ull);
  if (bad1 && chunk == 0) printf("  d2 = %a (dist %a): r1 %a shipped r %a second-order r %a\n", d2, dist, r1, rc, r2);
}

// Would pbDiv2Fast survive WITHOUT the Newton step on v_rcp_f32's result (2 instructions per pair less)?
// every denominator mantissa d in [1, 2) x every numerator mantissa a in [1, 2): 2^46 divisions.
__global__ __launch_bounds__(256) void k_div_raw(unsigned d0, unsigned long long *cnt) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned di = d0 + (t >> 3), chunk = t & 7u;
  const float d = __uint_as_float(0x3F800000u + di);
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
  unsigned badRaw = 0, badNewton = 0;
  const unsigned a0 = 0x3F800000u + (chunk << 20);
  for (unsigned i = 0; i < (1u << 20); i++) {
    const float a = __uint_as_float(a0 + i);
    const unsigned ref = __float_as_uint(a / d);
    badRaw += __float_as_uint(quot(a, d, r0)) != ref;
    badNewton += __float_as_uint(quot(a, d, r1)) != ref;
  }
  if (badRaw) atomicAdd(cnt + 0, (unsigned long long)badRaw);
  if (badNewton) atomicAdd(cnt + 1, (unsigned long long)badNewton);
}

// random exponents: d2 in [2^-88, 2^28], a with |a| <= dist (a coordinate difference), any sign
DEV uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_sampled(unsigned long long per, unsigned long long *cnt) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  unsigned long long bad = 0, badS = 0, seen = 0;
  for (unsigned long long k = 0; k < per; k++) {
    const uint64_t h1 = mix(tid * per + k), h2 = mix(h1);
    const unsigned ed = 127u - 88u + (unsigned)(h2 % 117u);  // exponent of d2: 2^-88 .. 2^28
    const float d2 = __uint_as_float(((unsigned)h1 & 0x007FFFFFu) | (ed << 23));
    const float dist = sqrtf(d2);
    const float s = rsqClamped(d2);
    badS += __float_as_uint(sqrtRsq(d2, s)) != __float_as_uint(dist);
    const float e1 = __builtin_fmaf(-dist, s, 1.0f);
    const float r1 = __builtin_fmaf(__builtin_fmaf(e1, e1, e1), s, s);  // the second-order step
    // numerator: dist scaled down by 2^-(0..44) with a random mantissa and sign, or exactly +0
    const unsigned edist = (__float_as_uint(dist) >> 23) & 255u;
    const unsigned ea = edist - (unsigned)((h2 >> 8) % 45u);
    float a = __uint_as_float(((unsigned)(h1 >> 32) & 0x807FFFFFu) | (ea << 23));
    if (((h2 >> 20) & 15u) == 0u) a = 0.0f;
    if (!(__builtin_fabsf(a) <= dist) || (a != 0.0f && __builtin_fabsf(a) < 0x1p-100f)) continue;
    seen++;
    bad += __float_as_uint(quot(a, dist, r1)) != __float_as_uint(a / dist);
  }
  if (seen) atomicAdd(cnt + 0, seen);
  if (bad) atomicAdd(cnt + 1, bad);
  if (badS) atomicAdd(cnt + 2, badS);
}

int main(int argc, char **argv) {
  const unsigned slices = argc > 1 ? (unsigned)atoi(argv[1]) : 64u;  // of 64: how much of part 2 to run
  unsigned long long *d, h[5];
  hipMalloc(&d, sizeof h);
  hipMemset(d, 0, sizeof h);
  if (argc > 2 && argv[2][0] == 'r') {   // `rsq_form_test 64 raw`: the pbDiv2Fast-without-Newton question only
    const unsigned per = (1u << 23) / 64u;
    for (unsigned sl = 0; sl < slices && sl < 64u; sl++)
      hipLaunchKernelGGL(k_div_raw, dim3(per * 8u / 256u), dim3(256), 0, 0, sl * per, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("pbDiv2Fast on all %.3g (denominator, numerator) mantissa pairs: mismatches against IEEE division with the raw "
           "v_rcp_f32 result %llu, with the Newton step (what ships) %llu\n", (double)slices * per * 8388608.0, h[0], h[1]);
    return 0;
  }
  hipLaunchKernelGGL(k_sqrt, dim3(1u << 20), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("part 1: %llu floats (0 and [2^-96, FLT_MAX)): sqrt from rsq + 2 Newton steps differs from sqrtf in %llu, "
         "+ 3 steps in %llu\n", h[0], h[1], h[2]);
  fflush(stdout);
  hipMemset(d, 0, sizeof h);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  const unsigned perSlice = (1u << 24) / 64u;  // d2 values per launch
  for (unsigned sl = 0; sl < slices && sl < 64u; sl++) {
    hipLaunchKernelGGL(k_div, dim3(perSlice * 8u / 256u), dim3(256), 0, 0, sl * perSlice, d);
    if ((sl & 7u) == 7u || sl + 1 == slices) {
      hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
      printf("part 2: %u/64 of the d2 range done: mismatches r1 (one Newton step) %llu  r2 (second-order step) %llu  "
             "shipped form %llu ; r1 != shipped r in %llu d2, r2 != shipped r in %llu d2\n",
             sl + 1, h[0], h[1], h[2], h[3], h[4]);
      fflush(stdout);
    }
  }
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("part 2: %.3g divisions per variant in %.1f s\n", (double)slices * perSlice * 8388608.0, ms * 1e-3);
  hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k_sampled, dim3(8192), dim3(256), 0, 0, 2000ull, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("part 3: %llu sampled (d2, a) at random exponents: quotient mismatches %llu, sqrt mismatches %llu\n", h[0], h[1], h[2]);
  return 0;
}
