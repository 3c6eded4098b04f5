// pb_selftest.hip -- on-device proofs that the fast exact forms of pb_device.hpp (pbSqrtFast, pbDiv2Fast,
// pbDistUnitFast) equal the compiler's IEEE sqrtf and division (pbSelfTest*: every float / every mantissa
// pair), and the shader-clock sampler of the roofline report.  Nothing here runs in a timestep.
#include "pb_engine.hpp"

namespace {

// ---- self-test of the fast exact math (pbSelfTest) ------------------------------------------
// every float bit pattern in pbSqrtFast's domain against hipcc's sqrtf
__global__ __launch_bounds__(256) void k_selftest_sqrt(unsigned long long *__restrict__ mismatches,
                                                       unsigned long long *__restrict__ checked) {
  const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 16u;
  uint32_t bad = 0, seen = 0;
  for (uint32_t k = 0; k < 16u; k++) {
    const uint32_t bits = base + k;
    const bool inDomain = bits == 0u || (bits >= 0x0F800000u && bits <= 0x7F800000u);
    if (!inDomain) continue;
    const float x = __uint_as_float(bits);
    seen++;
    if (__float_as_uint(pbSqrtFast(x)) != __float_as_uint(sqrtf(x))) bad++;
    // the one-transcendental pair geometry: its root for the same x (0 or >= 2^-96)
    // (finite x: the force kernel never sees an infinite d2 -- positions are clamped to the walls)
    if (bits != 0x7F800000u) {
      float dist, nx, ny;
      pbDistUnitFast(0.0f, 0.0f, x, dist, nx, ny);
      if (__float_as_uint(dist) != __float_as_uint(sqrtf(x))) bad++;
    }
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
  if (seen) atomicAdd(checked, (unsigned long long)seen);
}

// the static-friction hold (pb_device.hpp PbDevParams::holdV2 / holdF2): `sqrtf(x) < c` against `x < T(c)` for EVERY
// non-negative float bit pattern x (infinities and NaNs included), T computed on the host (pbSqrtThreshold)
__global__ __launch_bounds__(256) void k_selftest_hold(float c, float T, unsigned long long *__restrict__ mismatches,
                                                       unsigned long long *__restrict__ checked) {
  const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 16u;
  uint32_t bad = 0;
  for (uint32_t k = 0; k < 16u; k++) {
    const float x = __uint_as_float(base + k);
    if ((sqrtf(x) < c) != (x < T)) bad++;
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
  if (threadIdx.x == 0) atomicAdd(checked, 256ull * 16ull);
}

// sampled (numerator a, numerator b, denominator d) against hipcc's a/d, b/d, inside
// v_div_scale_f32's own "no scaling needed" region (which is pbDiv2Fast's domain)
PB_DEV bool pbDivNoScale(uint32_t nb, uint32_t db) {
  const int en = (int)((nb >> 23) & 255u), ed = (int)((db >> 23) & 255u);
  // denominator normal with a normal reciprocal (|d| <= 2^126); numerator >= 2^-100 so that the
  // residual n - d*q (24+24 bits below n's exponent) is exact -- at 2^-103, where v_div_scale_f32
  // itself stops scaling, one case in 3e9 rounds the other way; quotient neither near overflow
  // (exponent gap < 96) nor denormal
  if ((nb & 0x7FFFFFFFu) == 0u) return (nb == 0u) && ed >= 1 && ed <= 252;  // +0 numerator only
  return ed >= 1 && ed <= 252 && en >= 27 && en <= 254 && (en - ed) < 96 && (en - ed) > -125;
}

__global__ __launch_bounds__(256) void k_selftest_div(unsigned long long samplesPerThread, int focused,
                                                      unsigned long long *__restrict__ mismatches,
                                                      unsigned long long *__restrict__ checked) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  unsigned long long bad = 0, seen = 0;
  for (unsigned long long k = 0; k < samplesPerThread; k++) {
    uint64_t h1 = pbMix64(tid * samplesPerThread + k + (focused ? 0x1234567ull : 0ull));
    uint64_t h2 = pbMix64(h1 ^ 0x9E3779B97F4A7C15ull);
    uint32_t ab = (uint32_t)h1, bb = (uint32_t)(h1 >> 32), db = (uint32_t)h2;
    if (focused) {
      // the shapes the force kernel produces: |quotient| between 2^-60 and 2^8, d in [2^-50, 2^30]
      const uint32_t ed = 77u + (uint32_t)((h2 >> 32) % 81u);
      db = (db & 0x007FFFFFu) | (ed << 23);
      const uint32_t ea = ed + 8u - (uint32_t)((h2 >> 40) % 69u);
      const uint32_t eb = ed + 8u - (uint32_t)((h2 >> 48) % 69u);
      ab = (ab & 0x807FFFFFu) | (ea << 23);
      bb = (bb & 0x807FFFFFu) | (eb << 23);
      if (((h2 >> 56) & 15u) == 0u) ab = 0u;  // exact +0 numerators do occur (equal coordinates)
    }
    if (!pbDivNoScale(ab, db) || !pbDivNoScale(bb, db)) continue;
    const float a = __uint_as_float(ab), b = __uint_as_float(bb), d = __uint_as_float(db);
    float qa, qb;
    pbDiv2Fast(a, b, d, qa, qb);
    seen += 2;
    if (__float_as_uint(qa) != __float_as_uint(a / d)) bad++;
    if (__float_as_uint(qb) != __float_as_uint(b / d)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
  if (seen) atomicAdd(checked, seen);
}

// sampled pair geometry: d2 in [2^-88, 2^28] (what the force kernel can see), two coordinate differences no
// larger than the distance (either sign, or exactly +0): pbDistUnitFast against sqrtf and IEEE division.
// (The exhaustive version -- every mantissa pair, 2^47 divisions -- is tools/rsq_form_test.hip.)
__global__ __launch_bounds__(256) void k_selftest_geom(unsigned long long samplesPerThread,
                                                       unsigned long long *__restrict__ mismatches,
                                                       unsigned long long *__restrict__ checked) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  unsigned long long bad = 0, seen = 0;
  for (unsigned long long k = 0; k < samplesPerThread; k++) {
    const uint64_t h1 = pbMix64(tid * samplesPerThread + k + 0x5151ull), h2 = pbMix64(h1 ^ 0x9E3779B97F4A7C15ull);
    const uint32_t ed = 127u - 88u + (uint32_t)(h2 % 117u);
    const float d2 = __uint_as_float(((uint32_t)h1 & 0x007FFFFFu) | (ed << 23));
    const float ref = sqrtf(d2);
    const uint32_t eref = (__float_as_uint(ref) >> 23) & 255u;
    float a = __uint_as_float(((uint32_t)(h1 >> 32) & 0x807FFFFFu) | ((eref - (uint32_t)((h2 >> 8) % 45u)) << 23));
    float b = __uint_as_float(((uint32_t)(h2 >> 32) & 0x807FFFFFu) | ((eref - (uint32_t)((h2 >> 16) % 45u)) << 23));
    if (((h2 >> 24) & 15u) == 0u) a = 0.0f;
    if (!(fabsf(a) <= ref) || !(fabsf(b) <= ref)) continue;
    if ((a != 0.0f && fabsf(a) < 0x1p-100f) || fabsf(b) < 0x1p-100f) continue;
    float dist, nx, ny;
    pbDistUnitFast(a, b, d2, dist, nx, ny);
    seen += 2;
    if (__float_as_uint(dist) != __float_as_uint(ref)) bad++;
    if (__float_as_uint(nx) != __float_as_uint(a / ref)) bad++;
    if (__float_as_uint(ny) != __float_as_uint(b / ref)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
  if (seen) atomicAdd(checked, seen);
}

// EXHAUSTIVE pair geometry (pbSelfTestPairGeometry): pbDistUnitFast -- the function the kernels call, rare
// path included -- for d2 = every float of a slice of [1, 4) (the 2^24 mantissa x exponent-parity cases, 64
// slices of 2^18) against every numerator mantissa in [1, 2) (2^23): root vs sqrtf, quotient vs IEEE division.
// 8 threads per d2, 2^20 numerators each, two numerators per call (the x and the y component).
__global__ __launch_bounds__(256) void k_selftest_geom_exhaustive(uint32_t d0, unsigned long long *__restrict__ mismatches,
                                                                  unsigned long long *__restrict__ checked) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t di = d0 + (t >> 3), chunk = t & 7u;
  const float d2 = __uint_as_float(0x3F800000u + di);
  const float ref = sqrtf(d2);
  uint32_t bad = 0;
  const uint32_t a0 = 0x3F800000u + (chunk << 20);
  for (uint32_t i = 0; i < (1u << 20); i += 2u) {
    const float a = __uint_as_float(a0 + i), b = __uint_as_float(a0 + i + 1u);
    float dist, nx, ny;
    pbDistUnitFast(a, b, d2, dist, nx, ny);
    bad += __float_as_uint(dist) != __float_as_uint(ref);
    bad += __float_as_uint(nx) != __float_as_uint(a / ref);
    bad += __float_as_uint(ny) != __float_as_uint(b / ref);
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
  if (threadIdx.x == 0) atomicAdd(checked, 256ull << 20);
}

// EXHAUSTIVE division (pbSelfTestDivision): pbDiv2Fast for every denominator mantissa of a slice of [1, 2)
// (2^23 values, 64 slices of 2^17) against every numerator mantissa in [1, 2): 2^46 divisions in all.
__global__ __launch_bounds__(256) void k_selftest_div_exhaustive(uint32_t d0, unsigned long long *__restrict__ mismatches,
                                                                 unsigned long long *__restrict__ checked) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t di = d0 + (t >> 3), chunk = t & 7u;
  const float d = __uint_as_float(0x3F800000u + di);
  uint32_t bad = 0;
  const uint32_t a0 = 0x3F800000u + (chunk << 20);
  for (uint32_t i = 0; i < (1u << 20); i += 2u) {
    const float a = __uint_as_float(a0 + i), b = __uint_as_float(a0 + i + 1u);
    float qa, qb;
    pbDiv2Fast(a, b, d, qa, qb);
    bad += __float_as_uint(qa) != __float_as_uint(a / d);
    bad += __float_as_uint(qb) != __float_as_uint(b / d);
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
  if (threadIdx.x == 0) atomicAdd(checked, 256ull << 20);
}

// ---- shader-clock sampler (diagnostic) ---------------------------------------------------------
// ONE wave that sleeps for `ticks` of the 100 MHz real-time counter and reports how many shader
// cycles (s_memtime) went by meanwhile: launched on its own stream beside the force kernels it reads
// the clock the chip actually holds under that load (MI355X_MICROARCH.md, DVFS give-back item 6).
__global__ __launch_bounds__(64) void k_clock_sample(unsigned long long ticks, unsigned long long *__restrict__ out) {
  if (threadIdx.x != 0) return;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r = r0;
  while (r - r0 < ticks) {
    __builtin_amdgcn_s_sleep(64);
    r = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[0] = c1 - c0;
  out[1] = r - r0;
}

}  // namespace

extern "C" {

int pbSelfTest(unsigned long long div_samples, unsigned long long *sqrt_checked,
               unsigned long long *sqrt_mismatches, unsigned long long *div_checked,
               unsigned long long *div_mismatches) {
  unsigned long long *d = nullptr;
  PB_TRY(hipMalloc((void **)&d, 4 * sizeof(unsigned long long)));
  PB_TRY(hipMemset(d, 0, 4 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_selftest_sqrt, dim3(1u << 20), dim3(256), 0, 0, d + 1, d + 0);
  const unsigned threads = 4096u * 256u;
  const unsigned long long per = (div_samples / 2 + threads - 1) / threads;
  if (per) {
    hipLaunchKernelGGL(k_selftest_div, dim3(4096), dim3(256), 0, 0, per, 0, d + 3, d + 2);
    hipLaunchKernelGGL(k_selftest_div, dim3(4096), dim3(256), 0, 0, per, 1, d + 3, d + 2);
    hipLaunchKernelGGL(k_selftest_geom, dim3(4096), dim3(256), 0, 0, per, d + 3, d + 2);
  }
  PB_TRY(hipGetLastError());
  unsigned long long h[4];
  PB_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  PB_TRY(hipFree(d));
  if (sqrt_checked) *sqrt_checked = h[0];
  if (sqrt_mismatches) *sqrt_mismatches = h[1];
  if (div_checked) *div_checked = h[2];
  if (div_mismatches) *div_mismatches = h[3];
  return PB_OK;
}

int pbSelfTestHoldThreshold(float c, unsigned long long *checked, unsigned long long *mismatches) {
  unsigned long long *d = nullptr;
  PB_TRY(hipMalloc((void **)&d, 2 * sizeof(unsigned long long)));
  PB_TRY(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_selftest_hold, dim3(1u << 19), dim3(256), 0, 0, c, pbSqrtThreshold(c), d + 1, d + 0);
  PB_TRY(hipGetLastError());
  unsigned long long h[2];
  PB_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  PB_TRY(hipFree(d));
  if (checked) *checked = h[0];
  if (mismatches) *mismatches = h[1];
  return PB_OK;
}

int pbSelfTestPairGeometry(unsigned first_slice, unsigned slices, unsigned long long *checked,
                           unsigned long long *mismatches) {
  if (first_slice >= 64u || slices == 0u || first_slice + slices > 64u) return PB_ERR_ARG;
  unsigned long long *d = nullptr;
  PB_TRY(hipMalloc((void **)&d, 2 * sizeof(unsigned long long)));
  PB_TRY(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
  const uint32_t perSlice = (1u << 24) / 64u;  // d2 values per slice
  for (unsigned sl = first_slice; sl < first_slice + slices; sl++)
    hipLaunchKernelGGL(k_selftest_geom_exhaustive, dim3(perSlice * 8u / 256u), dim3(256), 0, 0, sl * perSlice, d + 1, d + 0);
  PB_TRY(hipGetLastError());
  unsigned long long h[2];
  PB_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  PB_TRY(hipFree(d));
  if (checked) *checked = h[0];
  if (mismatches) *mismatches = h[1];
  return PB_OK;
}

int pbSelfTestDivision(unsigned first_slice, unsigned slices, unsigned long long *checked,
                       unsigned long long *mismatches) {
  if (first_slice >= 64u || slices == 0u || first_slice + slices > 64u) return PB_ERR_ARG;
  unsigned long long *d = nullptr;
  PB_TRY(hipMalloc((void **)&d, 2 * sizeof(unsigned long long)));
  PB_TRY(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
  const uint32_t perSlice = (1u << 23) / 64u;  // denominators per slice
  for (unsigned sl = first_slice; sl < first_slice + slices; sl++)
    hipLaunchKernelGGL(k_selftest_div_exhaustive, dim3(perSlice * 8u / 256u), dim3(256), 0, 0, sl * perSlice, d + 1, d + 0);
  PB_TRY(hipGetLastError());
  unsigned long long h[2];
  PB_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  PB_TRY(hipFree(d));
  if (checked) *checked = h[0];
  if (mismatches) *mismatches = h[1];
  return PB_OK;
}

struct pbClockSample {
  hipStream_t stream = nullptr;
  unsigned long long *dev = nullptr;
};

int pbClockSampleBegin(pbClockSample **out, double seconds) {
  if (!out || !(seconds > 0.0) || seconds > 30.0) return PB_ERR_ARG;
  pbClockSample *h = new pbClockSample();
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void **)&h->dev, 2 * sizeof(unsigned long long)) != hipSuccess) {
    pbLastError() = "pbClockSampleBegin: stream/buffer creation failed";
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PB_ERR_HIP;
  }
  hipLaunchKernelGGL(k_clock_sample, dim3(1), dim3(64), 0, h->stream, (unsigned long long)(seconds * 1e8), h->dev);
  *out = h;
  return PB_OK;
}

int pbClockSampleEnd(pbClockSample *h, double *mhz, double *seconds_sampled) {
  if (!h) return PB_ERR_ARG;
  unsigned long long v[2] = {0, 0};
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(v, h->dev, sizeof v, hipMemcpyDeviceToHost);
  (void)hipFree(h->dev);
  (void)hipStreamDestroy(h->stream);
  delete h;
  if (e != hipSuccess || v[1] == 0) {
    pbLastError() = "pbClockSampleEnd: sampler kernel failed";
    return PB_ERR_HIP;
  }
  if (mhz) *mhz = (double)v[0] / (double)v[1] * 100.0;
  if (seconds_sampled) *seconds_sampled = (double)v[1] * 1e-8;
  return PB_OK;
}

}  // extern "C"
