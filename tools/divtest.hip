// dev tool: where does pbDiv2Fast differ from IEEE division?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "pb_device.hpp"
struct Ex { uint32_t a, d, fast, exact; };
__global__ void k(int focused, int which, unsigned long long per, Ex* ex, unsigned* nex, unsigned long long* hist) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  for (unsigned long long k = 0; k < per; k++) {
    uint64_t h1 = pbMix64(tid * per + k + (focused ? 0x1234567ull : 0ull));
    uint64_t h2 = pbMix64(h1 ^ 0x9E3779B97F4A7C15ull);
    uint32_t ab = which ? (uint32_t)(h1 >> 32) : (uint32_t)h1, db = (uint32_t)h2;
    if (focused) {
      const uint32_t ed = 77u + (uint32_t)((h2 >> 32) % 81u);
      db = (db & 0x007FFFFFu) | (ed << 23);
      const uint32_t ea = ed + 8u - (uint32_t)((h2 >> (which ? 48 : 40)) % 69u);
      ab = (ab & 0x807FFFFFu) | (ea << 23);
    }
    const int en = (ab >> 23) & 255, ed = (db >> 23) & 255;
    if ((ab & 0x7FFFFFFFu) == 0) continue;
    if (!(ed >= 1 && ed <= 252 && en >= 27 && en <= 254 && (en - ed) < 96 && (en - ed) > -125)) continue;
    float a = __uint_as_float(ab), d = __uint_as_float(db), qa, qb;
    pbDiv2Fast(a, a, d, qa, qb);
    float q = a / d;
    if (__float_as_uint(qa) != __float_as_uint(q)) {
      unsigned i = atomicAdd(nex, 1u);
      if (i < 64) ex[i] = Ex{ab, db, __float_as_uint(qa), __float_as_uint(q)};
      atomicAdd(&hist[(en - ed) + 128], 1ull);          // by exponent difference
      atomicAdd(&hist[256 + ed], 1ull);                  // by denominator exponent
      atomicAdd(&hist[512 + en], 1ull);                  // by numerator exponent
    }
  }
}
int main() {
  Ex* ex; unsigned* nex; unsigned long long* hist;
  hipMalloc(&ex, 64 * sizeof(Ex)); hipMalloc(&nex, 4); hipMalloc(&hist, 768 * 8);
  for (int focused = 0; focused < 4; focused++) {
    int which = focused >> 1;
    hipMemset(nex, 0, 4); hipMemset(hist, 0, 768 * 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, focused & 1, which, 1024ull, ex, nex, hist);
    Ex h[64]; unsigned n; unsigned long long hh[768];
    hipMemcpy(h, ex, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&n, nex, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost);
    printf("focused=%d mismatches=%u\n", focused, n);
    for (unsigned i = 0; i < (n < 8 ? n : 8); i++)
      printf("  a=%08x (%g) d=%08x (%g) fast=%08x exact=%08x\n", h[i].a, *(float*)&h[i].a, h[i].d, *(float*)&h[i].d, h[i].fast, h[i].exact);
    printf("  by en-ed:"); for (int i = 0; i < 256; i++) if (hh[i]) printf(" %d:%llu", i - 128, hh[i]); printf("\n");
    printf("  by ed:"); for (int i = 0; i < 256; i++) if (hh[256 + i]) printf(" %d:%llu", i, hh[256 + i]); printf("\n");
    printf("  by en:"); for (int i = 0; i < 256; i++) if (hh[512 + i]) printf(" %d:%llu", i, hh[512 + i]); printf("\n");
  }
  return 0;
}
