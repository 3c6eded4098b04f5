// pb_xorwow.hpp -- cuRAND-compatible XORWOW phase-noise generator (opt-in: `pb_rng curand`).
//
// Reference: curand_setup_kernel / add_normal_noise_kernel (particlebot_kernel_impl.cuh:36-51) call
//   curand_init(params.seed, /*subsequence*/ i, /*offset*/ 0, &state[i])   once per bot i, and
//   val[i] += std * curand_normal(&state[i])                               at every phase update.
// cuRAND itself is a third-party dependency that is NOT in this image and not vendored by the
// reference (Makefile:2 uses whatever $(CUDA_PATH) holds; no version pinned), so this file restates
// its PUBLISHED algorithm (curand_kernel.h, curandStateXORWOW):
//   * generator: Marsaglia's xorwow -- five 32-bit xorshift words + a Weyl sequence d += 362437,
//     output v[4] + d;
//   * curand_init: the seed's two halves are salted (xor) and multiplied by fixed odd constants and
//     mixed into Marsaglia's default state; then the xorshift words are advanced by subsequence *
//     2^67 steps with a 160x160 GF(2) jump matrix (d is unchanged: 362437 * 2^67 = 0 mod 2^32);
//   * curand_normal: Box-Muller on two consecutive outputs, u = x*2^-32 + 2^-33, v = y*2^-32*2pi +
//     2^-33*2pi; returns s*sin(v) and keeps s*cos(v) for the next call.
// What can be checked HERE: rocRAND's xorwow_engine (/opt/rocm/include/rocrand/rocrand_xorwow.h) is
// the same generator and the same 2^67 jump with different salt constants; with rocRAND's constants
// this code reproduces rocRAND's host engine bit for bit (tests/test_xorwow.py).  What cannot: the
// four cuRAND salt constants below are quoted from cuRAND's header, and curand_normal's logf /
// __sincosf are CUDA device functions -- the transform here uses this project's own fp32 polynomial
// log/sin/cos (shared bit for bit with the oracle), i.e. the normals agree with a CUDA run only to
// float rounding.  Label: integer stream cuRAND-compatible by construction, UNVERIFIED against CUDA.
#pragma once

#include <stdint.h>
#include <string.h>

#include "particlebot_hip.h"

#if defined(__HIPCC__)
#define PB_XW_HD __host__ __device__ __forceinline__
#else
#define PB_XW_HD inline
#endif

#define PB_XW_ROWS 160                       // state bits of the xorshift part
#define PB_XW_MAT_WORDS (PB_XW_ROWS * 5)     // one jump matrix: row r = image of the state with only bit r set
#define PB_XW_TABLE_MATS 32                  // J^(2^k), k = 0..31, J = one subsequence (2^67 steps)
#define PB_XW_TABLE_WORDS (PB_XW_TABLE_MATS * PB_XW_MAT_WORDS)

struct PbXorwowSalt {
  uint32_t xor0, xor1, mul0, mul1;
};

PB_XW_HD PbXorwowSalt pbXorwowSalt(int kind) {
  // cuRAND (curand_kernel.h, _curand_init_scratch) / rocRAND (rocrand_xorwow.h:113-116)
  return kind == PB_RNG_XORWOW_ROCRAND ? PbXorwowSalt{0x2c7f967fu, 0xa03697cbu, 1228688033u, 2073658381u}
                                       : PbXorwowSalt{0xaad26b49u, 0xf7dcefddu, 1099087573u, 2591861531u};
}

// state of subsequence 0 for a 64-bit seed
PB_XW_HD void pbXorwowSeed(pbRngState &s, uint64_t seed, int kind) {
  const PbXorwowSalt k = pbXorwowSalt(kind);
  const uint32_t s0 = (uint32_t)seed ^ k.xor0, s1 = (uint32_t)(seed >> 32) ^ k.xor1;
  const uint32_t t0 = k.mul0 * s0, t1 = k.mul1 * s1;
  s.d = 6615241u + t1 + t0;
  s.v[0] = 123456789u + t0;
  s.v[1] = 362436069u ^ t0;
  s.v[2] = 521288629u + t1;
  s.v[3] = 88675123u ^ t1;
  s.v[4] = 5783321u + t0;
  s.boxmuller_flag = 0;
  s.kind = kind;
  s.boxmuller_extra = 0.0f;
  s.reserved[0] = s.reserved[1] = s.reserved[2] = 0.0f;
}

// one step of the xorshift words (linear over GF(2))
PB_XW_HD void pbXorwowShift(uint32_t v[5]) {
  const uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1];
  v[1] = v[2];
  v[2] = v[3];
  v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

PB_XW_HD uint32_t pbXorwowNext(pbRngState &s) {
  pbXorwowShift(s.v);
  s.d += 362437u;
  return s.v[4] + s.d;
}

// v <- v * M  (M: PB_XW_MAT_WORDS words)
PB_XW_HD void pbXorwowApply(uint32_t v[5], const uint32_t *M) {
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
  for (int w = 0; w < 5; w++) {
    uint32_t bits = v[w];
    for (int b = 0; b < 32; b++) {
      const uint32_t m = 0u - ((bits >> b) & 1u);  // all ones if the bit is set
      const uint32_t *row = M + (w * 32 + b) * 5;
      r0 ^= row[0] & m;
      r1 ^= row[1] & m;
      r2 ^= row[2] & m;
      r3 ^= row[3] & m;
      r4 ^= row[4] & m;
    }
  }
  v[0] = r0;
  v[1] = r1;
  v[2] = r2;
  v[3] = r3;
  v[4] = r4;
}

// skip `subsequence` subsequences of 2^67 numbers each (curand_init's second argument)
PB_XW_HD void pbXorwowSkipSubsequences(pbRngState &s, uint32_t subsequence, const uint32_t *table) {
  for (int k = 0; k < PB_XW_TABLE_MATS && subsequence; k++, subsequence >>= 1)
    if (subsequence & 1u) pbXorwowApply(s.v, table + (size_t)k * PB_XW_MAT_WORDS);
}

// ---- fp32 log / sincos with a fixed operation order (no FMA contraction: -ffp-contract=off), so that
// ---- the HIP kernels and the oracle's own restatement agree bit for bit
PB_XW_HD float pbXwLog(float u) {  // u a positive normal float
  uint32_t bits;
  memcpy(&bits, &u, 4);
  int e = (int)(bits >> 23) - 127;
  uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
  float m;
  memcpy(&m, &mb, 4);  // [1, 2)
  if (m > 1.41421356f) {
    m = m * 0.5f;
    e += 1;
  }
  const float t = (m - 1.0f) / (m + 1.0f);
  const float t2 = t * t;
  float p = 0.111111111f;
  p = p * t2 + 0.142857143f;
  p = p * t2 + 0.2f;
  p = p * t2 + 0.333333333f;
  p = p * t2 + 1.0f;
  return 2.0f * t * p + (float)e * 0.693147181f;
}

PB_XW_HD void pbXwSinCos(float v, float &sn, float &cs) {  // v in [0, 2*pi + ulp]
  int q = (int)(v * 0.636619772f);  // quadrant = floor(v / (pi/2))
  float a = v - (float)q * 1.57079633f;
  if (a < 0.0f) {  // the product rounded up past a quadrant boundary
    q -= 1;
    a = a + 1.57079633f;
  }
  const float a2 = a * a;
  float s = -2.50521084e-8f;
  s = s * a2 + 2.75573192e-6f;
  s = s * a2 - 1.98412698e-4f;
  s = s * a2 + 8.33333333e-3f;
  s = s * a2 - 1.66666667e-1f;
  s = s * a2 + 1.0f;
  s = s * a;
  float c = 2.08767570e-9f;
  c = c * a2 - 2.75573192e-7f;
  c = c * a2 + 2.48015873e-5f;
  c = c * a2 - 1.38888889e-3f;
  c = c * a2 + 4.16666667e-2f;
  c = c * a2 - 0.5f;
  c = c * a2 + 1.0f;
  switch (q & 3) {
    case 0: sn = s, cs = c; break;
    case 1: sn = c, cs = -s; break;
    case 2: sn = -s, cs = -c; break;
    default: sn = -c, cs = s; break;
  }
}

// curand_normal(curandStateXORWOW_t*): Box-Muller pair, second value cached in the state
PB_XW_HD float pbXorwowNormal(pbRngState &s) {
  if (s.boxmuller_flag) {
    s.boxmuller_flag = 0;
    return s.boxmuller_extra;
  }
  const uint32_t x = pbXorwowNext(s);
  const uint32_t y = pbXorwowNext(s);
  float u, v;
  if (s.kind == PB_RNG_XORWOW_ROCRAND) {  // rocrand_normal.h:56-57
    u = 2.3283064e-10f + ((float)x * 2.3283064e-10f);
    v = 1.46291807e-09f + ((float)y * 1.46291807e-09f);
  } else {  // curand_normal.h _curand_box_muller: CURAND_2POW32_INV, CURAND_2POW32_INV_2PI and their halves
    const float c = 2.3283064e-10f, c2pi = 2.3283064e-10f * 6.2831855f;
    u = (float)x * c + (c / 2.0f);
    v = (float)y * c2pi + (c2pi / 2.0f);
  }
  const float r = sqrtf(-2.0f * pbXwLog(u));
  float sn, cs;
  pbXwSinCos(v, sn, cs);
  s.boxmuller_extra = cs * r;
  s.boxmuller_flag = 1;
  return sn * r;
}

// ---- host side: the jump table -------------------------------------------------------------------
#include <vector>

// table[k] = T^(2^67 * 2^k) for the one-step matrix T of pbXorwowShift, k = 0..31, built by repeated
// squaring (99 squarings of a 160x160 GF(2) matrix: a few milliseconds)
inline void pbXorwowBuildJumpTable(uint32_t *table) {
  std::vector<uint32_t> A(PB_XW_MAT_WORDS), B(PB_XW_MAT_WORDS);
  for (int r = 0; r < PB_XW_ROWS; r++) {  // T: row r = one step applied to basis state e_r
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[r / 32] = 1u << (r % 32);
    pbXorwowShift(v);
    memcpy(&A[(size_t)r * 5], v, sizeof v);
  }
  auto square = [&]() {  // B = A * A: row r of B = (row r of A) * A
    for (int r = 0; r < PB_XW_ROWS; r++) {
      uint32_t v[5];
      memcpy(v, &A[(size_t)r * 5], sizeof v);
      pbXorwowApply(v, A.data());
      memcpy(&B[(size_t)r * 5], v, sizeof v);
    }
    A.swap(B);
  };
  for (int i = 0; i < 67; i++) square();
  for (int k = 0; k < PB_XW_TABLE_MATS; k++) {
    memcpy(table + (size_t)k * PB_XW_MAT_WORDS, A.data(), sizeof(uint32_t) * PB_XW_MAT_WORDS);
    if (k + 1 < PB_XW_TABLE_MATS) square();
  }
}
