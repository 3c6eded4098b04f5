// pb_device.hpp -- device-side parameter block and per-bot physics for gfx950.
//
// Arithmetic contract: fp32, IEEE division and sqrt (hipcc's default
// correctly-rounded forms), NO FMA contraction (-ffp-contract=off), operation order exactly as the
// reference writes it, with  powf(x,2)/__powf(x,2) -> x*x  and  powf(x,0.5f) -> sqrtf(x).
// The CPU oracle follows the same rules, so the two agree bit for bit.
//
// Reference: particlebot_kernel_impl.cuh (cited per function).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

#include "particlebot_kernel.h"

// Flattened copy of SimParams that travels to kernels BY VALUE (kernarg segment -> SGPRs); one per
// simulation, so many simulations can live in one process (the reference keeps a single
// `__constant__ SimParams params`, particlebot_kernel_impl.cuh:27).
struct PbDevParams {
  uint32_t gridX, gridY, numCells;
  float originX, originY, cellX, cellY;
  uint32_t nCells;
  int32_t nDead;
  float gravity, spring, damping, shear, attraction, boundaryDamping, friction;
  float massFactor, frictionFactor, attractionFactor;
  float constraint, constraint_contraction;
  float light_x, light_y;
  float min_radius, max_radius, rise_period;
  int32_t Nx;
  uint32_t light_shadow, constrained_contraction, seed;
  float phase_std;
  float wallHalf;
  int32_t nobstacles;
  float x1obs[PB_MAX_OBSTACLES], x2obs[PB_MAX_OBSTACLES], y1obs[PB_MAX_OBSTACLES], y2obs[PB_MAX_OBSTACLES];
  int32_t n_cir;
  float xc[PB_MAX_OBSTACLES], yc[PB_MAX_OBSTACLES], rc[PB_MAX_OBSTACLES];
  // derived on the host (pbFlattenParams), in the reference's fp32 operation order: values that every bot would
  // otherwise work out for itself with an IEEE division per step although they only depend on the parameters
  uint32_t gridXLog2;     // gridX is a power of two (engine: checked by pbSimCreateBatch): row * gridX as a shift
  float actUpSlope;       // (max_radius - min_radius) / rise_period      (impl.cuh:145)
  float actDownSlope;     // (min_radius - max_radius) / rise_period      (impl.cuh:147)
  float actGain;          // max_speed * max_radius / constraint          (impl.cuh:162), max_speed = 0.1f
  // static-friction hold (impl.cuh:809-811): length(v) < 0.000001f && length(F) < 2 mu g.  sqrtf is correctly rounded,
  // hence monotone: sqrtf(x) < c  <=>  x < T(c), T(c) = the smallest float whose root is >= c (found on the host by
  // bisection over the float bit patterns with the same correctly rounded sqrtf): two compares of the squared lengths
  // instead of two IEEE square roots per bot per step, same decisions for every input (NaN included: both false)
  float holdV2;           // T(0.000001f)
  float holdF2, holdF2Payload;  // T(2 friction gravity), and with the payload's frictionFactor / massFactor
};

// smallest non-negative float x with sqrtf(x) >= c (+inf if there is none among the finite ones, 0 if c <= 0 or NaN:
// then `sqrtf(x) < c` is never true, and neither is `x < 0`)
static inline float pbSqrtThreshold(float c) {
  if (!(c > 0.0f)) return 0.0f;
  uint32_t lo = 0u, hi = 0x7F800000u;  // bit patterns of +0 .. +inf: sqrtf is monotone over them
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2u;
    float x;
    memcpy(&x, &mid, 4);
    volatile float r = sqrtf(x);
    if (r >= c) hi = mid;
    else lo = mid + 1u;
  }
  float x;
  memcpy(&x, &lo, 4);
  return x;
}

static inline void pbFlattenParams(PbDevParams &d, const SimParams &p, float wallHalf) {
  d.gridX = p.gridSize.x;
  d.gridY = p.gridSize.y;
  d.numCells = p.numCells;
  d.originX = p.worldOrigin.x;
  d.originY = p.worldOrigin.y;
  d.cellX = p.cellSize.x;
  d.cellY = p.cellSize.y;
  d.nCells = p.nCells;
  d.nDead = p.nDead;
  d.gravity = p.gravity;
  d.spring = p.spring;
  d.damping = p.damping;
  d.shear = p.shear;
  d.attraction = p.attraction;
  d.boundaryDamping = p.boundaryDamping;
  d.friction = p.friction;
  d.massFactor = p.massFactor;
  d.frictionFactor = p.frictionFactor;
  d.attractionFactor = p.attractionFactor;
  d.constraint = p.constraint;
  d.constraint_contraction = p.constraint_contraction;
  d.light_x = p.light_x;
  d.light_y = p.light_y;
  d.min_radius = p.min_radius;
  d.max_radius = p.max_radius;
  d.rise_period = p.rise_period;
  d.Nx = p.Nx;
  d.light_shadow = p.light_shadow;
  d.constrained_contraction = p.constrained_contraction;
  d.seed = p.seed;
  d.phase_std = p.phase_std;
  d.wallHalf = wallHalf > 0.0f ? wallHalf : 64.0f;
  d.nobstacles = p.nobstacles < 0 ? 0 : (p.nobstacles > PB_MAX_OBSTACLES ? PB_MAX_OBSTACLES : p.nobstacles);
  d.n_cir = p.n_cir_obstacles < 0 ? 0
                                  : (p.n_cir_obstacles > PB_MAX_OBSTACLES ? PB_MAX_OBSTACLES : p.n_cir_obstacles);
  d.gridXLog2 = 0;
  while ((1u << d.gridXLog2) < d.gridX && d.gridXLog2 < 31u) d.gridXLog2++;
  {
    // (volatile: keep the compiler from folding these into anything but the plain fp32 operations of the reference)
    volatile float up = (p.max_radius - p.min_radius) / p.rise_period;
    volatile float down = (p.min_radius - p.max_radius) / p.rise_period;
    const float max_speed = 0.1f;
    volatile float gain = max_speed * p.max_radius / p.constraint;
    d.actUpSlope = up;
    d.actDownSlope = down;
    d.actGain = gain;
    d.holdV2 = pbSqrtThreshold(0.000001f);
    volatile float hold = 2.0f * p.friction * p.gravity;
    volatile float fp = p.friction * p.frictionFactor, gp = p.gravity * p.massFactor;  // pbFrictionAndKick's order
    volatile float holdP = 2.0f * fp * gp;
    d.holdF2 = pbSqrtThreshold(hold);
    d.holdF2Payload = pbSqrtThreshold(holdP);
  }
  for (int i = 0; i < PB_MAX_OBSTACLES; i++) {
    const bool r = i < d.nobstacles;
    d.x1obs[i] = (r && p.x1obs) ? p.x1obs[i] : 0.0f;
    d.x2obs[i] = (r && p.x2obs) ? p.x2obs[i] : 0.0f;
    d.y1obs[i] = (r && p.y1obs) ? p.y1obs[i] : 0.0f;
    d.y2obs[i] = (r && p.y2obs) ? p.y2obs[i] : 0.0f;
    const bool c = i < d.n_cir;
    d.xc[i] = (c && p.x_cir_obs) ? p.x_cir_obs[i] : 0.0f;
    d.yc[i] = (c && p.y_cir_obs) ? p.y_cir_obs[i] : 0.0f;
    d.rc[i] = (c && p.r_cir_obs) ? p.r_cir_obs[i] : 0.0f;
  }
}

// Per-simulation precondition of the fast exact forms.  With the arena at most 4096 half-wide any
// two bots (aliased cells pair bots from opposite ends of the arena) are less than 2^13.5 apart, so
// a nonzero unit-vector component |n| = |r|/dist is >= 2^-44 / 2^13.5 = 2^-57.5; with every attraction
// constant a pair can see 0 or in [2^-40, 2^30], pbDiv2Fast's numerators A*n are 0 or >= 2^-97.5
// (domain: >= 2^-100) and its quotients A*n/gap^2 >= 2^-97.5 / 2^27 = 2^-124.5 stay normal (an arena
// twice as wide would reach 2^-127.5: denormal, outside the domain).
static inline bool pbFastMathAllowed(const PbDevParams &d) {
  auto okA = [](float a) { return a == 0.0f || (a >= 0x1p-40f && a <= 0x1p30f); };
  bool ok = okA(d.attraction) && d.wallHalf <= 4096.0f && d.wallHalf > 0.0f;
  if (d.nDead == -1) {
    const float a1 = d.attraction * d.attractionFactor;
    ok = ok && okA(a1) && okA(a1 * d.attractionFactor);
  }
  return ok;
}

// Precondition of the both-sums throughput form's attraction magnitude (pbPairEvalXY<FAST, true>): the root of
// |term|^2 comes from pbRootNewton, exact for 0 and for [2^-96, FLT_MAX).  A non-contact term is the 2.5 N band or
// A * n / gap^2 with gap^2 < 2^27 (above) and max(|nx|, |ny|) >= 2^-0.5, so |term|^2 >= A^2 * 2^-55: inside the domain
// when every attraction constant a pair can see is 0 or >= 2^-20 (the reference's default is 4.8e-5 = 2^-14.4).  A batch
// that fails it runs that form's plain IEEE path.
static inline bool pbAttractionMagnitudeSafe(const PbDevParams &d) {
  auto okA = [](float a) { return a == 0.0f || a >= 0x1p-20f; };
  bool ok = okA(d.attraction);
  if (d.nDead == -1) {
    const float a1 = d.attraction * d.attractionFactor;
    ok = ok && okA(a1) && okA(a1 * d.attractionFactor);
  }
  return ok;
}

// the layout scripts and reference-built callers rely on (particlebot_kernel.cuh:58-120)
#include <cstddef>
// (float2/uint2 are 8-byte aligned in HIP exactly as in CUDA, hence the hole after numCells)
static_assert(offsetof(SimParams, worldOrigin) == 16 && offsetof(SimParams, nCells) == 32 &&
                  offsetof(SimParams, nobstacles) == 144 && offsetof(SimParams, x1obs) == 152 &&
                  offsetof(SimParams, Nx) == 216 && offsetof(SimParams, max_time) == 248 &&
                  sizeof(SimParams) == 256,
              "SimParams layout drifted from the reference struct");

#define PB_DEV __device__ __forceinline__

// helper_math.h:1244/1287 semantics: dot = ax*bx + ay*by (two roundings), length = sqrtf(dot)
PB_DEV float pbDot(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
PB_DEV float pbLen(float x, float y) { return sqrtf(pbDot(x, y, x, y)); }

// ---- grid (impl.cuh:106-120) ---------------------------------------------------------------
PB_DEV int pbCellX(const PbDevParams &P, float x) { return (int)floorf((x - P.originX) / P.cellX); }
PB_DEV int pbCellY(const PbDevParams &P, float y) { return (int)floorf((y - P.originY) / P.cellY); }
PB_DEV uint32_t pbHash(const PbDevParams &P, int gx, int gy) {
  return ((uint32_t)gy & (P.gridY - 1u)) * P.gridX + ((uint32_t)gx & (P.gridX - 1u));
}

// ---- integration with wall clamp (impl.cuh:53-103) -------------------------------------------
PB_DEV void pbIntegrate(const PbDevParams &P, float &px, float &py, float &vx, float &vy, float rad,
                        float dt) {
  const float W = P.wallHalf;
  px = px + vx * dt;
  py = py + vy * dt;
  if (px > W - rad) {
    px = W - rad;
    vx *= P.boundaryDamping;
  }
  if (px < -W + rad) {
    px = -W + rad;
    vx *= P.boundaryDamping;
  }
  if (py > W - rad) {
    py = W - rad;
    vy *= P.boundaryDamping;
  }
  if (py < -W + rad) {
    py = -W + rad;
    vy *= P.boundaryDamping;
  }
}

// ---- radius actuation (impl.cuh:124-181).  Returns the new radius. ---------------------------
PB_DEV float pbActuate(const PbDevParams &P, float rad, float phase, int dead, float absA, float absR,
                       float time, float dt) {
  if (dead) return rad;
  if (phase > 10000000.0f) return rad;
  float t1 = time + phase;
  const float period = (P.Nx + 1) * P.rise_period;
  if (t1 < 0) t1 = t1 + 100 * (P.Nx + 1) * P.rise_period;
  if (t1 >= period) t1 = t1 - period * floorf(t1 / period);
  if (t1 >= 2 * P.rise_period) return rad;
  float target;
  // (the two slopes and the gain below only depend on the parameters: divided once, on the host, pbFlattenParams)
  if (t1 <= P.rise_period)
    target = P.min_radius + P.actUpSlope * t1;
  else
    target = P.max_radius + P.actDownSlope * (t1 - P.rise_period);
  const float want = target - rad;
  float dr = 0;
  const float max_speed = 0.1f;
  float torque = want * P.constraint * rad / max_speed / P.max_radius / dt;
  torque = fminf(torque, P.constraint);
  if (want > 0) {
    if (torque / rad > absR) dr = P.actGain * (torque / rad - absR) * dt;
  } else {
    if (P.constrained_contraction) {
      if (-P.constraint_contraction * want > absA * rad)
        dr = (P.constraint_contraction * want + absA * rad) / (P.constraint_contraction);
      dr = fmaxf(dr, -P.max_radius * dt);
    } else {
      dr = want;
    }
  }
  float r = rad + dr;
  if (r > P.max_radius) r = P.max_radius;
  if (r < P.min_radius) r = P.min_radius;
  return r;
}

// ---- pair force (impl.cuh:541-594).  B's velocity is fetched lazily (contact only). ----------
struct PbForce {
  float fx, fy, fa, fr;
};

template <class VelFetch>
PB_DEV void pbPair(const PbDevParams &P, float ax, float ay, float avx, float avy, float ra, float bx,
                   float by, float rb, float attraction, VelFetch velB, PbForce &F) {
  const float rx = bx - ax, ry = by - ay;
  const float dist = pbLen(rx, ry);
  const float reach = ra + rb;
  float tx = 0.0f, ty = 0.0f;
  if (dist < reach) {
    const float2 vb = velB();
    const float nx = rx / dist, ny = ry / dist;
    const float rvx = vb.x - avx, rvy = vb.y - avy;
    const float vn = pbDot(rvx, rvy, nx, ny);
    const float tvx = rvx - vn * nx, tvy = rvy - vn * ny;
    const float ks = -P.spring * (reach - dist);
    tx += ks * nx;
    ty += ks * ny;
    tx += P.damping * rvx;
    ty += P.damping * rvy;
    tx += P.shear * tvx;
    ty += P.shear * tvy;
    F.fx += tx;
    F.fy += ty;
    F.fr += pbLen(tx, ty);
  } else {
    const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;
    const float gap = dist - reach;
    if (gap < near1) {
      tx += fmin_attr * (rx / dist);
      ty += fmin_attr * (ry / dist);
    } else if (gap < near2) {
      const float c = fmin_attr + (attraction / (near2 * near2) - fmin_attr) / (near2 - near1) * (gap - near1);
      tx += c * (rx / dist);
      ty += c * (ry / dist);
    } else {
      const float g2 = gap * gap;
      tx += attraction * (rx / dist) / g2;
      ty += attraction * (ry / dist) / g2;
    }
    F.fx += tx;
    F.fy += ty;
    F.fa += pbLen(tx, ty);
  }
}

// ---- exact fp32 sqrt and division without the general-case scaffolding ------------------------
// hipcc lowers sqrtf(x) to: scale up if x < 2^-96, v_sqrt_f32 (1 ulp), try the two neighbouring
// floats with one fma each, scale down, pass zero/inf through: 16 instructions.  It lowers a/d to:
// v_div_scale x2, v_rcp_f32, one Newton step, quotient + two residual corrections, v_div_fmas,
// v_div_fixup: 11 instructions.  Inside a KNOWN domain the scaling and fix-up parts are no-ops and
// the remaining instructions give the same correctly rounded result:
//   pbSqrtFast(x)      == sqrtf(x)   for x == 0 or 2^-96 <= x <= FLT_MAX (and +inf)
//   pbDiv2Fast(a,b,d)  == (a/d, b/d) for normal d with |d| <= 2^126, each numerator either +0 or
//                                    >= 2^-100 in magnitude, and a normal quotient below 2^96
// The two quotients share the reciprocal and its Newton step; each takes ONE residual correction (see below).  pbSelfTestFastMath() (C-ABI
// pbSelfTest) checks both claims on the GPU: every float for sqrt, 2^32 sampled triples for div.
PB_DEV float pbSqrtFast(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float lo = __uint_as_float(__float_as_uint(y) - 1u);
  const float hi = __uint_as_float(__float_as_uint(y) + 1u);
  const float elo = __builtin_fmaf(-lo, y, x);
  const float ehi = __builtin_fmaf(-hi, y, x);
  float r = elo <= 0.0f ? lo : y;
  r = ehi > 0.0f ? hi : r;
  return r;
}

PB_DEV void pbDiv2Fast(float a, float b, float d, float &qa, float &qb) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  // ONE residual correction per quotient (round 3).  hipcc's division makes two; with the Newton-refined reciprocal
  // the second one never changes the result: tools/one_correction_test.hip compares this form with IEEE division on
  // ALL 2^46 (denominator, numerator) mantissa pairs -- 0 differences (profiles/r3_one_correction_exhaustive.txt) --
  // and pbSelfTestDivision repeats that through the C-ABI on this very function.
  float q = a * r;
  float t = __builtin_fmaf(-d, q, a);
  qa = __builtin_fmaf(t, r, q);
  q = b * r;
  t = __builtin_fmaf(-d, q, b);
  qb = __builtin_fmaf(t, r, q);
}

// Distance and unit vector of a pair from ONE transcendental (v_rsq_f32) instead of v_sqrt_f32 + v_rcp_f32:
//   s    = v_rsq_f32(d2), clamped to FLT_MAX (d2 == 0: coincident bots must still get dist = 0, not NaN, to
//          land in the reference's contact branch)
//   dist = ONE Newton step on y = d2*s with h = s/2            == sqrtf(d2)
//   r    = one Newton step on s against dist                   (the reciprocal pbDiv2Fast would refine from v_rcp_f32)
//   n    = (rx, ry) * r with ONE residual correction each           == (rx/dist, ry/dist)
// tools/rsq_form_test.hip checks this on the GPU EXHAUSTIVELY: the square root for every float that is 0 or in
// [2^-96, FLT_MAX) (1 879 048 193 values, 0 differ from sqrtf), and the quotient for every d2 in [1, 4) -- all 2^24
// mantissa x exponent-parity cases -- against every numerator mantissa (2^47 divisions, compared with the
// compiler's IEEE a/dist; scaling d2 by 4^k and a numerator by 2^m scales every intermediate exactly inside the
// domain below).  Result (profiles/r2_rsq_form_exhaustive.txt): r equals the v_rcp_f32-seeded reciprocal for
// every d2 except the two whose root has an all-ones mantissa (dist = 2 - ulp: the true 1/dist lies 2^-49 above
// a rounding tie, a Newton step from s = 2^k lands exactly ON the tie and rounds to even); only there do
// quotients differ (2 of 2^47).  In those cases e = 1 - dist*s is exactly 2^-24, which sends the wave to the
// v_rcp_f32 form (0 mismatches over the same 2^47).  Domain: as pbSqrtFast / pbDiv2Fast (d2 == 0 or >= 2^-88
// in the kernel; numerators +0 or >= 2^-100).
// The root part on its own: sqrtf(x) for x == 0 or 2^-96 <= x < FLT_MAX from ONE v_rsq_f32 and ONE Newton step; s = the
// (clamped) reciprocal root it started from, for callers that go on to the reciprocal.  5 VALU + 1 transcendental
// against pbSqrtFast's 8 + 1.
PB_DEV float pbRootNewton(float x, float &s) {
  s = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(x), 0.0f, 0x1.fffffep127f);
  const float h = 0.5f * s;
  float y = x * s;
  // ONE Newton step (round 3; two until then): exact for every float that is 0 or in [2^-96, FLT_MAX) --
  // tools/one_newton_root_test.hip, profiles/r3_one_newton_root_exhaustive.txt: 1 879 048 193 values, 0 differ from
  // sqrtf; pbSelfTest repeats it on pbDistUnitFast (which is this function + the unit vector) at the start of every
  // GPU test session
  const float e = __builtin_fmaf(-y, y, x);
  return __builtin_fmaf(e, h, y);
}

PB_DEV void pbDistUnitFast(float rx, float ry, float d2, float &dist, float &nx, float &ny) {
  float s;
  const float y = pbRootNewton(d2, s);
  dist = y;
  const float er = __builtin_fmaf(-y, s, 1.0f);
  float r = __builtin_fmaf(er, s, s);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(er == 0x1p-24f) != 0ull, 0)) {
    asm volatile("; rare: reciprocal of an all-ones root, v_rcp_f32 form" ::: "memory");
    const float c = __builtin_amdgcn_rcpf(y);
    r = __builtin_fmaf(__builtin_fmaf(-y, c, 1.0f), c, c);
  }
  // one residual correction per component (round 3): exact on all 2^47 (d2, numerator) mantissa pairs, the two
  // all-ones roots (through the branch above) included -- tools/one_correction_test.hip, pbSelfTestPairGeometry
  float q = rx * r;
  float t = __builtin_fmaf(-y, q, rx);
  nx = __builtin_fmaf(t, r, q);
  q = ry * r;
  t = __builtin_fmaf(-y, q, ry);
  ny = __builtin_fmaf(t, r, q);
}

// x is a non-negative float (or NaN): true when it is nonzero but below 2^-96, the only
// non-negative inputs for which pbSqrtFast may differ from sqrtf
PB_DEV bool pbTinyNonzero(float x) { return __float_as_uint(x) - 1u < 0x0F800000u - 1u; }

// A lane may use the fast forms for all its pairs when neither of its coordinates is within 2^-20
// of zero: then every nonzero coordinate difference is >= 2^-44 (it is a multiple of the ulp of a
// number >= 2^-20), which keeps every numerator in pbDiv2Fast's domain (see DESIGN.md "Fast exact
// math" for the chain of bounds, including the host-side check on the attraction constants).
PB_DEV bool pbLaneFastMathOk(float x, float y) { return fabsf(x) >= 0x1p-20f && fabsf(y) >= 0x1p-20f; }

// ---- the same pair force, evaluated without divergent branches --------------------------------
// In a dense blob every 64-lane wave holds, at every neighbour iteration, some lane in contact,
// some in the constant-attraction band and most in the 1/gap^2 tail, so the branchy form above
// executes ALL its paths each iteration.  Here the shared work (distance, unit vector) is hoisted,
// every regime's coefficient is computed once, and selects pick the result -- the selected value
// is produced by exactly the operations of the taken branch above, so results are bit-identical.
//
// K of the linear band, (A/near2^2 - min)/(near2-near1) (impl.cuh:585-586), only depends on A
PB_DEV float pbBandSlope(float attraction) {
  const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;
  return (attraction / (near2 * near2) - fmin_attr) / (near2 - near1);
}

// live: false for the lane's own slot (the reference skips j == index, impl.cuh:638); such a lane
// computes on garbage (0/0) and accumulates nothing.  velB is only called when some lane of the
// wave is in contact, and only by those lanes.
// FAST: use pbSqrtFast / pbDiv2Fast; the caller guarantees their domains (pbLaneFastMathOk for
// every lane of the wave + the per-simulation check pbFastMathAllowed), except for the force
// magnitude's square root, whose input is checked here wave-wide and sent to sqrtf if tiny.
//
// Split in two so that a loop can evaluate several neighbours' forces as independent instruction
// streams (the evaluation is one long dependent chain) and then add them in the reference's order.
struct PbPairTerm {
  float tx, ty, mag;
  bool contact;
};

template <bool FAST, class VelFetch>
PB_DEV PbPairTerm pbPairEval(const PbDevParams &P, bool live, float ax, float ay, float avx, float avy, float ra,
                             float bx, float by, float rb, float attraction, float slope, VelFetch velB) {
  const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;
  const float rx = bx - ax, ry = by - ay;
  const float d2 = pbDot(rx, ry, rx, ry);
  float dist, nx, ny;
  if (FAST) {
    dist = pbSqrtFast(d2);  // d2 is 0 or >= 2^-88 here (coordinate differences are 0 or >= 2^-44)
    pbDiv2Fast(rx, ry, dist, nx, ny);
  } else {
    dist = sqrtf(d2);
    nx = rx / dist;
    ny = ry / dist;
  }
  const float reach = ra + rb;
  const bool contact = dist < reach;
  // no contact: constant band, linear band, inverse-square tail
  const float gap = dist - reach;
  const float g2 = gap * gap;
  float farx, fary;
  if (FAST) {
    pbDiv2Fast(attraction * nx, attraction * ny, g2, farx, fary);
  } else {
    farx = attraction * nx / g2;
    fary = attraction * ny / g2;
  }
  const float band = gap < near1 ? fmin_attr : fmin_attr + slope * (gap - near1);
  float tx = gap < near2 ? band * nx : farx;
  float ty = gap < near2 ? band * ny : fary;
  // contact: spring + dashpot + shear
  if (__builtin_amdgcn_ballot_w64(contact && live) != 0ull) {
    float2 vb = make_float2(0.0f, 0.0f);
    if (contact) vb = velB();
    const float rvx = vb.x - avx, rvy = vb.y - avy;
    const float vn = pbDot(rvx, rvy, nx, ny);
    const float tvx = rvx - vn * nx, tvy = rvy - vn * ny;
    const float ks = -P.spring * (reach - dist);
    float cx = 0.0f, cy = 0.0f;
    cx += ks * nx;
    cy += ks * ny;
    cx += P.damping * rvx;
    cy += P.damping * rvy;
    cx += P.shear * tvx;
    cy += P.shear * tvy;
    if (contact) {
      tx = cx;
      ty = cy;
    } else {
      tx = 0.0f + tx;
      ty = 0.0f + ty;
    }
  } else {
    tx = 0.0f + tx;  // `tempforce += ...` onto (0,0): turns -0 into +0
    ty = 0.0f + ty;
  }
  const float m2 = pbDot(tx, ty, tx, ty);
  float mag;
  if (FAST) {
    mag = pbSqrtFast(m2);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && pbTinyNonzero(m2)) != 0ull, 0)) {
      // a real branch, not a select: the empty asm keeps hipcc from computing both roots every time
      asm volatile("; rare: force magnitude below 2^-48, full sqrtf" ::: "memory");
      mag = sqrtf(m2);
    }
  } else {
    mag = sqrtf(m2);
  }
  PbPairTerm r;
  r.tx = tx;
  r.ty = ty;
  r.mag = mag;
  r.contact = contact;
  return r;
}

// K neighbours evaluated side by side in the same basic blocks (one shared contact branch, one
// shared rare-sqrt branch), so the scheduler can interleave K independent dependency chains.
// Each term equals pbPairEval's for that neighbour.
// the three contact constants, copied out of the parameter block once per kernel (the block lives
// in device memory; left to itself hipcc re-reads it inside the neighbour loop)
struct PbContactK {
  float spring, damping, shear;
};

template <bool FAST, int K, class VelFetch>
PB_DEV void pbPairEvalK(const PbContactK &P, const bool (&live)[K], float ax, float ay, float avx, float avy,
                        float ra, const float (&bx)[K], const float (&by)[K], const float (&rb)[K],
                        const float (&attraction)[K], const float (&slope)[K], VelFetch velB,
                        PbPairTerm (&out)[K]) {
  const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;
  float nx[K], ny[K], dist[K], reach[K], tx[K], ty[K], gapv[K];
  bool contact[K];
  unsigned long long waveContact = 0ull, waveBand = 0ull;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const float rx = bx[k] - ax, ry = by[k] - ay;
    const float d2 = pbDot(rx, ry, rx, ry);
    if (FAST) {
      pbDistUnitFast(rx, ry, d2, dist[k], nx[k], ny[k]);
    } else {
      dist[k] = sqrtf(d2);
      nx[k] = rx / dist[k];
      ny[k] = ry / dist[k];
    }
    reach[k] = ra + rb[k];
    contact[k] = dist[k] < reach[k];
    const float gap = dist[k] - reach[k];
    const float g2 = gap * gap;
    float farx, fary;
    if (FAST) {
      pbDiv2Fast(attraction[k] * nx[k], attraction[k] * ny[k], g2, farx, fary);
    } else {
      farx = attraction[k] * nx[k] / g2;
      fary = attraction[k] * ny[k] / g2;
    }
    tx[k] = farx;
    ty[k] = fary;
    gapv[k] = gap;
    // wave-uniform masks, built from ballots of the plain comparisons (a bool carried across the
    // branches below costs two extra vector instructions per trip)
    const unsigned long long mLive = __builtin_amdgcn_ballot_w64(live[k]);
    const unsigned long long mContact = __builtin_amdgcn_ballot_w64(contact[k]);
    waveContact |= mLive & mContact;
    waveBand |= mLive & ~mContact & __builtin_amdgcn_ballot_w64(gap < near2);
  }
  // the two near bands (gap < 0.0019) are rare -- a pair crosses them in a few timesteps while it
  // makes or breaks contact -- so their coefficient is only worked out when some lane needs it
  if (waveBand != 0ull) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      const float band = gapv[k] < near1 ? fmin_attr : fmin_attr + slope[k] * (gapv[k] - near1);
      tx[k] = gapv[k] < near2 ? band * nx[k] : tx[k];
      ty[k] = gapv[k] < near2 ? band * ny[k] : ty[k];
    }
  }
  if (waveContact != 0ull) {
    float2 vb[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      vb[k] = make_float2(0.0f, 0.0f);
      if (contact[k]) vb[k] = velB(k);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const float rvx = vb[k].x - avx, rvy = vb[k].y - avy;
      const float vn = pbDot(rvx, rvy, nx[k], ny[k]);
      const float tvx = rvx - vn * nx[k], tvy = rvy - vn * ny[k];
      const float ks = -P.spring * (reach[k] - dist[k]);
      // The reference builds each term as `tempforce = (0,0); tempforce += ...`; the leading 0 +
      // only turns a -0 into +0.  A term's sign of zero is unobservable: it is consumed by
      // force += term (force is never -0, so adding either zero changes nothing) and by
      // length(term) (squares), so the adds are dropped here.  (pbPair keeps them.)
      float cx = ks * nx[k];
      float cy = ks * ny[k];
      cx += P.damping * rvx;
      cy += P.damping * rvy;
      cx += P.shear * tvx;
      cy += P.shear * tvy;
      tx[k] = contact[k] ? cx : tx[k];
      ty[k] = contact[k] ? cy : ty[k];
    }
  }
  float m2[K];
  bool anyTiny = false;
#pragma unroll
  for (int k = 0; k < K; k++) {
    m2[k] = pbDot(tx[k], ty[k], tx[k], ty[k]);
    out[k].tx = tx[k];
    out[k].ty = ty[k];
    out[k].contact = contact[k];
    if (FAST) {
      out[k].mag = pbSqrtFast(m2[k]);
      anyTiny = anyTiny || pbTinyNonzero(m2[k]);  // (a non-live lane's 0/0 is NaN: not "tiny")
    } else {
      out[k].mag = sqrtf(m2[k]);
    }
  }
  if (FAST) {
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(anyTiny) != 0ull, 0)) {
      asm volatile("; rare: a force magnitude below 2^-48, full sqrtf" ::: "memory");
#pragma unroll
      for (int k = 0; k < K; k++) out[k].mag = sqrtf(m2[k]);
    }
  }
}

// ---- the same pair term when Sum|F_attr| is a dead value ---------------------------------------
// The reference keeps two magnitude sums per bot: absForce_r (contact terms) and absForce_a (all the
// others).  Both are private scratch arrays (no getArray case, not in the dump); the only reader is
// the next step's updateRad_light_wave, and it reads absForce_a solely under
// `if (params.constrained_contraction)` (impl.cuh:167-169; 0 by default and in every shipped example).
// With that switch off the attraction magnitudes -- a square root per candidate pair -- are dead
// stores, and this form leaves them out: it returns the term (tx, ty) only.  A contact's squared
// magnitude is handed to `pushRep` (same lanes, same order as the reference's `absforce_r +=`); the
// caller takes the roots later, a handful per bot instead of one per trip (pbRepList below).
// Every value that IS produced comes from exactly the operations of pbPairEvalK.
struct PbPairXY {
  float tx, ty;
};

// WANT_A (round 5: the throughput form that keeps BOTH sums): also return, in magA, the magnitude of the term of a
// lane that is NOT in contact -- length(tempforce) of the attraction branches (impl.cuh:580-592), the root of the
// uncontracted dot product of the rounded components, as pbPairEvalK forms it -- and the lane's contact flag; a
// contact's magnitude still goes through pushRep.  magA of a contact lane or of a lane that is not live is garbage.
template <bool FAST, bool WANT_A = false, class VelFetch, class PushRep>
PB_DEV PbPairXY pbPairEvalXY(const PbContactK &P, bool live, float ax, float ay, float avx, float avy, float ra,
                             float bx, float by, float rb, VelFetch velB, float attraction, float slope,
                             PushRep pushRep, float *magA = nullptr, bool *isContact = nullptr) {
  const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;
  const float rx = bx - ax, ry = by - ay;
  const float d2 = pbDot(rx, ry, rx, ry);
  float dist, nx, ny;
  if (FAST) {
    pbDistUnitFast(rx, ry, d2, dist, nx, ny);
  } else {
    dist = sqrtf(d2);
    nx = rx / dist;
    ny = ry / dist;
  }
  const float reach = ra + rb;
  const bool contact = dist < reach;
  const float gap = dist - reach;
  const float g2 = gap * gap;
  float tx, ty;
  if (FAST) {
    pbDiv2Fast(attraction * nx, attraction * ny, g2, tx, ty);
  } else {
    tx = attraction * nx / g2;
    ty = attraction * ny / g2;
  }
  // One comparison decides the common case: gap < near2 covers the two near bands AND contact (a contact is
  // gap < 0: for finite floats dist < reach <=> dist - reach < 0, and NaN fails both).  Only a trip in which
  // some live lane passes it works out which lanes are which.
  const unsigned long long mNear = __builtin_amdgcn_ballot_w64(live) & __builtin_amdgcn_ballot_w64(gap < near2);
  if (mNear != 0ull) {
    const unsigned long long mContact = __builtin_amdgcn_ballot_w64(contact);
    if ((mNear & ~mContact) != 0ull) {
      const float band = gap < near1 ? fmin_attr : fmin_attr + slope * (gap - near1);
      tx = gap < near2 ? band * nx : tx;
      ty = gap < near2 ? band * ny : ty;
    }
    if ((mNear & mContact) != 0ull) {
      // (lanes out of contact compute on their neighbour's velocity too; their result is not selected)
      const float2 vb = velB();
      const float rvx = vb.x - avx, rvy = vb.y - avy;
      const float vn = pbDot(rvx, rvy, nx, ny);
      const float tvx = rvx - vn * nx, tvy = rvy - vn * ny;
      const float ks = -P.spring * (reach - dist);
      float cx = ks * nx;
      float cy = ks * ny;
      cx += P.damping * rvx;
      cy += P.damping * rvy;
      cx += P.shear * tvx;
      cy += P.shear * tvy;
      pushRep(contact && live, pbDot(cx, cy, cx, cy));
      tx = contact ? cx : tx;
      ty = contact ? cy : ty;
    }
  }
  if (WANT_A) {
    const float m2 = pbDot(tx, ty, tx, ty);
    float mag;
    if (FAST) {
      // (no check for a tiny m2: the caller only takes the FAST path for batches that pass pbAttractionMagnitudeSafe)
      float s;
      mag = pbRootNewton(m2, s);
    } else {
      mag = sqrtf(m2);
    }
    *magA = mag;
    *isContact = contact;
  }
  PbPairXY r;
  r.tx = tx;
  r.ty = ty;
  return r;
}

// A lane's pending contact magnitudes: squared values parked in LDS (one column per lane: entry k of
// lane t at col[k * STRIDE], conflict-free), roots taken and added to Sum|F_rep| in list order by
// flush().  push() is called by every lane of the wave inside the wave-uniform contact block: lanes
// out of contact write to their next free entry without claiming it (no exec-mask juggling), which is
// why a column has CAP + 1 entries.  A full column anywhere in the wave flushes the whole wave --
// the sum's order does not change, only when the additions happen.
template <bool FAST, int CAP, int STRIDE>
struct PbRepList {
  float *col;    // this lane's column
  uint32_t off;  // next free entry, in floats from col (a multiple of STRIDE)
  PB_DEV void init(float *column) {
    col = column;
    off = 0u;
  }
  PB_DEV void flush(float &fr) {
    for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(k < off) != 0ull; k += STRIDE) {
      const float m2 = col[k];
      float mag;
      if (FAST) {
        mag = pbSqrtFast(m2);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(k < off && pbTinyNonzero(m2)) != 0ull, 0)) {
          asm volatile("; rare: a contact magnitude below 2^-48, full sqrtf" ::: "memory");
          mag = sqrtf(m2);
        }
      } else {
        mag = sqrtf(m2);
      }
      if (k < off) fr += mag;
    }
    off = 0u;
  }
  PB_DEV void push(bool mine, float m2, float &fr) {
    col[off] = m2;
    off += mine ? (uint32_t)STRIDE : 0u;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(off == (uint32_t)(CAP * STRIDE)) != 0ull, 0)) flush(fr);
  }
};

PB_DEV void pbPairAdd(bool live, const PbPairTerm &t, PbForce &F) {
  if (live) {
    F.fx += t.tx;
    F.fy += t.ty;
    if (t.contact) F.fr += t.mag;
    else F.fa += t.mag;
  }
}

template <bool FAST, class VelFetch>
PB_DEV void pbPairFlat(const PbDevParams &P, bool live, float ax, float ay, float avx, float avy, float ra,
                       float bx, float by, float rb, float attraction, float slope, VelFetch velB,
                       PbForce &F) {
  pbPairAdd(live, pbPairEval<FAST>(P, live, ax, ay, avx, avy, ra, bx, by, rb, attraction, slope, velB), F);
}

// ---- streamlined pair force (force variant 3; NOT bit-identical) ----
// The same physics as collideSpheres (impl.cuh:541-594) in the algebraically streamlined form
// SURVEY.md 8(d) describes: one reciprocal square root gives the distance and the unit vector
// (v_rsq_f32, 1 ulp), 1/gap^2 is one v_rcp_f32 (1 ulp), a non-contact term's magnitude is its
// coefficient (|n| = 1) instead of length(), products and sums may contract to FMAs, and the
// caller adds a bot's contact terms after its attraction terms instead of interleaved.  Every
// operation is within 1-2 ulp of the reference's, which is the class of difference the reference's
// own CUDA build has against any restatement (__powf, nvcc's FMA contraction); the tests hold it to
// BASELINE.json's 1e-5 relative tolerance over teacher-forced windows.
struct PbGeomS {
  float nx, ny, dist;
};
// unit vector and distance from a to b; d2 must be nonzero (the caller substitutes 1 for the self slot)
PB_DEV PbGeomS pbGeomS(float rx, float ry, float d2) {
  const float inv = __builtin_amdgcn_rsqf(d2);
  PbGeomS g;
  g.dist = d2 * inv;
  g.nx = rx * inv;
  g.ny = ry * inv;
  return g;
}
// non-contact coefficient c >= 0: the term is c*n and its magnitude is c (impl.cuh:579-592)
PB_DEV float pbFarCoefS(float attraction, float gap) { return attraction * __builtin_amdgcn_rcpf(gap * gap); }
PB_DEV float pbBandCoefS(float attraction, float gap) {
  const float near1 = 0.0009f, fmin_attr = 2.5f;
  return gap < near1 ? fmin_attr : __builtin_fmaf(pbBandSlope(attraction), gap - near1, fmin_attr);
}
// contact term: spring + dashpot + shear (impl.cuh:553-573); returns its magnitude
PB_DEV float pbContactS(const PbContactK &P, const PbGeomS &g, float reach, float rvx, float rvy, float &cx,
                        float &cy) {
  const float vn = __builtin_fmaf(rvx, g.nx, rvy * g.ny);
  const float tvx = __builtin_fmaf(-vn, g.nx, rvx), tvy = __builtin_fmaf(-vn, g.ny, rvy);
  const float ks = -P.spring * (reach - g.dist);
  cx = __builtin_fmaf(P.shear, tvx, __builtin_fmaf(P.damping, rvx, ks * g.nx));
  cy = __builtin_fmaf(P.shear, tvy, __builtin_fmaf(P.damping, rvy, ks * g.ny));
  return __builtin_amdgcn_sqrtf(__builtin_fmaf(cx, cx, cy * cy));
}

// common tail of obstacle contacts (impl.cuh:711-726 and :781-797): spring term (sx,sy) along the
// contact normal (dx,dy), dashpot and shear against the bot's own velocity
PB_DEV void pbObstacleTail(const PbDevParams &P, float vx, float vy, float dx, float dy, float sx, float sy,
                           PbForce &F) {
  const float rvx = -vx, rvy = -vy;
  const float vn = pbDot(rvx, rvy, dx, dy);
  const float tvx = rvx - vn * dx, tvy = rvy - vn * dy;
  float tx = 0.0f, ty = 0.0f;
  tx += sx;
  ty += sy;
  tx += P.damping * rvx;
  ty += P.damping * rvy;
  tx += P.shear * tvx;
  ty += P.shear * tvy;
  F.fx += tx;
  F.fy += ty;
  F.fr += pbLen(tx, ty);
}

// circular (impl.cuh:703-728) and rectangular (impl.cuh:729-798) obstacles
PB_DEV void pbObstacles(const PbDevParams &P, float px, float py, float vx, float vy, float rad, PbForce &F) {
  for (int k = 0; k < P.n_cir; k++) {
    const float ox = P.xc[k], oy = P.yc[k], orad = P.rc[k];
    const float ex = px - ox, ey = py - oy;
    const float d2 = ex * ex + ey * ey;
    const float reach = rad + orad;
    if (d2 < reach * reach) {
      float dx = -px + ox, dy = -py + oy;
      const float l = pbLen(dx, dy);
      dx = dx / l;
      dy = dy / l;
      const float ks = 2.0f * P.spring * (rad + orad - sqrtf(d2));
      pbObstacleTail(P, vx, vy, dx, dy, ks * (-dx), ks * (-dy), F);
    }
  }
  float dx = 0.0f, dy = 0.0f, overlap = 0.0f;
  for (int k = 0; k < P.nobstacles; k++) {
    const float x1 = P.x1obs[k], x2 = P.x2obs[k], y1 = P.y1obs[k], y2 = P.y2obs[k];
    bool hit = false;
    if (py > y1 && py < y2) {
      if (px > x1 - rad && px < x2 - rad) {
        hit = true;
        dx = 1.0f;
        dy = 0.0f;
        overlap = px - x1 + rad;
      }
      if (px < x2 + rad && px > x1 + rad) {
        hit = true;
        dx = -1.0f;
        dy = 0.0f;
        overlap = -px + x2 + rad;
      }
    } else if (px > x1 && px < x2) {
      if (py > y1 - rad && py < y2 - rad) {
        hit = true;
        dx = 0.0f;
        dy = 1.0f;
        overlap = py - y1 + rad;
      }
      if (py < y2 + rad && py > y1 + rad) {
        hit = true;
        dx = 0.0f;
        dy = -1.0f;
        overlap = -py + y2 + rad;
      }
    } else {
      // corners, first match wins, in the reference's order (x2,y2) (x1,y2) (x1,y1) (x2,y1)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float cx = (c == 0 || c == 3) ? x2 : x1;
        const float cy = (c < 2) ? y2 : y1;
        const float ex = px - cx, ey = py - cy;
        const float d2 = ex * ex + ey * ey;
        if (!hit && d2 < rad * rad) {
          const float l = pbLen(ex, ey);
          dx = -ex / l;
          dy = -ey / l;
          overlap = rad - sqrtf(d2);
          hit = true;
        }
      }
    }
    if (hit) {
      const float ks = -2.0f * P.spring * overlap;
      pbObstacleTail(P, vx, vy, dx, dy, ks * dx, ks * dy, F);
    }
  }
}

// static/kinetic friction and the velocity update (impl.cuh:801-825)
PB_DEV void pbFrictionAndKick(const PbDevParams &P, bool payload, float fx, float fy, float dt, float &vx,
                              float &vy) {
  float friction = P.friction;
  float gravity = P.gravity;
  if (payload) {
    friction *= P.frictionFactor;
    gravity *= P.massFactor;
  }
  // length(v) < 0.000001f && length(F) < 2 friction gravity, on the squared lengths (PbDevParams::holdV2 / holdF2)
  if (pbDot(vx, vy, vx, vy) < P.holdV2 && pbDot(fx, fy, fx, fy) < (payload ? P.holdF2Payload : P.holdF2)) {
    fx = 0.0f;
    fy = 0.0f;
  }
  if (payload) {
    vx = vx + fx / P.massFactor * dt;
    vy = vy + fy / P.massFactor * dt;
  } else {
    vx = vx + fx * dt;
    vy = vy + fy * dt;
  }
  const float fric = friction * gravity * dt;
  const float speed = pbLen(vx, vy);
  if (speed < fric) {
    vx = 0.0f;
    vy = 0.0f;
  } else {
    vx -= fric * (vx / speed);
    vy -= fric * (vy / speed);
  }
}

// ---- light shadow tests (impl.cuh:184-262) ---------------------------------------------------
PB_DEV int pbSegHit(float x0, float y0, float x1, float y1, float x3, float y3, float x4, float y4) {
  if (fabsf((x4 - x3) / (x1 - x0)) == fabsf((y4 - y3) / (y1 - y0))) return 0;
  float t, t1;
  if (fabsf(y4 - y3) > 0) {
    t = (x3 - x0 - (y3 - y0) * (x3 - x4) / (y3 - y4)) *
        ((y3 - y4) / ((x1 - x0) * (y3 - y4) - (y1 - y0) * (x3 - x4)));
    if (t <= 0 || t >= 1) return 0;
    t1 = (y3 - y0 - t * (y1 - y0)) / (y3 - y4);
    if (t1 <= 0 || t1 >= 1) return 0;
  } else if (fabsf(x4 - x3) > 0) {
    t = (y3 - y0 - (x3 - x0) * (y3 - y4) / (x3 - x4)) *
        ((x3 - x4) / ((y1 - y0) * (x3 - x4) - (x1 - x0) * (y3 - y4)));
    if (t <= 0 || t >= 1) return 0;
    t1 = (x3 - x0 - t * (x1 - x0)) / (x3 - x4);
    if (t1 <= 0 || t1 >= 1) return 0;
  } else {
    return 0;
  }
  return 1;
}

PB_DEV int pbCircleHit(float lx, float ly, float px, float py, float ox, float oy, float orad) {
  const float C1 = lx * lx + ly * ly;
  const float C2 = px * px + py * py;
  const float C3 = ox * ox + oy * oy;
  const float C4 = lx * px + ly * py;
  const float C5 = lx * ox + ly * oy;
  const float C6 = px * ox + py * oy;
  const float A = C1 + C2 - 2 * C4;
  const float B = -2 * C1 + 2 * C4 + 2 * C5 - 2 * C6;
  const float C = C1 + C3 - 2 * C5 - orad * orad;
  const float D = B * B - 4 * A * C;
  if (D >= 0) {
    const float R1 = (-B + sqrtf(D)) / 2 / A;
    const float R2 = (-B - sqrtf(D)) / 2 / A;
    if (R1 > 0 && R1 < 1) return 1;
    if (R2 > 0 && R2 < 1) return 1;
  }
  return 0;
}

PB_DEV int pbInShadow(const PbDevParams &P, float px, float py) {
  for (int i = 0; i < P.n_cir; i++)
    if (pbCircleHit(P.light_x, P.light_y, px, py, P.xc[i], P.yc[i], P.rc[i])) return 1;
  for (int i = 0; i < P.nobstacles; i++) {
    const float x1 = P.x1obs[i], x2 = P.x2obs[i], y1 = P.y1obs[i], y2 = P.y2obs[i];
    if (pbSegHit(P.light_x, P.light_y, px, py, x1, y1, x1, y2)) return 1;
    if (pbSegHit(P.light_x, P.light_y, px, py, x1, y2, x2, y2)) return 1;
    if (pbSegHit(P.light_x, P.light_y, px, py, x2, y2, x2, y1)) return 1;
    if (pbSegHit(P.light_x, P.light_y, px, py, x2, y1, x1, y1)) return 1;
  }
  return 0;
}

// phase from distance to the light (impl.cuh:264-290); `old` is returned when shadowed with
// light_shadow not in {1,2}
PB_DEV float pbPhase(const PbDevParams &P, float px, float py, float spacing, float min_d, float old) {
  const float dist = pbLen(px - P.light_x, py - P.light_y);
  bool visible = true;
  if (P.light_shadow) {
    if (pbInShadow(P, px, py)) visible = false;
  }
  if (!visible) {
    float ph = old;
    if (P.light_shadow == 1) ph = -(P.Nx - 1) * P.rise_period;
    if (P.light_shadow == 2) ph = 9999999999.0f;
    return ph;
  }
  return (min_d - dist) / (spacing)*P.rise_period;
}

// ---- PB-RNG v1: counter-based standard normal for (seed, bot, draw) ---------------------------
// Replaces cuRAND XORWOW + curand_normal (impl.cuh:36-51), which cannot be pinned here.
// Polynomial log / sin / cos in plain fp32 so CPU and GPU agree bit for bit.
PB_DEV uint64_t pbMix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}

PB_DEV float pbNormal(uint32_t seed, uint32_t bot, uint32_t draw) {
  uint64_t x = (((uint64_t)seed << 32) | (uint64_t)bot) + 0x9E3779B97F4A7C15ull * (uint64_t)(draw + 1u);
  x = pbMix64(x);
  x = pbMix64(x ^ 0xD1342543DE82EF95ull);
  const uint32_t k1 = (uint32_t)(x >> 40) & 0xFFFFFFu;
  const uint32_t k2 = (uint32_t)(x >> 8) & 0xFFFFFFu;
  const uint32_t v = k1 + 1u;  // u1 = v * 2^-24 in (0,1]
  int e = 31 - __clz((int)v);
  float m = (float)v * __uint_as_float((uint32_t)(127 - e) << 23);  // exact, in [1,2)
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
  const float lnu = 2.0f * t * p + (float)(e - 24) * 0.693147181f;
  const float r = sqrtf(-2.0f * lnu);
  const uint32_t q = k2 >> 22;
  const float a = (float)(k2 & 0x3FFFFFu) * (1.0f / 4194304.0f) * 1.57079633f;
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
  const float cv = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
  return r * cv;
}

// ---- error helpers shared by the host code of every translation unit -------------------------
#include <cstdio>
#include <cstdlib>

// legacy boundary: print + exit(EXIT_FAILURE), as include/helper_cuda.h:1000-1029 does
#define PB_CHECK_ABORT(expr)                                                                         \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      fprintf(stderr, "HIP error at %s:%d code=%d(%s) \"%s\"\n", __FILE__, __LINE__, (int)e_,        \
              hipGetErrorName(e_), #expr);                                                           \
      exit(EXIT_FAILURE);                                                                            \
    }                                                                                                \
  } while (0)
