// pb_stream.hip -- k_force_stream: the opt-in STREAMLINED force kernel (force variant 3), the one kernel
// of the engine that is not bit-identical to the reference restatement (DESIGN.md section 3
// "Streamlined arithmetic"; SURVEY.md 8(d) caveat 1).  Reference: collideD / collideSpheres
// (particlebot_kernel_impl.cuh:541-831), arithmetic algebraically streamlined.
#include <typeinfo>

#include "pb_engine.hpp"

namespace {

// Streamlined force kernel (force variant 3): same inputs, outputs and fusion as k_force, pair
// arithmetic from pbGeomS/pbFarCoefS/pbContactS.  Results are NOT bit-identical to the reference
// restatement; they stay within 1e-5 relative of it over teacher-forced windows (DESIGN.md
// "Streamlined").  Two passes per bot:
//   1. all candidates: distance, unit vector, attraction coefficient; accumulate force (and Sum|F_attr| when it
//      has a reader); a candidate in contact only has its slot pushed onto the lane's list in LDS
//   2. the lane's contacts (a handful): spring/dashpot/shear, |F|, accumulate force and Sum|F_rep|
// so the contact arithmetic runs for ~8 trips per bot instead of for every trip in which ANY lane of
// the wave is in contact (nearly all 50 in a dense blob).  One bot per lane; throughput form only.
// Round 3: ONE comparison decides the common trip (gap < near2 covers the two near bands and contact, which is
// gap < 0): the contact test, the band coefficient, the selects and the list push only run in trips in which some
// lane of the wave passes it (22 % of the trips on the bench lattice, 36 % in a random blob); ASUM = false (no
// member reads absForce_a, pbSimSetForceSums) drops the Sum|F_attr| accumulation and its store; the once-per-bot
// tail (friction, kick, actuation, integration) uses v_rcp/v_rsq forms instead of IEEE divisions and roots.
#ifndef PB_STREAM_NEAR
#define PB_STREAM_NEAR 1e-7f  // a candidate this close to contact (or closer, or in contact) goes to the contact pass
#endif
#ifndef PB_STREAM_CAP
// contacts per lane listed in LDS (further ones are evaluated in place, i.e. summed earlier than the listed ones).  ONE
// value for both walks, so that they agree bit for bit in pile-ups too (12 until round 6; 10 lets the flattened walk's
// 10 KB range queue fit beside the 10 KB of lists at 8 workgroups per CU; a bot of a blob has ~6 contacts)
#define PB_STREAM_CAP 10
#endif
#define PB_WALK_QUEUE 5  // queue entries per lane: ranges 1..4 of the compacted list + the sentinel

// static/kinetic friction and the velocity update (impl.cuh:801-825), streamlined: squared comparisons instead
// of two square roots, the kinetic-friction unit vector from one v_rsq_f32
PB_DEV void pbFrictionAndKickS(const PbDevParams &P, bool payload, float fx, float fy, float dt, float &vx, float &vy) {
  float friction = P.friction, gravity = P.gravity;
  if (payload) {
    friction *= P.frictionFactor;
    gravity *= P.massFactor;
  }
  // the static-friction hold (impl.cuh:809-811) is the other discontinuity of a step (a held bot either stays exactly
  // still or starts to move): decided exactly as the exact kernels decide it -- uncontracted squared lengths against
  // the host-computed thresholds that make `x < T` the same predicate as `sqrtf(x) < c` (PbDevParams::holdV2/holdF2)
  if (pbDot(vx, vy, vx, vy) < P.holdV2 && pbDot(fx, fy, fx, fy) < (payload ? P.holdF2Payload : P.holdF2)) {
    fx = 0.0f;
    fy = 0.0f;
  }
  const float k = payload ? dt * __builtin_amdgcn_rcpf(P.massFactor) : dt;
  vx = __builtin_fmaf(fx, k, vx);
  vy = __builtin_fmaf(fy, k, vy);
  const float fric = friction * gravity * dt;
  const float s2 = __builtin_fmaf(vx, vx, vy * vy);
  if (s2 < fric * fric) {
    vx = 0.0f;
    vy = 0.0f;
  } else {
    const float c = 1.0f - fric * __builtin_amdgcn_rsqf(s2);  // v -= fric * v/|v|
    vx *= c;
    vy *= c;
  }
}

// radius actuation (impl.cuh:124-181), streamlined: reciprocals of the (wave-uniform) constants and of the radius
// from v_rcp_f32.  Same branches, same clamps.
PB_DEV float pbActuateS(const PbDevParams &P, float rad, float phase, int dead, float absA, float absR, float time,
                        float dt) {
  if (dead) return rad;
  if (phase > 10000000.0f) return rad;
  float t1 = time + phase;
  const float period = (P.Nx + 1) * P.rise_period;
  if (t1 < 0) t1 = t1 + 100 * (P.Nx + 1) * P.rise_period;
  if (t1 >= period) t1 = t1 - period * floorf(t1 * __builtin_amdgcn_rcpf(period));
  if (t1 >= 2 * P.rise_period) return rad;
  const float slope = (P.max_radius - P.min_radius) * __builtin_amdgcn_rcpf(P.rise_period);
  const float target = t1 <= P.rise_period ? __builtin_fmaf(slope, t1, P.min_radius)
                                           : __builtin_fmaf(-slope, t1 - P.rise_period, P.max_radius);
  const float want = target - rad;
  float dr = 0;
  const float max_speed = 0.1f;
  // torque = want * constraint * rad / max_speed / max_radius / dt, capped at `constraint`
  float torque = want * P.constraint * rad * __builtin_amdgcn_rcpf(max_speed * P.max_radius * dt);
  torque = fminf(torque, P.constraint);
  if (want > 0) {
    const float tr = torque * __builtin_amdgcn_rcpf(rad);
    if (tr > absR) dr = max_speed * P.max_radius * __builtin_amdgcn_rcpf(P.constraint) * (tr - absR) * dt;
  } else {
    if (P.constrained_contraction) {
      if (-P.constraint_contraction * want > absA * rad)
        dr = __builtin_fmaf(P.constraint_contraction, want, absA * rad) * __builtin_amdgcn_rcpf(P.constraint_contraction);
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

// Round 3 also tried the neighbour rows STAGED IN LDS, one private tile per wave (tools/experiments/
// pb_stream_lds_tile.hip.txt): the loop itself then runs at the pace of the loop
// without loads (40.8 us against 52.7 us at 10^6 bots -- the vector-memory pipeline, 64 lanes x 16 B per trip,
// co-limits this kernel), but the staging prologue, the 40 KB of LDS per workgroup and the waves that need more than
// one tile (grid-row ends, the bench lattice's alternating row densities) gave it all back: 49.3 us in the best
// case, 60-78 us as a complete kernel.  Not shipped.
// WALK (round 5 experiment, shipped for this kernel in round 6; profiles/r5_blob_walk.txt, profiles/r6_blob_v3.*): how a
// lane visits its 25-cell stencil.
//   false  row by row: the wave advances stencil row by stencil row and runs the LONGEST row of its 64 lanes each time
//          (trips per wave = sum over the five rows of the longest range): lane utilisation 0.95 on the bench lattice,
//          0.73 on BASELINE configs[4]'s blobs, whose cell occupancies vary
//   true   flattened: every lane walks its own five ranges back to back (trips per wave = the longest LIST of its 64
//          lanes).  The non-empty ranges wait in a per-lane queue in LDS; one address iterator runs two elements ahead
//          of the evaluation and pops the queue when it leaves a range; a lane past the end of its list is masked.
//          Same candidates in the same order for every bot: BIT-IDENTICAL to WALK false.  -10 % on a 10^6-bot blob,
//          -7...-9 % on configs[4], 0 % on the lattice (the lanes' lists then no longer line up: 22 instead of 18
//          distinct cache lines per wave-load).  The host picks it per batch at every re-sort from the trip counts of
//          both walks (k_walk_trips, pb_engine.hip; pbSimSetStreamWalk pins it).
template <bool PAYLOAD, bool ASUM, bool WALK>
__global__ __launch_bounds__(TILE) void k_force_stream(const PbDevParams *__restrict__ params,
                                                       const float4 *__restrict__ prIn,
                                                       const float2 *__restrict__ velIn, float4 *__restrict__ prOut,
                                                       float2 *__restrict__ velOut, const float *__restrict__ phase,
                                                       const int *__restrict__ dead, float *__restrict__ absA,
                                                       float *__restrict__ absR, const uint32_t *__restrict__ orig,
                                                       const uint32_t *__restrict__ cellSAll, uint32_t n, float dt,
                                                       float timeNext, int doRadiusNext, uint32_t perXcd, int fuse) {
  constexpr int CAP = PB_STREAM_CAP;
  // one block: the range queue of the flattened walk (uint2 columns) in front of the contact lists (column = lane:
  // conflict-free)
  constexpr int QW = WALK ? 2 * PB_WALK_QUEUE : 0;
  __shared__ __attribute__((aligned(8))) uint32_t ldsBlock[(QW + CAP) * TILE];
  uint2(*const rangeQ)[TILE] = reinterpret_cast<uint2(*)[TILE]>(ldsBlock);
  uint32_t(*const contacts)[TILE] = reinterpret_cast<uint32_t(*)[TILE]>(ldsBlock + QW * TILE);
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t tile = perXcd ? (blockIdx.x & 7u) * perXcd + (blockIdx.x >> 3) : blockIdx.x;
  uint32_t l = tile * TILE + threadIdx.x;
  if (!WALK && l >= n) return;
  // (WALK has wave-wide reductions: a lane beyond the last bot stays, shadows the last bot and stores nothing)
  const bool alive = l < n;
  if (WALK && !alive) l = n - 1u;
  const uint32_t s = blockIdx.y * n + l;
  const uint32_t *__restrict__ cellS = cellSAll + (size_t)blockIdx.y * (P.numCells + 1u);

  const float4 me = prIn[s];
  float2 v = velIn[s];
  bool selfPayload = false;
  if (PAYLOAD) selfPayload = (orig[s] == P.nCells - 1u);
  const float att1 = selfPayload ? P.attractionFactor : 1.0f;
  const float attraction0 = P.attraction;
  const float slope0 = pbBandSlope(attraction0);
  const PbContactK CK{P.spring, P.damping, P.shear};
  const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;

  float fx = 0.0f, fy = 0.0f, fa = 0.0f;
  float fr = 0.0f * absR[s];  // impl.cuh:688
  uint32_t cnt = 0;

  // Round 4: the contact decision is the one discontinuity of the pair force (2.5 N of attraction floor against a
  // spring that starts at 0), and a placed blob is full of pairs that touch EXACTLY (dist == reach to the last bit;
  // held bots keep them so for hundreds of steps).  dist from v_rsq_f32 is 1-2 ulp off, which decided those pairs at
  // random: 5-8 x the flips of an FMA-contracted build of the reference's own arithmetic
  // (tests/test_gpu_fma_bracket.py).  So the first pass lists every candidate with gap < PB_STREAM_NEAR (~7 ulp of a
  // typical distance) and THIS pass, a handful of trips per bot, decides: away from the threshold by the sign of
  // the gap, within 2 * PB_STREAM_NEAR of it as the reference does -- IEEE root of the uncontracted dot product
  // against the sum of the radii (impl.cuh:549-555).  A listed pair that is not in contact gets the attraction floor
  // (gap < 0.0009: 2.5 N along n whatever the attraction constant, impl.cuh:581-583).
  auto contactOf = [&](uint32_t j, const float4 &q) __attribute__((always_inline)) {
    const float rx = q.x - me.x, ry = q.y - me.y;
    const float d2 = fmaxf(__builtin_fmaf(rx, rx, ry * ry), 1e-30f);
    PbGeomS g = pbGeomS(rx, ry, d2);
    // The spring term is 1000 * (reach - dist): every ulp of dist (1.5e-8 at 0.2) is 1.5e-5 N, the largest rounding
    // error of the whole force sum and what decides a resting bot's static-friction hold.  One Newton step on the
    // v_rsq_f32 root (exact root of d2, pbDistUnitFast) for the handful of contacts of a bot: 3 instructions per
    // contact trip, and the error of the contracted dot product is then the only one left, as in an FMA build.
    {
      const float s1 = __builtin_amdgcn_rsqf(d2), e = __builtin_fmaf(-g.dist, g.dist, d2);
      g.dist = __builtin_fmaf(e, 0.5f * s1, g.dist);
    }
    const float reach = me.z + q.z, gap = g.dist - reach;
    bool contact = gap < 0.0f;
    if (fabsf(gap) < 2.0f * PB_STREAM_NEAR) {
      const float xx = rx * rx, d2e = xx + ry * ry;  // (-ffp-contract=off: two roundings, as the reference)
      if (d2e > 0x1p-90f) contact = pbSqrtFast(d2e) < reach;  // == sqrtf on its domain (pbSelfTest: every float)
    }
    if (contact) {
      const float2 vb = velIn[j];
      float cx, cy;
      const float mag = pbContactS(CK, g, reach, vb.x - v.x, vb.y - v.y, cx, cy);
      fx += cx;
      fy += cy;
      fr += mag;
    } else {
      fx = __builtin_fmaf(fmin_attr, g.nx, fx);
      fy = __builtin_fmaf(fmin_attr, g.ny, fy);
      if (ASUM) fa += fmin_attr;
    }
  };
  // No test for the bot's own slot in the common trip: with d2 clamped away from zero the self pair has n = 0 and
  // gap = -reach, so it lands in the contact block, which skips it.  (Two distinct bots at the same point, NaN in
  // the reference, give a zero force.)  slotOf(): the candidate's global slot, only worked out when it is in contact.
  auto one = [&](const float4 &q, auto slotOf) __attribute__((always_inline)) {
    const float rx = q.x - me.x, ry = q.y - me.y;
    const float d2 = fmaxf(__builtin_fmaf(rx, rx, ry * ry), 1e-30f);
    const float inv = __builtin_amdgcn_rsqf(d2);                   // 1/dist
    // dist = the CORRECTLY ROUNDED root of d2 (x * rsq(x) + one Newton step: exact for every float of the domain,
    // pbSelfTest / tools/one_newton_root_test.hip), then gap = dist - reach rounded as the reference rounds it.  Round 4:
    // with gap = fma(d2, rsq, -reach) the distance was 1 ulp off the reference's in half the pairs, and where the
    // attraction law is steep (A / gap^2 and the band ramp: ~1.4e4 N per unit of gap below ~0.002) one ulp of 0.2 is
    // 2e-4 N -- the largest difference to the reference of the whole force sum, and what made this kernel's states
    // drift from the oracle's 2.4 x as fast as an FMA build of the reference's own arithmetic drifts
    // (tests/test_gpu_fma_bracket.py: 43 -> 14 flipped bots where the FMA build has 16).  4 instructions per trip.
    const float y0 = d2 * inv, gap = __builtin_fmaf(__builtin_fmaf(-y0, y0, d2), 0.5f * inv, y0) - (me.z + q.z);
    const float A = PAYLOAD ? attraction0 * q.w * att1 : attraction0;
    float coef = pbFarCoefS(A, gap);
    // gap < near2: one of the two near bands or contact (gap < 0) -- rare: a wave-uniform branch on one ballot
    if (__builtin_amdgcn_ballot_w64(gap < near2) != 0ull) {
      // dist < reach -- or within a few ulp of it: the contact pass decides those exactly (contactOf)
      const bool contact = gap < PB_STREAM_NEAR;
      const float K = PAYLOAD ? pbBandSlope(A) : slope0;
      const float band = gap < near1 ? fmin_attr : __builtin_fmaf(K, gap - near1, fmin_attr);
      coef = gap < near2 ? band : coef;
      coef = contact ? 0.0f : coef;
      if (contact) {
        const uint32_t j = slotOf();
        if (j != s) {  // (the bot's own slot: a zero force, not worth a trip of the second pass)
          if (cnt < (uint32_t)CAP) contacts[cnt][threadIdx.x] = j;
          else contactOf(j, q);  // list full (pathological compression): evaluate in place
          cnt++;
        }
      }
    }
    const float ci = coef * inv;  // term = coef * n = (coef / dist) * r
    fx = __builtin_fmaf(ci, rx, fx);
    fy = __builtin_fmaf(ci, ry, fy);
    if (ASUM) fa += coef;
  };

  const int gx = pbCellX(P, me.x), gy = pbCellY(P, me.y);
  const uint32_t GX = P.gridX;
  const uint32_t mx0 = (uint32_t)(gx - 2) & (GX - 1u);
  const uint32_t first = (GX - mx0) < 5u ? (GX - mx0) : 5u;
  const int nseg = first < 5u ? 2 : 1;
  // (wave-uniform) the flattened walk is written for stencil rows that do not wrap in x: one range per row
  if (WALK && __all(nseg == 1)) {
    const char *const prBytes = (const char *)prIn;
    auto at = [&](uint32_t off) __attribute__((always_inline)) { return *(const float4 *)(prBytes + off); };
    // the five ranges: ten independent cell-table reads, one round trip
    uint32_t lo[5], hi[5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const uint32_t row = ((uint32_t)(gy + r - 2) & (P.gridY - 1u)) << P.gridXLog2;
      lo[r] = cellS[row + mx0] * 16u;
      hi[r] = cellS[row + mx0 + 5u] * 16u;
    }
    // compacted list of the non-empty ones, E_0 in registers, E_1 .. E_5 in the lane's LDS column; behind the last
    // real range the sentinel (0, ~0): the iterator then walks up from the array's first slot and never leaves it
    // (a list is shorter than the array, so those reads stay inside it; they are never evaluated)
    uint32_t aEnd = 0xFFFFFFF0u, a0 = 0u, m = 0u, k = 0u;
    uint2 *const qCol = &rangeQ[0][threadIdx.x];
#pragma unroll
    for (int e = 0; e < PB_WALK_QUEUE; e++) qCol[e * TILE] = make_uint2(0u, 0xFFFFFFF0u);
#pragma unroll
    for (int r = 0; r < 5; r++) {
      if (hi[r] > lo[r]) {
        m += (hi[r] - lo[r]) >> 4;
        if (k == 0u) a0 = lo[r], aEnd = hi[r];
        else qCol[(k - 1u) * TILE] = make_uint2(lo[r], hi[r]);
        k++;
      }
    }
    if (!alive) m = 0u;
    // trips of the wave = its longest list
    uint32_t T = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) T = max(T, (uint32_t)__shfl_xor((int)T, d));
    T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
    const uint2 *qNext = qCol;
    // the address iterator: a_k = offset of list element k.  next(a): the element after a -- a + 16, or the first of
    // the next queued range when that leaves the current one
    auto next = [&](uint32_t a) __attribute__((always_inline)) {
      a += 16u;
      if (a >= aEnd) {
        const uint2 e = *qNext;
        qNext += TILE;
        a = e.x;
        aEnd = e.y;
      }
      return a;
    };
    if (T != 0u) {
      uint32_t a1 = next(a0);
      float4 q0 = at(a0), q1 = at(a1);
      uint32_t a2;
      float4 q2;
      for (uint32_t t = 0;;) {
        a2 = next(a1);
        q2 = at(a2);
        if (t < m) one(q0, [&]() { return a0 >> 4; });
        if (++t >= T) break;
        a0 = next(a2);
        q0 = at(a0);
        if (t < m) one(q1, [&]() { return a1 >> 4; });
        if (++t >= T) break;
        a1 = next(a0);
        q1 = at(a1);
        if (t < m) one(q2, [&]() { return a2 >> 4; });
        if (++t >= T) break;
      }
    }
  } else {
    // Segment loop rolled and software-pipelined two deep, as in pbSweep: while segment si runs,
    // the cell-table bounds of segment si + 2 and the first two posrad of segment si + 1 are in flight.  Inside a
    // segment posrad loads run two neighbours ahead, three registers rotating roles; the loop runs on 32-bit byte
    // offsets.  Up to two slots past a range are read (spare elements at the end of the array), never evaluated.
    const char *const prBytes = (const char *)prIn;
    auto at = [&](uint32_t off) __attribute__((always_inline)) { return *(const float4 *)(prBytes + off); };
    const uint32_t selfOff = s * 16u;
    auto bounds = [&](int si, uint32_t &lo, uint32_t &hi) __attribute__((always_inline)) {
      lo = hi = selfOff;
      if (si < 10) {
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) << P.gridXLog2;
        lo = cellS[row + ((si & 1) ? 0u : mx0)] * 16u;
        hi = cellS[row + ((si & 1) ? 5u - first : mx0 + first)] * 16u;
      }
    };
    const int stride = nseg == 1 ? 2 : 1;  // per lane: two ranges per grid row only at the x-wrap
    uint32_t loA, hiA, loB, hiB;
    bounds(0, loA, hiA);
    bounds(stride, loB, hiB);
    float4 qA0 = at(loA), qA1 = at(loA + 16u);
#pragma unroll 1
    for (int si = 0; si < 10; si += stride) {
      const uint32_t lo = loA, end = hiA;
      float4 q0 = qA0, q1 = qA1;
      loA = loB;
      hiA = hiB;
      qA0 = at(loA);
      qA1 = at(loA + 16u);
      bounds(si + 2 * stride, loB, hiB);
      if (lo < end) {
        uint32_t off = lo;
        for (;;) {
          const float4 q2 = at(off + 32u);
          one(q0, [&]() { return off >> 4; });
          if ((off += 16u) >= end) break;
          q0 = at(off + 32u);
          one(q1, [&]() { return off >> 4; });
          if ((off += 16u) >= end) break;
          q1 = at(off + 32u);
          one(q2, [&]() { return off >> 4; });
          if ((off += 16u) >= end) break;
        }
      }
    }
  }
  const uint32_t listed = cnt < (uint32_t)CAP ? cnt : (uint32_t)CAP;
  for (uint32_t k = 0; k < listed; k++) {
    const uint32_t j = contacts[k][threadIdx.x];
    contactOf(j, prIn[j]);
  }

  PbForce F{fx, fy, fa, fr};
  pbObstacles(P, me.x, me.y, v.x, v.y, me.z, F);
  pbFrictionAndKickS(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);
  float4 out = me;
  if (fuse) {
    if (doRadiusNext) out.z = pbActuateS(P, me.z, phase[s], dead[s], F.fa, F.fr, timeNext, dt);
    pbIntegrate(P, out.x, out.y, v.x, v.y, out.z, dt);
  }
  if (!WALK || alive) {
    prOut[s] = out;
    velOut[s] = v;
    if (ASUM) absA[s] = F.fa;
    absR[s] = F.fr;
  }
}

}  // namespace

namespace {
template <bool PL, bool AS, bool WK>
std::string streamNameT() {
  auto b = [](bool v) { return std::string(v ? "true" : "false"); };
  return "k_force_stream<" + b(PL) + ", " + b(AS) + ", " + b(WK) + ">" + pbKernelArgs(typeid(&k_force_stream<PL, AS, WK>).name());
}
}  // namespace

// one place maps (payload, sums kept, walk) to an instantiation: the launch and the kernel's name
#define PB_STREAM_FORMS(S, X)                                      \
  do {                                                             \
    const bool asum_ = attractionSumsKept(S), wk_ = pbStreamWalk(S); \
    if ((S)->payload) {                                            \
      if (asum_) { if (wk_) X(true, true, true); else X(true, true, false); }       \
      else { if (wk_) X(true, false, true); else X(true, false, false); }           \
    } else {                                                       \
      if (asum_) { if (wk_) X(false, true, true); else X(false, true, false); }     \
      else { if (wk_) X(false, false, true); else X(false, false, false); }         \
    }                                                              \
  } while (0)

std::string pbForceStreamName(const pbSim *S) {
  std::string name;
#define PB_NAME(PL, AS, WK) name = streamNameT<PL, AS, WK>()
  PB_STREAM_FORMS(S, PB_NAME);
#undef PB_NAME
  return name;
}

void pbLaunchForceStream(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext) {
  const uint32_t tiles = cdiv(S->n, TILE);
  const uint32_t perXcd = tiles >= 64u ? cdiv(tiles, 8u) : 0u;
  const dim3 grid(perXcd ? perXcd * 8u : tiles, S->nsims);
#define PB_STREAM(PL, AS, WK)                                                                                   \
  hipLaunchKernelGGL((k_force_stream<PL, AS, WK>), grid, dim3(TILE), 0, S->stream, S->dP, S->pr[c], S->vel[c], \
                     S->pr[o], S->vel[o], S->phase[c], S->dead[c], S->absA[c], S->absR[c], S->orig[c],         \
                     S->cellS, S->n, dt, tNext, doRadiusNext, perXcd, (int)fuse)
  PB_STREAM_FORMS(S, PB_STREAM);
#undef PB_STREAM
}
