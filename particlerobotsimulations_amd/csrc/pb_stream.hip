// pb_stream.hip -- k_force_stream: the opt-in STREAMLINED force kernel (force variant 3), the one kernel
// of the engine that is not bit-identical to the reference restatement (DESIGN.md section 5
// "Streamlined arithmetic"; SURVEY.md 8(d) caveat 1).  Reference: collideD / collideSpheres
// (particlebot_kernel_impl.cuh:541-831), arithmetic algebraically streamlined.
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
#define PB_STREAM_CAP 12  // contacts per lane listed in LDS (further ones are evaluated in place)
#endif

// static/kinetic friction and the velocity update (impl.cuh:801-825), streamlined: squared comparisons instead
// of two square roots, the kinetic-friction unit vector from one v_rsq_f32
PB_DEV void pbFrictionAndKickS(const PbDevParams &P, bool payload, float fx, float fy, float dt, float &vx, float &vy) {
  float friction = P.friction, gravity = P.gravity;
  if (payload) {
    friction *= P.frictionFactor;
    gravity *= P.massFactor;
  }
  const float hold = 2.0f * friction * gravity;
  if (__builtin_fmaf(vx, vx, vy * vy) < 1e-12f && __builtin_fmaf(fx, fx, fy * fy) < hold * hold) {
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
// pb_stream_lds_tile.hip.txt; DESIGN.md section 5 "Streamlined"): the loop itself then runs at the pace of the loop
// without loads (40.8 us against 52.7 us at 10^6 bots -- the vector-memory pipeline, 64 lanes x 16 B per trip,
// co-limits this kernel), but the staging prologue, the 40 KB of LDS per workgroup and the waves that need more than
// one tile (grid-row ends, the bench lattice's alternating row densities) gave it all back: 49.3 us in the best
// case, 60-78 us as a complete kernel.  Not shipped.
template <bool PAYLOAD, bool ASUM>
__global__ __launch_bounds__(TILE) void k_force_stream(const PbDevParams *__restrict__ params,
                                                       const float4 *__restrict__ prIn,
                                                       const float2 *__restrict__ velIn, float4 *__restrict__ prOut,
                                                       float2 *__restrict__ velOut, const float *__restrict__ phase,
                                                       const int *__restrict__ dead, float *__restrict__ absA,
                                                       float *__restrict__ absR, const uint32_t *__restrict__ orig,
                                                       const uint32_t *__restrict__ cellSAll, uint32_t n, float dt,
                                                       float timeNext, int doRadiusNext, uint32_t perXcd, int fuse) {
  __shared__ uint32_t contacts[PB_STREAM_CAP][TILE];  // column = lane: conflict-free
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t tile = perXcd ? (blockIdx.x & 7u) * perXcd + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t l = tile * TILE + threadIdx.x;
  if (l >= n) return;
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
      const float h = 0.5f * __builtin_amdgcn_rsqf(d2), e = __builtin_fmaf(-g.dist, g.dist, d2);
      g.dist = __builtin_fmaf(e, h, g.dist);
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
    const float gap = __builtin_fmaf(d2, inv, -(me.z + q.z));      // dist - reach
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
          if (cnt < (uint32_t)PB_STREAM_CAP) contacts[cnt][threadIdx.x] = j;
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
  {
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
  const uint32_t listed = cnt < (uint32_t)PB_STREAM_CAP ? cnt : (uint32_t)PB_STREAM_CAP;
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
  prOut[s] = out;
  velOut[s] = v;
  if (ASUM) absA[s] = F.fa;
  absR[s] = F.fr;
}


// ---- round 4: the streamlined kernel with ONE LDS PATCH PER WORKGROUP (stream form 1) -----------------------------
// VERDICT round 3 item 4.  Round 3 showed that the vector-memory pipeline co-limits k_force_stream (the same
// instruction stream without neighbour loads: 41.8 us against 52.7) and that a wave-private tile loses the gain to its
// prologue, to 40 KB of LDS per workgroup and to waves that span two grid rows.  This form stages the neighbourhood
// ONCE PER WORKGROUP:
//   * a tile is the <= 256 bots filed under a BLOCK of cells, PB_BAND grid rows high (pbCutBand, k_build_tiles: cut at
//     each re-sort on the device; one 4-byte read-back gives the launch its grid); XCD-aware tile order;
//   * per tile: the lanes' CURRENT cells (lists are stale between re-sorts, impl.cuh:680) are reduced to a bounding
//     box, the box + 2 cells of stencil is <= PB_BAND + 6 grid rows x one contiguous slot range each, staged by all
//     four waves with coalesced 16-B loads into a float4 image (16 KB), and every lane walks its 5 x 5 stencil out of
//     LDS with ds_read_b128; contact lists hold 16-bit patch indices (5 KB);
//   * a tile whose box does not fit (more than PB_PATCH_SLOTS slots or PB_BAND + 6 rows, or touching the grid's
//     x-wrap) walks global memory instead (plain loop; rare by construction).
// (First built over strips of ONE grid row: 64.7 us per step at 10^6 bots against form 0's 53.0 -- sparse rows limit a
//  strip to ~200 bots, 30 KB of LDS leave 5 workgroups per CU, and a row's bots spread over three rows once they move.)
// Arithmetic, summation order within a lane and results are those of k_force_stream.
#ifndef PB_PATCH_SLOTS
#define PB_PATCH_SLOTS 1024  // 16 KB of float4 records
#endif
#define PB_PATCH_ROWS (PB_BAND + 6)  // the band, two stencil rows on each side, one row of drift on each side
// the same list length as form 0: a bot with more contacts evaluates the surplus in place, i.e. in another order
#define PB_PATCH_CAP PB_STREAM_CAP
// a contact-list entry: patch index (< 2^11) | patch row << 11, in 16 bits
#define PB_PATCH_TAG_SHIFT 11
static_assert(PB_PATCH_SLOTS + 4 <= (1 << PB_PATCH_TAG_SHIFT) && PB_PATCH_ROWS <= 16, "contact-list entries are 16 bits");
// what a tile's neighbourhood may need AS FILED (band + 4 rows) so that one more row of drift on each side still fits
#define PB_PATCH_BUDGET ((PB_PATCH_SLOTS * (PB_BAND + 4)) / (PB_BAND + 6) - 32)

// Tiles of one BAND of PB_BAND grid rows, greedily along x: a tile is the bots filed under the cells of columns
// c0..c1 of the band's rows -- PB_BAND slot ranges -- and ends where one more column would exceed TILE bots or push
// its neighbourhood as filed (columns c0-2..c1+2 of rows band-2..band+PB_BAND+1) over `budget` slots.  A column that
// alone exceeds either (pathological compression) is cut row by row into runs of <= TILE slots.  Two-dimensional
// tiles instead of strips of one row: the staged neighbourhood is (PB_BAND+4)/PB_BAND of... 3x the tile's own bots
// instead of 5x, it fits 16 KB with room for the rows that stale lists add (a while after a re-sort the bots filed
// under one row sit in three), and tiles fill their 256 lanes whatever the row densities are.
// emit(tile) per tile; returns the number of tiles.
template <class Emit>
__device__ uint32_t pbCutBand(const uint32_t *__restrict__ cellS, uint32_t GX, uint32_t GY, uint32_t band, uint32_t budget,
                              Emit emit) {
  const uint32_t R = band * (uint32_t)PB_BAND;
  size_t row[PB_BAND];
  for (int j = 0; j < PB_BAND; j++) row[j] = (size_t)(R + (uint32_t)j) * GX;
  // nothing filed under the band at all?
  uint32_t any = 0;
  for (int j = 0; j < PB_BAND; j++)
    if (R + (uint32_t)j < GY) any += cellS[row[j] + GX] - cellS[row[j]];
  if (!any) return 0;
  auto need = [&](uint32_t c0, uint32_t c1) {
    const uint32_t a = c0 >= 2u ? c0 - 2u : 0u, b = c1 + 3u <= GX ? c1 + 3u : GX;
    uint32_t sum = 0;
    for (int dr = -2; dr < PB_BAND + 2; dr++) {
      const size_t rr = (size_t)((R + (uint32_t)dr) & (GY - 1u)) * GX;
      sum += cellS[rr + b] - cellS[rr + a];
    }
    return sum;
  };
  // The tile's rows are listed by falling bot count: lanes are dealt to waves in that order, so a wave mostly holds
  // bots of rows of like density, whose stencil rows have like lengths (the sweep runs stencil row by stencil row: a
  // wave that mixes a dense and a sparse row of the bench lattice makes 75 trips instead of 60).
  auto emitCols = [&](uint32_t c0, uint32_t c1) {
    uint32_t st[PB_BAND], ct[PB_BAND];
    for (int j = 0; j < PB_BAND; j++) {
      const bool in = R + (uint32_t)j < GY;
      st[j] = in ? cellS[row[j] + c0] : 0u;
      ct[j] = in ? cellS[row[j] + c1 + 1u] - st[j] : 0u;
    }
    for (int a = 1; a < PB_BAND; a++)  // insertion sort, stable
      for (int b = a; b > 0 && ct[b] > ct[b - 1]; b--) {
        const uint32_t ts = st[b], tc = ct[b];
        st[b] = st[b - 1];
        ct[b] = ct[b - 1];
        st[b - 1] = ts;
        ct[b - 1] = tc;
      }
    PbTile t;
    uint32_t acc = 0;
    for (int j = 0; j < PB_BAND; j++) {
      t.start[j] = st[j];
      acc += ct[j];
      t.cum[j] = acc;
    }
    emit(t);
  };
  uint32_t nt = 0, c0 = 0, count = 0;
  bool open = false;
  for (uint32_t x = 0; x < GX; x++) {
    uint32_t col = 0;
    for (int j = 0; j < PB_BAND; j++)
      if (R + (uint32_t)j < GY) col += cellS[row[j] + x + 1u] - cellS[row[j] + x];
    if (!col) continue;
    if (open && (count + col > (uint32_t)TILE || need(c0, x) > budget)) {
      emitCols(c0, x - 1u);  // (empty columns at the tile's end cost nothing)
      nt++;
      open = false;
    }
    if (!open) {
      if (col > (uint32_t)TILE || need(x, x) > budget) {
        // one column too many for a tile: its rows one at a time, in runs of <= TILE slots
        for (int j = 0; j < PB_BAND; j++) {
          if (R + (uint32_t)j >= GY) continue;
          const uint32_t lo = cellS[row[j] + x], hi = cellS[row[j] + x + 1u];
          for (uint32_t a = lo; a < hi; a += (uint32_t)TILE) {
            PbTile t;
            const uint32_t c = hi - a < (uint32_t)TILE ? hi - a : (uint32_t)TILE;
            for (int i = 0; i < PB_BAND; i++) {
              t.start[i] = a;
              t.cum[i] = i < j ? 0u : c;
            }
            emit(t);
            nt++;
          }
        }
        continue;
      }
      open = true;
      c0 = x;
      count = 0;
    }
    count += col;
  }
  if (open) {
    emitCols(c0, GX - 1u);
    nt++;
  }
  return nt;
}

__global__ __launch_bounds__(1024) void k_build_tiles(const uint32_t *__restrict__ cellS, uint32_t GX, uint32_t GY,
                                                      PbTile *__restrict__ tiles, uint32_t *__restrict__ ntilesOut,
                                                      uint32_t maxTiles, uint32_t budget) {
  __shared__ uint32_t part[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t bands = (GY + (uint32_t)PB_BAND - 1u) / (uint32_t)PB_BAND;
  const uint32_t per = (bands + 1023u) / 1024u;
  const uint32_t b0 = t * per, b1 = b0 + per < bands ? b0 + per : bands;
  uint32_t cnt = 0;
  for (uint32_t b = b0; b < b1; b++) cnt += pbCutBand(cellS, GX, GY, b, budget, [](const PbTile &) {});
  part[t] = cnt;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1) {
    const uint32_t v = t >= d ? part[t - d] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t at = part[t] - cnt;
  for (uint32_t b = b0; b < b1; b++)
    (void)pbCutBand(cellS, GX, GY, b, budget, [&](const PbTile &tile) {
      if (at < maxTiles) tiles[at] = tile;
      at++;
    });
  if (t == 1023u) {
    ntilesOut[0] = part[1023];  // (the host refuses a count above maxTiles: launchPatch)
    ntilesOut[1] = 0u;          // workgroups that walked global memory since this layout was cut (statistics)
  }
}

template <bool PAYLOAD, bool ASUM>
__global__ __launch_bounds__(TILE) void k_force_patch(const PbDevParams *__restrict__ params,
                                                      const float4 *__restrict__ prIn, const float2 *__restrict__ velIn,
                                                      float4 *__restrict__ prOut, float2 *__restrict__ velOut,
                                                      const float *__restrict__ phase, const int *__restrict__ dead,
                                                      float *__restrict__ absA, float *__restrict__ absR,
                                                      const uint32_t *__restrict__ orig,
                                                      const uint32_t *__restrict__ cellS,
                                                      const PbTile *__restrict__ tiles,
                                                      const uint32_t *__restrict__ ntilesPtr, float dt, float timeNext,
                                                      int doRadiusNext, int fuse) {
  __shared__ float4 patch[PB_PATCH_SLOTS + 4];
  __shared__ uint16_t contacts[PB_PATCH_CAP][TILE];
  __shared__ int wred[TILE / 64][4];
  __shared__ int rowDelta[PB_PATCH_ROWS + 1];  // patch index of a global slot of patch row r = slot + rowDelta[r]
  const PbDevParams &P = params[0];
  const uint32_t ntiles = *ntilesPtr;
  const uint32_t perXcd = (ntiles + 7u) >> 3;
  const uint32_t tile = (blockIdx.x & 7u) * perXcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= perXcd || tile >= ntiles) return;
  const PbTile T = tiles[tile];
  const uint32_t tid = threadIdx.x;
  const uint32_t count = T.cum[PB_BAND - 1];
  const bool active = tid < count;
  // lane -> (band row, slot): the tile's bots are PB_BAND slot runs, one per band row; idle lanes shadow the tile's
  // last bot (they take part in the staging and store nothing)
  const uint32_t l = active ? tid : count - 1u;
  uint32_t s = T.start[0] + l;
#pragma unroll
  for (int j = 1; j < PB_BAND; j++) s = l >= T.cum[j - 1] ? T.start[j] + (l - T.cum[j - 1]) : s;

  const float4 me = prIn[s];
  const int gx = pbCellX(P, me.x), gy = pbCellY(P, me.y);
  // bounding box of the tile's current cells: wave reduction, then four values per wave through LDS
  {
    int a = gx, b = gx, c = gy, d = gy;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      a = min(a, __shfl_xor(a, m));
      b = max(b, __shfl_xor(b, m));
      c = min(c, __shfl_xor(c, m));
      d = max(d, __shfl_xor(d, m));
    }
    if ((tid & 63u) == 0u) {
      wred[tid >> 6][0] = a;
      wred[tid >> 6][1] = b;
      wred[tid >> 6][2] = c;
      wred[tid >> 6][3] = d;
    }
  }
  __syncthreads();
  int bx0 = wred[0][0], bx1 = wred[0][1], by0 = wred[0][2], by1 = wred[0][3];
#pragma unroll
  for (int w = 1; w < TILE / 64; w++) {
    bx0 = min(bx0, wred[w][0]);
    bx1 = max(bx1, wred[w][1]);
    by0 = min(by0, wred[w][2]);
    by1 = max(by1, wred[w][3]);
  }
  const int x0 = __builtin_amdgcn_readfirstlane(bx0) - 2, x1 = __builtin_amdgcn_readfirstlane(bx1) + 2;
  const int R0 = __builtin_amdgcn_readfirstlane(by0) - 2;
  const int nrows = __builtin_amdgcn_readfirstlane(by1) + 2 - R0 + 1;
  const uint32_t GX = P.gridX;
  bool fits = x0 >= 0 && x1 < (int)GX && nrows <= PB_PATCH_ROWS;
  // one contiguous slot range per patch row (wave-uniform: scalar loads)
  uint32_t rlo[PB_PATCH_ROWS], rbase[PB_PATCH_ROWS + 1];
  rbase[0] = 0;
#pragma unroll
  for (int r = 0; r < PB_PATCH_ROWS; r++) {
    uint32_t lo = 0, hi = 0;
    if (fits && r < nrows) {
      const uint32_t row = ((uint32_t)(R0 + r) & (P.gridY - 1u)) << P.gridXLog2;
      lo = cellS[row + (uint32_t)x0];
      hi = cellS[row + (uint32_t)x1 + 1u];
    }
    rlo[r] = lo;
    rbase[r + 1] = rbase[r] + (hi - lo);
  }
  const uint32_t totalSlots = rbase[PB_PATCH_ROWS];
  fits = fits && totalSlots <= (uint32_t)PB_PATCH_SLOTS;

  const float att1 = (PAYLOAD && orig[s] == P.nCells - 1u) ? P.attractionFactor : 1.0f;
  const bool selfPayload = PAYLOAD && orig[s] == P.nCells - 1u;
  float2 v = velIn[s];
  const float attraction0 = P.attraction;
  const float slope0 = pbBandSlope(attraction0);
  const PbContactK CK{P.spring, P.damping, P.shear};
  const float near1 = 0.0009f, near2 = 0.0019f, fmin_attr = 2.5f;
  float fx = 0.0f, fy = 0.0f, fa = 0.0f;
  float fr = 0.0f * absR[s];  // impl.cuh:688

  // the contact pass of k_force_stream (see there): exact decision near the threshold, Newton-refined distance
  auto contactOf = [&](uint32_t j, const float4 &q) __attribute__((always_inline)) {
    const float rx = q.x - me.x, ry = q.y - me.y;
    const float d2 = fmaxf(__builtin_fmaf(rx, rx, ry * ry), 1e-30f);
    PbGeomS g = pbGeomS(rx, ry, d2);
    {
      const float h = 0.5f * __builtin_amdgcn_rsqf(d2), e = __builtin_fmaf(-g.dist, g.dist, d2);
      g.dist = __builtin_fmaf(e, h, g.dist);
    }
    const float reach = me.z + q.z, gap = g.dist - reach;
    bool contact = gap < 0.0f;
    if (fabsf(gap) < 2.0f * PB_STREAM_NEAR) {
      const float xx = rx * rx, d2e = xx + ry * ry;
      if (d2e > 0x1p-90f) contact = pbSqrtFast(d2e) < reach;
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
  // one candidate; listIt() runs for a candidate that belongs on the contact list (inside the rare block)
  auto one = [&](const float4 &q, auto listIt) __attribute__((always_inline)) {
    const float rx = q.x - me.x, ry = q.y - me.y;
    const float d2 = fmaxf(__builtin_fmaf(rx, rx, ry * ry), 1e-30f);
    const float inv = __builtin_amdgcn_rsqf(d2);
    const float gap = __builtin_fmaf(d2, inv, -(me.z + q.z));
    // (.w is 1 for every bot but the payload, k_set_state: the product is exact, and using the fourth component keeps
    //  the LDS read a ds_read_b128 -- 4 LDS cycles; the compiler would shrink it to a ds_read_b96, 8 cycles)
    const float A = PAYLOAD ? attraction0 * q.w * att1 : attraction0 * q.w;
    float coef = pbFarCoefS(A, gap);
    if (__builtin_amdgcn_ballot_w64(gap < near2) != 0ull) {
      const bool contact = gap < PB_STREAM_NEAR;
      const float K = PAYLOAD ? pbBandSlope(A) : slope0;
      const float band = gap < near1 ? fmin_attr : __builtin_fmaf(K, gap - near1, fmin_attr);
      coef = gap < near2 ? band : coef;
      coef = contact ? 0.0f : coef;
      if (contact) listIt();
    }
    const float ci = coef * inv;
    fx = __builtin_fmaf(ci, rx, fx);
    fy = __builtin_fmaf(ci, ry, fy);
    if (ASUM) fa += coef;
  };

  uint32_t cnt = 0;
  if (__builtin_amdgcn_readfirstlane(fits ? 1 : 0)) {
    // ---- stage the patch: 6 coalesced 16-B loads per lane, all in flight, then the LDS writes ----
    if (tid < (uint32_t)PB_PATCH_ROWS) {
      int dlt = 0;
#pragma unroll
      for (int r = 0; r < PB_PATCH_ROWS; r++)
        if ((int)tid == r) dlt = (int)rbase[r] - (int)rlo[r];
      rowDelta[tid] = dlt;
    }
    constexpr int CHUNKS = (PB_PATCH_SLOTS + TILE - 1) / TILE;
    float4 stg[CHUNKS];
#pragma unroll
    for (int k = 0; k < CHUNKS; k++) {
      const uint32_t i = (uint32_t)k * TILE + tid;
      uint32_t src = rlo[0] + i;
#pragma unroll
      for (int r = 1; r < PB_PATCH_ROWS; r++) src = i >= rbase[r] ? rlo[r] + (i - rbase[r]) : src;
      stg[k] = prIn[i < totalSlots ? src : s];  // (beyond the patch: any valid slot; not written below)
    }
#pragma unroll
    for (int k = 0; k < CHUNKS; k++) {
      const uint32_t i = (uint32_t)k * TILE + tid;
      if (i < totalSlots) patch[i] = stg[k];
    }
    __syncthreads();
    if (!active) return;

    const char *const pb = (const char *)patch;
    auto at = [&](uint32_t off) __attribute__((always_inline)) { return *(const float4 *)(pb + off); };
    const int prMine = gy - R0;  // patch row of the lane's own grid row (2 .. nrows-3)
    // bounds of stencil row k (0..4): patch BYTE offsets; the cell table stays in global memory (two loads per row,
    // issued two rows ahead)
    // (self: where the lane's OWN slot would sit in that patch row -- with stale lists a bot is not necessarily filed
    //  under the row it is in now, so the test is on the global slot, row by row)
    auto bounds = [&](int k, uint32_t &lo, uint32_t &hi, uint32_t &tag, uint32_t &self) __attribute__((always_inline)) {
      lo = hi = 0u;
      tag = self = 0u;
      if (k < 5) {
        const int pr = prMine + k - 2;
        const uint32_t row = ((uint32_t)(gy + k - 2) & (P.gridY - 1u)) << P.gridXLog2;
        const int dl = rowDelta[pr];
        lo = (uint32_t)((int)cellS[row + (uint32_t)(gx - 2)] + dl) * 16u;
        hi = (uint32_t)((int)cellS[row + (uint32_t)(gx + 3)] + dl) * 16u;
        tag = (uint32_t)pr << PB_PATCH_TAG_SHIFT;
        self = (uint32_t)((int)s + dl) * 16u;
      }
    };
    // rows rolled and software-pipelined as in k_force_stream: while row k runs, the cell-table bounds of row k + 2
    // and the first two records of row k + 1 are in flight; inside a row the LDS reads run two candidates ahead,
    // three registers rotating roles.  Up to two records past a range are read (never evaluated).
    uint32_t loA, hiA, tagA, selfA, loB, hiB, tagB, selfB;
    bounds(0, loA, hiA, tagA, selfA);
    bounds(1, loB, hiB, tagB, selfB);
    float4 qA0 = at(loA), qA1 = at(loA + 16u);
#pragma unroll 1
    for (int k = 0; k < 5; k++) {
      const uint32_t lo = loA, end = hiA, tag = tagA, selfOff = selfA;
      float4 q0 = qA0, q1 = qA1;
      loA = loB;
      hiA = hiB;
      tagA = tagB;
      selfA = selfB;
      qA0 = at(loA);
      qA1 = at(loA + 16u);
      bounds(k + 2, loB, hiB, tagB, selfB);
      if (lo < end) {
        uint32_t off = lo;
        auto push = [&](const float4 &q) __attribute__((always_inline)) {
          if (off != selfOff) {
            if (cnt < (uint32_t)PB_PATCH_CAP) contacts[cnt][tid] = (uint16_t)((off >> 4) | tag);
            else contactOf((off >> 4) - (uint32_t)rowDelta[tag >> PB_PATCH_TAG_SHIFT], q);
            cnt++;
          }
        };
        for (;;) {
          const float4 q2 = at(off + 32u);
          one(q0, [&]() { push(q0); });
          if ((off += 16u) >= end) break;
          q0 = at(off + 32u);
          one(q1, [&]() { push(q1); });
          if ((off += 16u) >= end) break;
          q1 = at(off + 32u);
          one(q2, [&]() { push(q2); });
          if ((off += 16u) >= end) break;
        }
      }
    }
    const uint32_t listed = cnt < (uint32_t)PB_PATCH_CAP ? cnt : (uint32_t)PB_PATCH_CAP;
    for (uint32_t k = 0; k < listed; k++) {
      const uint32_t e = contacts[k][tid], li = e & ((1u << PB_PATCH_TAG_SHIFT) - 1u);
      contactOf(li - (uint32_t)rowDelta[e >> PB_PATCH_TAG_SHIFT], patch[li]);
    }
  } else {
    // ---- the tile's box does not fit the patch: walk global memory (x-wrap: two ranges per row); the contact
    //      list (global slots) lives in the unused patch memory, so the order of the additions is form 0's ----
    if (tid == 0u) atomicAdd(const_cast<uint32_t *>(ntilesPtr) + 1, 1u);  // statistics (pbSimGetStreamStats)
    if (!active) return;
    uint32_t(*glist)[TILE] = reinterpret_cast<uint32_t(*)[TILE]>(patch);
    const uint32_t mx0 = (uint32_t)(gx - 2) & (GX - 1u);
    const uint32_t first = (GX - mx0) < 5u ? (GX - mx0) : 5u;
    for (int k = 0; k < 5; k++) {
      const uint32_t row = ((uint32_t)(gy + k - 2) & (P.gridY - 1u)) << P.gridXLog2;
      for (int half = 0; half < (first < 5u ? 2 : 1); half++) {
        const uint32_t lo = cellS[row + (half ? 0u : mx0)], hi = cellS[row + (half ? 5u - first : mx0 + first)];
        for (uint32_t j = lo; j < hi; j++) {
          const float4 q = prIn[j];
          one(q, [&]() {
            if (j != s) {
              if (cnt < (uint32_t)PB_PATCH_CAP) glist[cnt][tid] = j;
              else contactOf(j, q);
              cnt++;
            }
          });
        }
      }
    }
    const uint32_t listed = cnt < (uint32_t)PB_PATCH_CAP ? cnt : (uint32_t)PB_PATCH_CAP;
    for (uint32_t k = 0; k < listed; k++) {
      const uint32_t j = glist[k][tid];
      contactOf(j, prIn[j]);
    }
  }

  PbForce F{fx, fy, fa, fr};
  pbObstacles(P, me.x, me.y, v.x, v.y, me.z, F);
  pbFrictionAndKickS(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);
  float4 out = me;
  if (fuse) {
    if (doRadiusNext) out.z = pbActuateS(P, me.z, phase[s], dead[s], F.fa, F.fr, timeNext, dt);
    pbIntegrate(P, out.x, out.y, v.x, v.y, out.z, dt);
  }
  prOut[s] = out;
  velOut[s] = v;
  if (ASUM) absA[s] = F.fa;
  absR[s] = F.fr;
}

}  // namespace

// stream form 1 (pbSimSetStreamForm): one arena, the LDS-patch kernel over row-aligned tiles
static bool launchPatch(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext) {
  const uint32_t GX = S->hP[0].gridX, GY = S->hP[0].gridY;
  // room for the tile table: a closed tile is followed by a column that did not fit, so two consecutive tiles of a
  // band hold more than TILE bots between them unless the slot budget cut them short; oversized columns cost extra
  const uint32_t maxTiles = 4u * cdiv(S->n, TILE) + 2u * cdiv(GY, PB_BAND) + 64u;
  if (!S->tiles) {
    if (hipMalloc(&S->tiles, sizeof(PbTile) * (size_t)maxTiles) != hipSuccess) return false;
    if (hipMalloc(&S->ntiles, 4 * sizeof(uint32_t)) != hipSuccess) return false;
    S->tilesEpoch = ~0ull;
  }
  if (S->tilesEpoch != S->layoutEpoch) {  // the slot layout changed (re-sort, restored layout): cut the bands anew
    hipLaunchKernelGGL(k_build_tiles, dim3(1), dim3(1024), 0, S->stream, S->cellS, GX, GY, S->tiles, S->ntiles, maxTiles,
                       (uint32_t)PB_PATCH_BUDGET);
    // the launch grid needs the count on the host: one 4-byte read-back per re-sort (every 18 000 steps)
    if (hipMemcpyAsync(&S->tilesHost, S->ntiles, sizeof(uint32_t), hipMemcpyDeviceToHost, S->stream) != hipSuccess ||
        hipStreamSynchronize(S->stream) != hipSuccess)
      return false;
    S->tilesEpoch = S->layoutEpoch;
  }
  if (S->tilesHost == 0u || S->tilesHost > maxTiles) return false;  // (a layout the table cannot hold: form 0 runs)
  const dim3 grid(8u * cdiv(S->tilesHost, 8u));
  const bool asum = attractionSumsKept(S);
#define PB_PATCH(PL, AS)                                                                                         \
  hipLaunchKernelGGL((k_force_patch<PL, AS>), grid, dim3(TILE), 0, S->stream, S->dP, S->pr[c], S->vel[c], S->pr[o], \
                     S->vel[o], S->phase[c], S->dead[c], S->absA[c], S->absR[c], S->orig[c], S->cellS, S->tiles,    \
                     S->ntiles, dt, tNext, doRadiusNext, (int)fuse)
  if (S->payload && asum) PB_PATCH(true, true);
  else if (S->payload) PB_PATCH(true, false);
  else if (asum) PB_PATCH(false, true);
  else PB_PATCH(false, false);
#undef PB_PATCH
  return true;
}

void pbLaunchForceStream(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext) {
  if (S->streamForm == 1 && S->nsims == 1 && launchPatch(S, fuse, c, o, dt, tNext, doRadiusNext)) return;
  const uint32_t tiles = cdiv(S->n, TILE);
  const uint32_t perXcd = tiles >= 64u ? cdiv(tiles, 8u) : 0u;
  const dim3 grid(perXcd ? perXcd * 8u : tiles, S->nsims);
  const bool asum = attractionSumsKept(S);
#define PB_STREAM(PL, AS)                                                                                   \
  hipLaunchKernelGGL((k_force_stream<PL, AS>), grid, dim3(TILE), 0, S->stream, S->dP, S->pr[c], S->vel[c], \
                     S->pr[o], S->vel[o], S->phase[c], S->dead[c], S->absA[c], S->absR[c], S->orig[c],     \
                     S->cellS, S->n, dt, tNext, doRadiusNext, perXcd, (int)fuse)
  if (S->payload && asum) PB_STREAM(true, true);
  else if (S->payload) PB_STREAM(true, false);
  else if (asum) PB_STREAM(false, true);
  else PB_STREAM(false, false);
#undef PB_STREAM
}
