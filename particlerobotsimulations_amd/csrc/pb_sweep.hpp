// pb_sweep.hpp -- the neighbour sweep of one bot over the stale cell lists (device code shared by the
// per-step force kernel k_force, pb_force.hip, and the resident multi-step kernel, pb_resident.hip).
// Reference: collideD's 25-cell loop + collideCell (particlebot_kernel_impl.cuh:597-700).
#pragma once

#include <type_traits>

#include "pb_device.hpp"
#include "pb_engine.hpp"

#ifndef PB_FORCE_WAVES
#define PB_FORCE_WAVES 1
#endif
#ifndef PB_NB2_WAVES
#define PB_NB2_WAVES 8  // minimum waves per SIMD the two-neighbours-per-trip form is compiled for
#endif
#ifndef PB_THROUGHPUT_NB
#define PB_THROUGHPUT_NB 1
#endif
#ifndef PB_PREFETCH_DEPTH
#define PB_PREFETCH_DEPTH 1  // neighbours in flight ahead of the one being evaluated (throughput sweep)
#endif
#ifndef PB_REP_CAP
#define PB_REP_CAP 8  // pending contact magnitudes per lane before the wave flushes (PbRepList)
#endif
// NB (template parameter of k_force): neighbours evaluated side by side per loop trip of the
// one-lane-per-bot form.  1 is what ships.  2 (two independent dependency chains per wave, the
// software-pipelined two-wide loop in pbSweepC) is a build-time experiment: measured on MI355X at
// 10^6 bots it is bit-identical and SLOWER at every register budget -- 124.9 us/step at 8 waves/SIMD
// (64 VGPRs, 28 spilled), 120.8 at 7 (72), 119.8 at 6 (80), 121.4 at 5 (81, no bound) against 114.0
// for NB = 1 (63 VGPRs, 8 waves/SIMD): waves, not ILP inside a wave, are what fills the VALU pipe.

#ifndef PB_ASUM_XY
#define PB_ASUM_XY 1  // throughput form with both sums: 1 = the dead-sum trip + the attraction magnitude (round 5), 0 = pbPairEvalK
#endif
#ifndef PB_LAZY_VEL
#define PB_LAZY_VEL 1  // (0: the round-2 form, the neighbour's velocity prefetched with every posrad)
#endif
#ifndef PB_SWEEP_EXPERIMENT
#define PB_SWEEP_EXPERIMENT 0
#endif
#ifndef PB_TL_STAMP
#define PB_TL_STAMP(word) do { } while (0)  // (pb_force.hip defines it in the -DPB_TIMELINE diagnostic build)
#endif

// Ordered sum over the L lanes of a group, as a systolic chain: every lane holds the group's running
// sums F (identical in all L lanes) and its own term t.  Step 1: a = F + t.  Steps 2..L: a = (a of the
// lane to the left, a DPP row_shr:1 operand of the add itself) + t.  After L steps the group's LAST
// lane holds ((F + t_0) + t_1) + ... + t_{L-1} -- the reference's order -- and broadcasts it back
// (ds_swizzle).  Lanes further left hold partial chains that started in a neighbouring group; they
// are never used.  A lane without a term (the bot's own slot, the tail of the list) adds +0, which
// changes nothing (the sums are never -0).  4 quantities x (L adds + 1 broadcast) instructions per
// trip; the former form (every lane fetching and adding all L terms itself) took ~12 L.
// value of the lane to the left: inside a 16-lane DPP row for groups of up to 16 lanes (row_shr:1), across the
// whole wave for groups of 32 or 64 (wave_shr:1, gfx9)
template <int L>
__device__ __forceinline__ float pbShr1(float v) {
  if (L <= 16)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, false));
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
template <int L>
__device__ __forceinline__ float pbGroupLast(float v) {
  // broadcast the value of the group's last lane to its L lanes (L <= 32: ds_swizzle bit-mask mode inside
  // 32-lane halves; L == 64: the wave's last lane through an SGPR)
  if (L == 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  constexpr int PAT = ((L - 1) << 5) | (0x1F & ~(L - 1));
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PAT));
}
template <int L>
__device__ __forceinline__ void pbGroupSum(bool live, const PbPairTerm &t, PbForce &F) {
  const float tx = live ? t.tx : 0.0f, ty = live ? t.ty : 0.0f;
  const float ta = (live && !t.contact) ? t.mag : 0.0f, tr = (live && t.contact) ? t.mag : 0.0f;
  float ax = F.fx + tx, ay = F.fy + ty, aa = F.fa + ta, ar = F.fr + tr;
#pragma unroll
  for (int e = 1; e < L; e++) {
    ax = pbShr1<L>(ax) + tx;
    ay = pbShr1<L>(ay) + ty;
    aa = pbShr1<L>(aa) + ta;
    ar = pbShr1<L>(ar) + tr;
  }
  F.fx = pbGroupLast<L>(ax);
  F.fy = pbGroupLast<L>(ay);
  F.fa = pbGroupLast<L>(aa);
  F.fr = pbGroupLast<L>(ar);
}

// the same chain for one quantity / for the force components only (dead-sum form, pbPairEvalXY)
template <int L>
__device__ __forceinline__ void pbGroupSum1(float t, float &f) {
  float a = f + t;
#pragma unroll
  for (int e = 1; e < L; e++) a = pbShr1<L>(a) + t;
  f = pbGroupLast<L>(a);
}
template <int L>
__device__ __forceinline__ void pbGroupSumXY(bool live, const PbPairXY &t, PbForce &F) {
  const float tx = live ? t.tx : 0.0f, ty = live ? t.ty : 0.0f;
  float ax = F.fx + tx, ay = F.fy + ty;
#pragma unroll
  for (int e = 1; e < L; e++) {
    ax = pbShr1<L>(ax) + tx;
    ay = pbShr1<L>(ay) + ty;
  }
  F.fx = pbGroupLast<L>(ax);
  F.fy = pbGroupLast<L>(ay);
}

// Flattened neighbour list of one bot (L > 1 form): plain scalars passed by value, so that they
// stay in registers wherever the sweep is inlined (arrays or by-reference captures here ended up in
// scratch memory with data-dependent indices).
struct PbSegList {
  uint32_t o0, o1, o2, o3, o4, o5, o6, o7, o8, o9;  // slot = list position + o_r inside segment r
  uint32_t c1, c2, c3, c4, c5, c6, c7, c8, c9;      // first list position of segments 1..9
  __device__ __forceinline__ void set(int r, uint32_t off, uint32_t start) {
    switch (r) {
      case 0: o0 = off; break;
      case 1: o1 = off, c1 = start; break;
      case 2: o2 = off, c2 = start; break;
      case 3: o3 = off, c3 = start; break;
      case 4: o4 = off, c4 = start; break;
      case 5: o5 = off, c5 = start; break;
      case 6: o6 = off, c6 = start; break;
      case 7: o7 = off, c7 = start; break;
      case 8: o8 = off, c8 = start; break;
      default: o9 = off, c9 = start; break;
    }
  }
};
__device__ __forceinline__ uint32_t pbSegSlot(const PbSegList SL, uint32_t m, uint32_t self, uint32_t k) {
  uint32_t o = SL.o0;
  o = k >= SL.c1 ? SL.o1 : o;
  o = k >= SL.c2 ? SL.o2 : o;
  o = k >= SL.c3 ? SL.o3 : o;
  o = k >= SL.c4 ? SL.o4 : o;
  o = k >= SL.c5 ? SL.o5 : o;
  o = k >= SL.c6 ? SL.o6 : o;
  o = k >= SL.c7 ? SL.o7 : o;
  o = k >= SL.c8 ? SL.o8 : o;
  o = k >= SL.c9 ? SL.o9 : o;
  return k < m ? k + o : self;  // beyond the list: the bot's own slot, never accumulated
}
// the same when no row of the stencil wraps (segments 1, 3, 5, 7, 9 are empty): half the chain
__device__ __forceinline__ uint32_t pbSegSlot5(const PbSegList SL, uint32_t m, uint32_t self, uint32_t k) {
  uint32_t o = SL.o0;
  o = k >= SL.c2 ? SL.o2 : o;
  o = k >= SL.c4 ? SL.o4 : o;
  o = k >= SL.c6 ? SL.o6 : o;
  o = k >= SL.c8 ? SL.o8 : o;
  return k < m ? k + o : self;
}

// A bot's flattened list only depends on the (stale) cell table and on the cell the bot is in; the
// resident kernel keeps it across timesteps and rebuilds it (10 table reads) only in the steps in
// which some bot of the wave has moved to another cell.
struct PbSegCache {
  PbSegList SL;
  uint32_t m;
  int gx, gy;
};

// Neighbour sweep of one bot: the 25-cell stencil as 5 grid rows x up to 2 slot ranges (x-wrap), in
// the reference's order (impl.cuh:617-655).  prIn/velIn are indexed by (global slot - base): the
// per-step kernel passes the HBM arrays and base 0, the resident kernel its LDS copy and the
// simulation's first slot.  s is the bot's own index into prIn.
// L: lanes per bot.  L == 1 is the throughput form (one bot per lane).  L > 1 (small batches that
// cannot fill the chip) gives each bot L adjacent lanes: they evaluate L candidates of the bot's
// flattened neighbour list at a time, then every lane of the group adds the L terms in list order
// (ds_swizzle broadcasts inside the group), so the sums -- and their order -- are those of L == 1.
// The serial chain per bot shrinks ~L/2-fold at ~2x the total VALU work.
// ASUM: maintain Sum|F_attr| (F.fa).  false (branch-free forms; the caller guarantees that no
// simulation of the batch has constrained_contraction set, see pbPairEvalXY): F.fa is left alone;
// in the throughput form the contact magnitudes go through the lane's LDS column repCol
// (PbRepList, columns REPSTRIDE floats apart).
template <bool PAYLOAD, bool FLAT, bool FAST, int L, int NB, bool CACHED, class PR, class VL, class OffT = uint32_t,
          bool ASUM = true, int REPSTRIDE = TILE>
__device__ __forceinline__ void pbSweepC(const PbDevParams &P, PR prIn, VL velIn,
                                        const uint32_t *__restrict__ cellS, uint32_t base, uint32_t s,
                                         uint32_t sub, const float4 &me, const float2 &v, float att1, PbForce &F,
                                         PbSegCache &cache, float *repCol = nullptr) {
  const int gx = pbCellX(P, me.x), gy = pbCellY(P, me.y);
  const float slope0 = pbBandSlope(P.attraction);
  const float attraction0 = P.attraction;
  const PbContactK CK{P.spring, P.damping, P.shear};
  const uint32_t GX = P.gridX;
  const uint32_t mx0 = (uint32_t)(gx - 2) & (GX - 1u);
  const uint32_t first = (GX - mx0) < 5u ? (GX - mx0) : 5u;  // cells before the x-wrap
  const int nseg = first < 5u ? 2 : 1;
  if (L > 1) {
    // ---- flattened candidate list, L candidates per trip, ordered group sum --------------------
    // 5 grid rows x up to 2 ranges (x-wrap) = 10 list segments; segment r covers list positions
    // [c[r], c[r+1]) and maps position k to slot k + o[r].
    PbSegList SL;
    uint32_t m;
    // (wave-uniform) rebuild unless every lane's cached list is still for the cell it is in
    if (!CACHED || __any(cache.gx != gx || cache.gy != gy)) {
      uint32_t cum = 0;
#pragma unroll
      for (int si = 0; si < 10; si++) {
        const int sg = si & 1;
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) << P.gridXLog2;
        uint32_t lo = 0, hi = 0;
        if (sg < nseg) {
          lo = cellS[row + (sg == 0 ? mx0 : 0u)] - base;
          hi = cellS[row + (sg == 0 ? mx0 + first : 5u - first)] - base;
        }
        SL.set(si, lo - cum, cum);
        cum += hi - lo;
      }
      m = cum;
      if (CACHED) {
        cache.SL = SL;
        cache.m = m;
        cache.gx = gx;
        cache.gy = gy;
      }
    } else {
      SL = cache.SL;
      m = cache.m;
    }
    // wave-uniform: away from the x-wrap (nearly always) the position -> slot chain has 5 links, not 10
    auto run = [&](auto wrapTag) __attribute__((always_inline)) {
      constexpr bool WRAP = decltype(wrapTag)::value;
      auto slotOf = [=](uint32_t k) __attribute__((always_inline)) {
        return WRAP ? pbSegSlot(SL, m, s, k) : pbSegSlot5(SL, m, s, k);
      };
      uint32_t jn = slotOf(sub);
      float4 qn = prIn[jn];
      float2 wn = velIn[jn];
      for (uint32_t b0 = 0; b0 < m; b0 += L) {
        const uint32_t j = jn;
        const float4 q = qn;
        const float2 w = wn;
        jn = slotOf(b0 + L + sub);
        qn = prIn[jn];
        wn = velIn[jn];
        const bool live[1] = {j != s};
        const float bx[1] = {q.x}, by[1] = {q.y}, rb[1] = {q.z};
        const float A[1] = {PAYLOAD ? attraction0 * q.w * att1 : attraction0};
        const float K[1] = {PAYLOAD ? pbBandSlope(A[0]) : slope0};
        if (ASUM) {
          PbPairTerm t[1];
          pbPairEvalK<FAST, 1>(CK, live, me.x, me.y, v.x, v.y, me.z, bx, by, rb, A, K, [&](int) { return w; }, t);
          // the group's L terms join the running sums in list order
          pbGroupSum<L>(live[0], t[0], F);
        } else {
          // dead-sum form: no Sum|F_attr|; a contact's magnitude and the Sum|F_rep| chain only in the
          // trips in which some lane of the wave is in contact
          const PbPairXY t = pbPairEvalXY<FAST>(
              CK, live[0], me.x, me.y, v.x, v.y, me.z, q.x, q.y, q.z, [&]() { return w; }, A[0], K[0], [&](bool mine, float m2) {
                float mag;
                if (FAST) {
                  mag = pbSqrtFast(m2);
                  if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine && pbTinyNonzero(m2)) != 0ull, 0)) {
                    asm volatile("; rare: a contact magnitude below 2^-48, full sqrtf" ::: "memory");
                    mag = sqrtf(m2);
                  }
                } else {
                  mag = sqrtf(m2);
                }
                pbGroupSum1<L>(mine ? mag : 0.0f, F.fr);
              });
          pbGroupSumXY<L>(live[0], t, F);
        }
      }
    };
    if (__all(nseg == 1)) run(std::false_type{});
    else run(std::true_type{});
    return;
  }
  if (FLAT && NB == 1) {
    // One bot per lane, one neighbour per trip (the throughput form).
    //  * The loop over the 10 segments is rolled (one copy of the pair loop in the binary) and
    //    software-pipelined two deep: while segment si runs, the cell-table bounds of segment
    //    si + 2 and the first posrad of segment si + 1 are in flight.  Loaded just in time they are
    //    two dependent memory round trips per segment, ~20 per bot, that only other waves can hide
    //    -- and at the start and the end of a launch there are none.
    //  * Inside a segment the next neighbour's posrad is already in flight, the loop is unrolled
    //    by two with the two registers swapping roles (no copy at the back-edge), and it runs on
    //    32-bit BYTE offsets from the array base (one add and one compare per trip; the
    //    neighbour's velocity sits at half the offset).  One slot past a range is still inside
    //    the array (spare elements) and is never evaluated.
    const char *const prBytes = (const char *)&prIn[0];
    const char *const velBytes = (const char *)&velIn[0];
    // OffT: 32-bit byte offsets (batches below 2^28 bots: one add and one compare per trip, loads with a
    // scalar base + 32-bit vector offset) or 64-bit ones (larger batches, up to 2^32 slots)
    const OffT selfOff = (OffT)s * 16u;
    auto at = [&](OffT off) __attribute__((always_inline)) { return *(const float4 *)(prBytes + off); };
    auto vat = [&](OffT off) __attribute__((always_inline)) { return *(const float2 *)(velBytes + (off >> 1)); };
    // (round 5: the form that keeps both sums parks its contact magnitudes in LDS too; PB_ASUM_XY 0 = the former
    //  pbPairEvalK + pbPairAdd path, kept for A/B)
    constexpr bool REPL = !ASUM || PB_ASUM_XY;
    PbRepList<FAST, PB_REP_CAP, REPSTRIDE> rep;
    if (REPL) rep.init(repCol);
    // (64-bit address arithmetic with a constant displacement: the displacement becomes the load's
    //  immediate offset, so the look-ahead loads need no address instructions of their own)
#if PB_SWEEP_EXPERIMENT >= 2
    // timing experiment (wrong results): no neighbour posrad loads at all
    auto atI = [&](OffT off, int imm) __attribute__((always_inline)) {
      float4 q = me;
      asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z) : "v"(off));
      q.x += 0.3f;
      return q;
    };
#else
    auto atI = [&](OffT off, int imm) __attribute__((always_inline)) {
      return *(const float4 *)(prBytes + (uint64_t)off + imm);
    };
#endif
#if PB_SWEEP_EXPERIMENT >= 1
    // timing experiment (wrong results): no neighbour velocity loads
    auto vatI = [&](OffT hoff, int imm) __attribute__((always_inline)) {
      float2 w = v;
      asm volatile("" : "+v"(w.x), "+v"(w.y) : "v"(hoff));
      return w;
    };
#else
    auto vatI = [&](OffT hoff, int imm) __attribute__((always_inline)) {
      return *(const float2 *)(velBytes + (uint64_t)hoff + imm);
    };
#endif
    const OffT selfOff16 = selfOff + 16u;
    auto one = [&](const float4 &q, auto velOf, bool isLive) __attribute__((always_inline)) {
      const bool live[1] = {isLive};
      const float bx[1] = {q.x}, by[1] = {q.y}, rb[1] = {q.z};
      const float A[1] = {PAYLOAD ? attraction0 * q.w * att1 : attraction0};
      const float K[1] = {PAYLOAD ? pbBandSlope(A[0]) : slope0};
      if (ASUM && !PB_ASUM_XY) {
        PbPairTerm t[1];
        pbPairEvalK<FAST, 1>(CK, live, me.x, me.y, v.x, v.y, me.z, bx, by, rb, A, K, [&](int) { return velOf(); }, t);
        pbPairAdd(live[0], t[0], F);
      } else if (ASUM) {
        // both sums: the dead-sum trip + the magnitude of the lane's attraction term (Sum|F_attr| in list order, as
        // absforce_a += length(tempforce), impl.cuh:580-592); contact magnitudes through the LDS list as below
        float magA;
        bool contact;
        const PbPairXY t = pbPairEvalXY<FAST, true>(CK, live[0], me.x, me.y, v.x, v.y, me.z, q.x, q.y, q.z, velOf, A[0],
                                                    K[0], [&](bool mine, float m2) { rep.push(mine, m2, F.fr); }, &magA,
                                                    &contact);
        if (live[0]) {
          asm volatile("");
          F.fx += t.tx;
          F.fy += t.ty;
          if (!contact) F.fa += magA;
        }
      } else {
        const PbPairXY t = pbPairEvalXY<FAST>(CK, live[0], me.x, me.y, v.x, v.y, me.z, q.x, q.y, q.z, velOf, A[0],
                                              K[0], [&](bool mine, float m2) { rep.push(mine, m2, F.fr); });
        if (live[0]) {
          // (a real exec-masked block -- two scalar instructions -- instead of two selects per trip)
          asm volatile("");
          F.fx += t.tx;
          F.fy += t.ty;
        }
      }
    };
    // byte offsets [lo, hi) of segment si; empty beyond the last one and for the second range of a
    // row away from the x-wrap
    auto bounds = [&](int si, OffT &lo, OffT &hi) __attribute__((always_inline)) {
      lo = hi = selfOff;
      if (si < 10) {
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) << P.gridXLog2;
        lo = (OffT)(cellS[row + ((si & 1) ? 0u : mx0)] - base) * 16u;
        hi = (OffT)(cellS[row + ((si & 1) ? 5u - first : mx0 + first)] - base) * 16u;
      }
    };
    // segment numbers advance by 2 (one range per grid row) except for a lane at the x-wrap, whose
    // rows split into two ranges: per-lane stride, the wave runs until its last lane is done
    const int stride = nseg == 1 ? 2 : 1;
    OffT loA, hiA, loB, hiB;
    bounds(0, loA, hiA);
    bounds(stride, loB, hiB);
    float4 qA = at(loA);
    float2 vA = vat(loA);
    PB_TL_STAMP(4);
#pragma unroll 1
    for (int si = 0; si < 10; si += stride) {
      if (si == 4) PB_TL_STAMP(5);
      const OffT lo = loA, end = hiA;
      float4 q0 = qA;
      float2 v0 = vA;
      loA = loB;
      hiA = hiB;
      qA = at(loA);                        // first posrad of the next segment
      vA = vat(loA);
      bounds(si + 2 * stride, loB, hiB);   // bounds of the one after
      if (lo < end) {
#if PB_PREFETCH_DEPTH == 2
        // look-ahead of TWO neighbours (three register sets rotating through a loop unrolled by three); velocities
        // fetched inside the contact block
        OffT off = lo, hoff = lo >> 1;
        const OffT endm16 = end - 16u, endm32 = end > 32u ? end - 32u : 0u, endm48 = end > 48u ? end - 48u : 0u;
        float4 q1 = atI(off, 16);
        for (;;) {
          const float4 q2 = atI(off, 32);
          one(q0, [&]() { return vatI(hoff, 0); }, off != selfOff);
          if (off >= endm16) break;
          q0 = atI(off, 48);
          one(q1, [&]() { return vatI(hoff, 8); }, off != selfOff - 16u);
          if (off >= endm32) break;
          q1 = atI(off, 64);
          one(q2, [&]() { return vatI(hoff, 16); }, off != selfOff - 32u);
          if (off >= endm48) break;
          off += 48u;
          hoff += 24u;
        }
#else
        // two neighbours per turn of the loop: `off` is the even one's byte offset, hoff = off / 2 the
        // offset of its velocity
        OffT off = lo, hoff = lo >> 1;
        const OffT endm = end - 16u;
#if PB_LAZY_VEL
        // experiment: the neighbour's velocity fetched inside the contact block (22 % of the trips on the bench
        // lattice) instead of with every posrad: one vector-memory instruction per trip instead of two
        (void)v0;
        for (;;) {
          const float4 q1 = atI(off, 16);
          one(q0, [&]() { return vatI(hoff, 0); }, off != selfOff);
          if (off >= endm) break;
          off += 32u;
          hoff += 16u;
          q0 = atI(off, 0);
          one(q1, [&]() { return vatI(hoff, -8); }, off != selfOff16);
          if (off >= end) break;
        }
#else
        for (;;) {
          const float4 q1 = atI(off, 16);
          const float2 v1 = vatI(hoff, 8);
          one(q0, [&]() { return v0; }, off != selfOff);
          if (off >= endm) break;
          off += 32u;
          hoff += 16u;
          q0 = atI(off, 0);
          v0 = vatI(hoff, 0);
          one(q1, [&]() { return v1; }, off != selfOff16);
          if (off >= end) break;
        }
#endif
#endif
      }
    }
    if (REPL) rep.flush(F.fr);
    return;
  }
  if (FLAT && NB == 2) {
    // The same sweep with TWO neighbours per trip, evaluated side by side in the same basic blocks
    // (pbPairEvalK<FAST, 2>: two independent dependency chains for the scheduler to interleave) and
    // added in slot order.  The one-per-trip form above leaves ~a third of the SIMD's issue slots
    // empty (a wave's pair evaluation is one long dependent chain and a launch's last waves run
    // nearly alone); this form trades registers (<= 64, still 8 waves per SIMD) for ILP.  A range of
    // odd length evaluates one slot past its end (spare elements; never accumulated).
    const char *const prBytes = (const char *)&prIn[0];
    const char *const velBytes = (const char *)&velIn[0];
    const uint32_t selfOff = s * 16u;
    auto at = [&](uint32_t off) __attribute__((always_inline)) { return *(const float4 *)(prBytes + off); };
    auto vat = [&](uint32_t off) __attribute__((always_inline)) { return *(const float2 *)(velBytes + (off >> 1)); };
    auto two = [&](const float4 &qa, const float2 &va, const float4 &qb, const float2 &vb, uint32_t off,
                   uint32_t end) __attribute__((always_inline)) {
      const bool live[2] = {off != selfOff, (off + 16u != selfOff) && (off + 16u < end)};
      const float bx[2] = {qa.x, qb.x}, by[2] = {qa.y, qb.y}, rb[2] = {qa.z, qb.z};
      const float A[2] = {PAYLOAD ? attraction0 * qa.w * att1 : attraction0,
                          PAYLOAD ? attraction0 * qb.w * att1 : attraction0};
      const float K[2] = {PAYLOAD ? pbBandSlope(A[0]) : slope0, PAYLOAD ? pbBandSlope(A[1]) : slope0};
      PbPairTerm t[2];
      pbPairEvalK<FAST, 2>(CK, live, me.x, me.y, v.x, v.y, me.z, bx, by, rb, A, K,
                           [&](int k) { return k == 0 ? va : vb; }, t);
      pbPairAdd(live[0], t[0], F);
      pbPairAdd(live[1], t[1], F);
    };
    auto bounds = [&](int si, uint32_t &lo, uint32_t &hi) __attribute__((always_inline)) {
      lo = hi = selfOff;
      if (si < 10) {
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) << P.gridXLog2;
        lo = (cellS[row + ((si & 1) ? 0u : mx0)] - base) * 16u;
        hi = (cellS[row + ((si & 1) ? 5u - first : mx0 + first)] - base) * 16u;
      }
    };
    const int stride = nseg == 1 ? 2 : 1;
    uint32_t loA, hiA, loB, hiB;
    bounds(0, loA, hiA);
    bounds(stride, loB, hiB);
    float4 qA = at(loA);
    float2 vA = vat(loA);
#pragma unroll 1
    for (int si = 0; si < 10; si += stride) {
      const uint32_t lo = loA, end = hiA;
      float4 q0 = qA, q1 = at(lo + 16u);
      float2 v0 = vA, v1 = vat(lo + 16u);
      loA = loB;
      hiA = hiB;
      qA = at(loA);  // first posrad of the next segment
      vA = vat(loA);
      bounds(si + 2 * stride, loB, hiB);  // bounds of the one after
      if (lo < end) {
        uint32_t off = lo;
        for (;;) {
          const float4 n0 = at(off + 32u), n1 = at(off + 48u);
          const float2 w0 = vat(off + 32u), w1 = vat(off + 48u);
          two(q0, v0, q1, v1, off, end);
          if ((off += 32u) >= end) break;
          q0 = at(off + 32u);
          q1 = at(off + 48u);
          v0 = vat(off + 32u);
          v1 = vat(off + 48u);
          two(n0, w0, n1, w1, off, end);
          if ((off += 32u) >= end) break;
        }
      }
    }
    return;
  }
  // rolled on purpose: one copy of the pair loop in the binary (unrolling the five rows made ten)
#pragma unroll 1
  for (int si = 0; si < 10; si++) {
    if ((si & 1) && nseg == 1) continue;  // second range of a row only exists at the x-wrap
    const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) << P.gridXLog2;
    const uint32_t lo = cellS[row + ((si & 1) ? 0u : mx0)] - base;
    const uint32_t hi = cellS[row + ((si & 1) ? 5u - first : mx0 + first)] - base;
    if (FLAT) {
      // NB neighbours per trip, evaluated side by side (independent dependency chains for the
      // scheduler to interleave) and then summed in slot order.  The next trip's posrad loads
      // are already in flight (software pipeline).  Out-of-range slots alias the lane's own
      // slot s, which is never accumulated.  With NB > 1 (latency form) the neighbours'
      // velocities travel with their posrad instead of being fetched inside the contact branch.
      constexpr bool PREVEL = NB > 1;
      float4 q[NB];
      float2 vq[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const uint32_t i0 = lo + k < hi ? lo + k : s;
        q[k] = prIn[i0];
        if (PREVEL) vq[k] = velIn[i0];
      }
      for (uint32_t j = lo; j < hi; j += NB) {
        bool live[NB];
        uint32_t idx[NB];
        float bx[NB], by[NB], rb[NB], A[NB], K[NB];
        float2 vb[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
          idx[k] = j + k < hi ? j + k : s;
          live[k] = idx[k] != s;
          bx[k] = q[k].x;
          by[k] = q[k].y;
          rb[k] = q[k].z;
          if (PREVEL) vb[k] = vq[k];
          // payload factors ride in q.w / att1 (impl.cuh:629-633, 640-649)
          A[k] = PAYLOAD ? attraction0 * q[k].w * att1 : attraction0;
          K[k] = PAYLOAD ? pbBandSlope(A[k]) : slope0;
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
          // NB == 1: plain j + 1, no clamp -- one slot past the range is still inside the array
          // (spare element at the end) and is never evaluated
          const uint32_t i1 = (NB == 1 || j + NB + k < hi) ? j + NB + k : s;
          q[k] = prIn[i1];
          if (PREVEL) vq[k] = velIn[i1];
        }
        PbPairTerm t[NB];
        pbPairEvalK<FAST, NB>(CK, live, me.x, me.y, v.x, v.y, me.z, bx, by, rb, A, K,
                              [&](int k) { return PREVEL ? vb[k] : velIn[idx[k]]; }, t);
#pragma unroll
        for (int k = 0; k < NB; k++) pbPairAdd(live[k], t[k], F);
      }
    } else {
      for (uint32_t j = lo; j < hi; j++) {
        const float4 q = prIn[j];
        const float A = PAYLOAD ? P.attraction * q.w * att1 : P.attraction;
        if (j != s) pbPair(P, me.x, me.y, v.x, v.y, me.z, q.x, q.y, q.z, A, [&]() { return velIn[j]; }, F);
      }
    }
  }
}

template <bool PAYLOAD, bool FLAT, bool FAST, int L, int NB, class OffT, bool ASUM = true, class PR, class VL>
__device__ __forceinline__ void pbSweep(const PbDevParams &P, PR prIn, VL velIn, const uint32_t *__restrict__ cellS,
                                        uint32_t base, uint32_t s, uint32_t sub, const float4 &me, const float2 &v,
                                        float att1, PbForce &F, float *repCol = nullptr) {
  PbSegCache none;
  pbSweepC<PAYLOAD, FLAT, FAST, L, NB, false, PR, VL, OffT, ASUM>(P, prIn, velIn, cellS, base, s, sub, me, v, att1, F,
                                                                  none, repCol);
}
