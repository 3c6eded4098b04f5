// pb_engine.hip -- pbSim: the resident, fused particle-robot engine for MI355X (gfx950).
//
// Reference path: Particlebot::update (particlebot.cpp:170-300) and the kernels it launches
// (particlebot_kernel_impl.cuh).  Same arithmetic, different shape:
//
//  * State lives in HBM in CELL-SORTED order between re-sorts (slot s <-> original bot orig[s]).
//    The reference re-gathers pos/vel/rad through a stale permutation every step
//    (reorderDataAndFindCellStartD, impl.cuh:469-538) and memsets a 1 MiB cellStart table; here
//    the gather is the identity and the cell table changes only at a re-sort.
//  * Cell ranges are a DENSE exclusive scan cellS[0..numCells]: cell h owns slots
//    [cellS[h], cellS[h+1]).  With a row-major hash the 5 cells x-2..x+2 of one grid row are one
//    contiguous slot range, so the 25-cell stencil is 5 ranges (10 table reads), visited in the
//    reference's order (row y-2..y+2, then x, then ascending slot) => identical fp32 sums.
//  * One kernel per timestep: k_force<FUSE> computes step n's forces and velocity kick and then,
//    for the same bot, step n+1's radius actuation and integration, reading buffer A and writing
//    buffer B (ping-pong), because neighbours still need the step-n positions.  Per bot it reads
//    posrad 16 + vel 8 + phase 4 + dead 4 + absForce_r 4 B and writes posrad 16 + vel 8 +
//    absForce 8 B; neighbour reads hit L1/L2.
//  * Workgroup -> tile mapping is XCD-aware: the 8 XCDs each walk one contiguous eighth of the
//    sorted array, so a tile's neighbour rows (+-2 grid rows) are in the same XCD's L2.
//  * posrad = (x, y, radius, attraction factor): one 16-byte load per neighbour; the payload's
//    attractionFactor (impl.cuh:629-633,640-644) rides in .w so the pair loop has no index lookups.
//  * A pbSim is a BATCH of nsims >= 1 independent simulations of equal size stepped by the same
//    launches (blockIdx.y = simulation): one parameter block per simulation in device memory, all
//    arrays concatenated (slot = sim*n + local), sort keys = sim*numCells + cell hash, one dense
//    cell table per simulation.  An ensemble of small blobs costs one launch per timestep, not one
//    per simulation; a single arena is the nsims == 1 case.
//  * Three shapes of the same arithmetic, chosen per batch (launchForce / residentWanted), all
//    bit-identical: k_force with L = 1 (one bot per lane: throughput), k_force with L = 2/4/8 lanes
//    per bot (batches that cannot fill the chip), and k_resident (simulations of <= 1024 bots: one
//    workgroup per simulation, state in registers/LDS, many timesteps per launch).
//  * k_force_stream is the one kernel that is NOT bit-identical: the opt-in streamlined arithmetic
//    of force variant 3.
#include <algorithm>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#include "particlebot_hip.h"
#include "pb_device.hpp"
#include "pb_internal.hpp"
#include "pb_xorwow.hpp"

namespace {

thread_local std::string g_lastError;

#define PB_TRY(expr)                                                                                 \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      g_lastError = std::string(hipGetErrorName(e_)) + " at " + __FILE__ + ":" + std::to_string(__LINE__) + \
                    " in " #expr;                                                                    \
      return PB_ERR_HIP;                                                                             \
    }                                                                                                \
  } while (0)

#ifndef PB_TILE
#define PB_TILE 256
#endif
constexpr int TILE = PB_TILE;
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

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

#ifdef PB_TIMELINE
// Diagnostic build only (-DPB_TIMELINE, tools/timeline.py): every workgroup of k_force stamps the
// 100 MHz real-time counter when it starts, when its first wave reaches its first neighbour pair, when
// that wave is half way through its stencil and when the workgroup ends, with the XCD and CU it ran
// on, into a buffer (8 words per workgroup) set by pbDebugSetTimeline.  Never in the shipped library.
__device__ unsigned long long *pbTimelineBuf = nullptr;
#define PB_TL_STAMP(word)                                                                              \
  do {                                                                                                 \
    if (pbTimelineBuf && threadIdx.x == 0)                                                             \
      pbTimelineBuf[8ull * (blockIdx.y * gridDim.x + blockIdx.x) + (word)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define PB_TL_STAMP(word) do { } while (0)
#endif

// ---- kernels ------------------------------------------------------------------------------

// stand-alone radius actuation + integration for one step (impl.cuh:124-181 + :53-103), in place
__global__ __launch_bounds__(TILE) void k_state(const PbDevParams *__restrict__ params, float4 *__restrict__ pr,
                                                float2 *__restrict__ vel, const float *__restrict__ phase,
                                                const int *__restrict__ dead, const float *__restrict__ absA,
                                                const float *__restrict__ absR, uint32_t n, float time, float dt,
                                                int doRadius) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t s = blockIdx.y * n + l;
  float4 q = pr[s];
  float2 v = vel[s];
  if (doRadius) q.z = pbActuate(P, q.z, phase[s], dead[s], absA[s], absR[s], time, dt);
  pbIntegrate(P, q.x, q.y, v.x, v.y, q.z, dt);
  pr[s] = q;
  vel[s] = v;
}

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
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) * GX;
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
              CK, live[0], me.x, me.y, v.x, v.y, me.z, q.x, q.y, q.z, w.x, w.y, A[0], K[0], [&](bool mine, float m2) {
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
    PbRepList<FAST, PB_REP_CAP, REPSTRIDE> rep;
    if (!ASUM) rep.init(repCol);
    // (64-bit address arithmetic with a constant displacement: the displacement becomes the load's
    //  immediate offset, so the look-ahead loads need no address instructions of their own)
    auto atI = [&](OffT off, int imm) __attribute__((always_inline)) {
      return *(const float4 *)(prBytes + (uint64_t)off + imm);
    };
    auto vatI = [&](OffT hoff, int imm) __attribute__((always_inline)) {
      return *(const float2 *)(velBytes + (uint64_t)hoff + imm);
    };
    const OffT selfOff16 = selfOff + 16u;
    auto one = [&](const float4 &q, const float2 &vq, bool isLive) __attribute__((always_inline)) {
      const bool live[1] = {isLive};
      const float bx[1] = {q.x}, by[1] = {q.y}, rb[1] = {q.z};
      const float A[1] = {PAYLOAD ? attraction0 * q.w * att1 : attraction0};
      const float K[1] = {PAYLOAD ? pbBandSlope(A[0]) : slope0};
      if (ASUM) {
        PbPairTerm t[1];
        pbPairEvalK<FAST, 1>(CK, live, me.x, me.y, v.x, v.y, me.z, bx, by, rb, A, K, [&](int) { return vq; }, t);
        pbPairAdd(live[0], t[0], F);
      } else {
        const PbPairXY t = pbPairEvalXY<FAST>(CK, live[0], me.x, me.y, v.x, v.y, me.z, q.x, q.y, q.z, vq.x, vq.y, A[0],
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
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) * GX;
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
        // look-ahead of TWO neighbours (three register sets rotating through a loop unrolled by three)
        OffT off = lo, hoff = lo >> 1;
        const OffT endm16 = end - 16u, endm32 = end > 32u ? end - 32u : 0u, endm48 = end > 48u ? end - 48u : 0u;
        float4 q1 = atI(off, 16);
        float2 v1 = vatI(hoff, 8);
        for (;;) {
          const float4 q2 = atI(off, 32);
          const float2 v2 = vatI(hoff, 16);
          one(q0, v0, off != selfOff);
          if (off >= endm16) break;
          q0 = atI(off, 48);
          v0 = vatI(hoff, 24);
          one(q1, v1, off != selfOff - 16u);
          if (off >= endm32) break;
          q1 = atI(off, 64);
          v1 = vatI(hoff, 32);
          one(q2, v2, off != selfOff - 32u);
          if (off >= endm48) break;
          off += 48u;
          hoff += 24u;
        }
#else
        // two neighbours per turn of the loop: `off` is the even one's byte offset, hoff = off / 2 the
        // offset of its velocity
        OffT off = lo, hoff = lo >> 1;
        const OffT endm = end - 16u;
        for (;;) {
          const float4 q1 = atI(off, 16);
          const float2 v1 = vatI(hoff, 8);
          one(q0, v0, off != selfOff);
          if (off >= endm) break;
          off += 32u;
          hoff += 16u;
          q0 = atI(off, 0);
          v0 = vatI(hoff, 0);
          one(q1, v1, off != selfOff16);
          if (off >= end) break;
        }
#endif
      }
    }
    if (!ASUM) rep.flush(F.fr);
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
        const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) * GX;
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
    const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) * GX;
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

// Forces + kick of step n (impl.cuh:657-831); with FUSE also radius + integration of step n+1.
// PAYLOAD: object-transport mode (nDead == -1), per-pair attraction factors.  FLAT: branch-free
// pair evaluation instead of the reference-shaped branches (pbPair).
// FASTOK: the simulation passed pbFastMathAllowed, so waves whose lanes all pass
// pbLaneFastMathOk may use the exact fast sqrt/division forms.
// BIG: 64-bit byte offsets in the neighbour sweep (batches of 2^28 bots and more, throughput form only).
// ASUM: maintain absForce_a.  false (throughput form, batches without constrained contraction):
// the attraction magnitudes are dead values and are neither computed nor stored (pbPairEvalXY).
template <bool FUSE, bool PAYLOAD, bool FLAT, bool FASTOK, int L, int NB, bool BIG = false, bool ASUM = true>
__global__ __launch_bounds__(TILE, (NB == 2 ? PB_NB2_WAVES : PB_FORCE_WAVES)) void k_force(const PbDevParams *__restrict__ params,
                                                const float4 *__restrict__ prIn, const float2 *__restrict__ velIn,
                                                float4 *__restrict__ prOut, float2 *__restrict__ velOut,
                                                const float *__restrict__ phase, const int *__restrict__ dead,
                                                float *__restrict__ absA, float *__restrict__ absR,
                                                const uint32_t *__restrict__ orig,
                                                const uint32_t *__restrict__ cellSAll, uint32_t n, float dt,
                                                float timeNext, int doRadiusNext, uint32_t perXcd) {
  const PbDevParams &P = params[blockIdx.y];
  // XCD-aware tile order (large simulations): workgroups b, b+8, b+16, ... share an XCD
  // (round-robin dispatch); give each XCD one contiguous eighth of the tiles (gridDim.x = 8*perXcd).
  // perXcd == 0: plain order (small simulations, a handful of tiles each).
  const uint32_t tile = perXcd ? (blockIdx.x & 7u) * perXcd + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t l = tile * (TILE / L) + threadIdx.x / L;  // all L lanes of a group share the bot
  const uint32_t sub = threadIdx.x % L;
#ifdef PB_TIMELINE
  // (stored at once: a start stamp kept in registers to the end cost the kernel a wave per SIMD)
  PB_TL_STAMP(0);
#endif
  if (l >= n) return;
  const uint32_t s = blockIdx.y * n + l;  // global slot; the cell table holds global slots too
  const uint32_t *__restrict__ cellS = cellSAll + (size_t)blockIdx.y * (P.numCells + 1u);

  const float4 me = prIn[s];
  float2 v = velIn[s];
  bool selfPayload = false;
  if (PAYLOAD) selfPayload = (orig[s] == P.nCells - 1u);
  const float att1 = selfPayload ? P.attractionFactor : 1.0f;

  PbForce F;
  F.fx = 0.0f;
  F.fy = 0.0f;
  F.fa = 0.0f;
  F.fr = 0.0f * absR[s];  // impl.cuh:688

  // wave-uniform choice: the fast exact forms need every lane's coordinates away from zero
  using OffT = typename std::conditional<BIG, uint64_t, uint32_t>::type;
  static_assert(ASUM || (FLAT && NB == 1), "the dead-sum form exists for the branch-free sweeps only");
  constexpr bool REPLIST = !ASUM && L == 1;  // (L > 1: magnitudes are rooted inside the contact block)
  __shared__ float repLds[REPLIST ? (PB_REP_CAP + 1) * TILE : 1];
  float *const repCol = &repLds[REPLIST ? threadIdx.x : 0];
  if (FLAT && FASTOK && __all(pbLaneFastMathOk(me.x, me.y)))
    pbSweep<PAYLOAD, FLAT, true, L, NB, OffT, ASUM>(P, prIn, velIn, cellS, 0u, s, sub, me, v, att1, F, repCol);
  else
    pbSweep<PAYLOAD, FLAT, false, L, NB, OffT, ASUM>(P, prIn, velIn, cellS, 0u, s, sub, me, v, att1, F, repCol);
  pbObstacles(P, me.x, me.y, v.x, v.y, me.z, F);
  pbFrictionAndKick(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);

  float4 out = me;
  if (FUSE) {
    if (doRadiusNext) out.z = pbActuate(P, me.z, phase[s], dead[s], F.fa, F.fr, timeNext, dt);
    pbIntegrate(P, out.x, out.y, v.x, v.y, out.z, dt);
  }
  if (sub == 0) {  // the L lanes of a group hold identical results
    prOut[s] = out;
    velOut[s] = v;
    if (ASUM) absA[s] = F.fa;
    absR[s] = F.fr;
  }
#ifdef PB_TIMELINE
  if (pbTimelineBuf && threadIdx.x == 0) {
    unsigned long long *row = pbTimelineBuf + 8ull * (blockIdx.y * gridDim.x + blockIdx.x);
    row[1] = __builtin_amdgcn_s_memrealtime();
    row[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, bits 0..3
    row[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID (wave/simd/cu/sh/se)
  }
#endif
}

// Streamlined force kernel (force variant 3): same inputs, outputs and fusion as k_force, pair
// arithmetic from pbGeomS/pbFarCoefS/pbContactS.  Results are NOT bit-identical to the reference
// restatement; they stay within 1e-5 relative of it over teacher-forced windows (DESIGN.md
// "Streamlined").  Two passes per bot:
//   1. all candidates: distance, unit vector, attraction coefficient; accumulate force and Sum|F_attr|;
//      a candidate in contact only has its slot pushed onto the lane's list in LDS
//   2. the lane's contacts (a handful): spring/dashpot/shear, |F|, accumulate force and Sum|F_rep|
// so the contact arithmetic runs for ~8 trips per bot instead of for every trip in which ANY lane of
// the wave is in contact (nearly all 50 in a dense blob).  One bot per lane; throughput form only.
#ifndef PB_STREAM_CAP
#define PB_STREAM_CAP 12
#endif
#ifndef PB_STREAM_PAIRS
// 1 = two candidates per loop trip sharing the near-band ballots and the contact push (build-time
// experiment, VERDICT r1 item 7: halve the scalar/branch instructions).  Measured on MI355X at 10^6
// bots: 59.4 us/step (74 VGPRs, 6 waves/SIMD) against 55.6 for the one-per-trip loop below (59 VGPRs):
// the scalar instructions were not what held the kernel back; the one-per-trip loop ships.
#define PB_STREAM_PAIRS 0
#endif
template <bool FUSE, bool PAYLOAD>
__global__ __launch_bounds__(TILE) void k_force_stream(const PbDevParams *__restrict__ params,
                                                       const float4 *__restrict__ prIn,
                                                       const float2 *__restrict__ velIn, float4 *__restrict__ prOut,
                                                       float2 *__restrict__ velOut, const float *__restrict__ phase,
                                                       const int *__restrict__ dead, float *__restrict__ absA,
                                                       float *__restrict__ absR, const uint32_t *__restrict__ orig,
                                                       const uint32_t *__restrict__ cellSAll, uint32_t n, float dt,
                                                       float timeNext, int doRadiusNext, uint32_t perXcd) {
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
  const PbContactK CK{P.spring, P.damping, P.shear};
  const float near2 = 0.0019f;

  float fx = 0.0f, fy = 0.0f, fa = 0.0f;
  float fr = 0.0f * absR[s];  // impl.cuh:688
  uint32_t cnt = 0;

  auto contactOf = [&](uint32_t j, const float4 &q) __attribute__((always_inline)) {
    const float rx = q.x - me.x, ry = q.y - me.y;
    const PbGeomS g = pbGeomS(rx, ry, fmaxf(__builtin_fmaf(rx, rx, ry * ry), 1e-30f));
    const float2 vb = velIn[j];
    float cx, cy;
    const float mag = pbContactS(CK, g, me.z + q.z, vb.x - v.x, vb.y - v.y, cx, cy);
    fx += cx;
    fy += cy;
    fr += mag;
  };
  // No test for the bot's own slot: with d2 clamped away from zero the self pair has n = 0 and
  // gap = -reach, so it lands on the contact list, where it evaluates to a zero force (n = 0,
  // relative velocity 0).  (Two distinct bots at the same point, NaN in the reference, also give 0.)
  auto one = [&](const float4 &q, uint32_t off) __attribute__((always_inline)) {
    const float rx = q.x - me.x, ry = q.y - me.y;
    const float d2 = fmaxf(__builtin_fmaf(rx, rx, ry * ry), 1e-30f);
    const float inv = __builtin_amdgcn_rsqf(d2);                   // 1/dist
    const float gap = __builtin_fmaf(d2, inv, -(me.z + q.z));      // dist - reach
    const bool contact = gap < 0.0f;                               // dist < reach
    const float A = PAYLOAD ? attraction0 * q.w * att1 : attraction0;
    float coef = pbFarCoefS(A, gap);
    // the two near bands are rare: wave-uniform branch on ballots of the plain comparisons
    const unsigned long long mNear =
        __builtin_amdgcn_ballot_w64(gap < near2) & ~__builtin_amdgcn_ballot_w64(contact);
    if (mNear != 0ull) coef = gap < near2 ? pbBandCoefS(A, gap) : coef;
    coef = contact ? 0.0f : coef;
    const float ci = coef * inv;  // term = coef * n = (coef / dist) * r
    fx = __builtin_fmaf(ci, rx, fx);
    fy = __builtin_fmaf(ci, ry, fy);
    fa += coef;
    if (contact) {
      if (cnt < (uint32_t)PB_STREAM_CAP) contacts[cnt][threadIdx.x] = off >> 4;
      else contactOf(off >> 4, q);  // list full (pathological compression): evaluate in place
      cnt++;
    }
  };

#if PB_STREAM_PAIRS
  // the same for two candidates (slots off and off + 16 bytes) side by side
  auto two = [&](const float4 &qa, const float4 &qb, uint32_t off) __attribute__((always_inline)) {
    const float rxa = qa.x - me.x, rya = qa.y - me.y, rxb = qb.x - me.x, ryb = qb.y - me.y;
    const float d2a = fmaxf(__builtin_fmaf(rxa, rxa, rya * rya), 1e-30f);
    const float d2b = fmaxf(__builtin_fmaf(rxb, rxb, ryb * ryb), 1e-30f);
    const float inva = __builtin_amdgcn_rsqf(d2a), invb = __builtin_amdgcn_rsqf(d2b);
    const float gapa = __builtin_fmaf(d2a, inva, -(me.z + qa.z)), gapb = __builtin_fmaf(d2b, invb, -(me.z + qb.z));
    const bool ca = gapa < 0.0f, cb = gapb < 0.0f;
    const float Aa = PAYLOAD ? attraction0 * qa.w * att1 : attraction0;
    const float Ab = PAYLOAD ? attraction0 * qb.w * att1 : attraction0;
    float coa = pbFarCoefS(Aa, gapa), cob = pbFarCoefS(Ab, gapb);
    const unsigned long long mNear =
        (__builtin_amdgcn_ballot_w64(gapa < near2) & ~__builtin_amdgcn_ballot_w64(ca)) |
        (__builtin_amdgcn_ballot_w64(gapb < near2) & ~__builtin_amdgcn_ballot_w64(cb));
    if (mNear != 0ull) {
      coa = gapa < near2 ? pbBandCoefS(Aa, gapa) : coa;
      cob = gapb < near2 ? pbBandCoefS(Ab, gapb) : cob;
    }
    coa = ca ? 0.0f : coa;
    cob = cb ? 0.0f : cob;
    const float cia = coa * inva, cib = cob * invb;
    fx = __builtin_fmaf(cia, rxa, fx);
    fy = __builtin_fmaf(cia, rya, fy);
    fa += coa;
    fx = __builtin_fmaf(cib, rxb, fx);
    fy = __builtin_fmaf(cib, ryb, fy);
    fa += cob;
    if (ca || cb) {
      if (ca) {
        if (cnt < (uint32_t)PB_STREAM_CAP) contacts[cnt][threadIdx.x] = off >> 4;
        else contactOf(off >> 4, qa);
        cnt++;
      }
      if (cb) {
        if (cnt < (uint32_t)PB_STREAM_CAP) contacts[cnt][threadIdx.x] = (off >> 4) + 1u;
        else contactOf((off >> 4) + 1u, qb);
        cnt++;
      }
    }
  };
#endif

  const int gx = pbCellX(P, me.x), gy = pbCellY(P, me.y);
  const uint32_t GX = P.gridX;
  const uint32_t mx0 = (uint32_t)(gx - 2) & (GX - 1u);
  const uint32_t first = (GX - mx0) < 5u ? (GX - mx0) : 5u;
  const int nseg = first < 5u ? 2 : 1;
  // Segment loop rolled and software-pipelined two deep, as in pbSweep: while segment si runs, the
  // cell-table bounds of segment si + 2 and the first two posrad of segment si + 1 are in flight.
  // Inside a segment posrad loads run two neighbours ahead, three registers rotating roles; the
  // loop runs on 32-bit byte offsets.  Up to two slots past a range are read (spare elements at
  // the end of the array), never evaluated.
  const char *const prBytes = (const char *)prIn;
  auto at = [&](uint32_t off) __attribute__((always_inline)) { return *(const float4 *)(prBytes + off); };
  const uint32_t selfOff = s * 16u;
  auto bounds = [&](int si, uint32_t &lo, uint32_t &hi) __attribute__((always_inline)) {
    lo = hi = selfOff;
    if (si < 10) {
      const uint32_t row = ((uint32_t)(gy + (si >> 1) - 2) & (P.gridY - 1u)) * GX;
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
#if PB_STREAM_PAIRS
    // Two candidates per trip: their near-band tests share one pair of ballots and one wave-uniform
    // branch, their contact pushes one exec-masked block (the one-per-trip form below spends one scalar
    // or branch instruction per two vector ones on exactly these), and the scheduler gets two
    // independent rsq/rcp chains.  An odd candidate at the end of a range is handled alone.  Posrad loads
    // run one pair ahead; up to three slots past a range are read (spare elements), never evaluated.
    if (lo < end) {
      uint32_t off = lo;
      for (;;) {
        if (off + 16u >= end) {  // one candidate left in this lane's range
          one(q0, off);
          break;
        }
        const float4 n0 = at(off + 32u), n1 = at(off + 48u);
        two(q0, q1, off);
        if ((off += 32u) >= end) break;
        if (off + 16u >= end) {
          one(n0, off);
          break;
        }
        q0 = at(off + 32u);
        q1 = at(off + 48u);
        two(n0, n1, off);
        if ((off += 32u) >= end) break;
      }
    }
#else
    if (lo < end) {
      uint32_t off = lo;
      for (;;) {
        const float4 q2 = at(off + 32u);
        one(q0, off);
        if ((off += 16u) >= end) break;
        q0 = at(off + 32u);
        one(q1, off);
        if ((off += 16u) >= end) break;
        q1 = at(off + 32u);
        one(q2, off);
        if ((off += 16u) >= end) break;
      }
    }
#endif
  }
  const uint32_t listed = cnt < (uint32_t)PB_STREAM_CAP ? cnt : (uint32_t)PB_STREAM_CAP;
  for (uint32_t k = 0; k < listed; k++) {
    const uint32_t j = contacts[k][threadIdx.x];
    contactOf(j, prIn[j]);
  }

  PbForce F{fx, fy, fa, fr};
  pbObstacles(P, me.x, me.y, v.x, v.y, me.z, F);
  pbFrictionAndKick(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);
  float4 out = me;
  if (FUSE) {
    if (doRadiusNext) out.z = pbActuate(P, me.z, phase[s], dead[s], F.fa, F.fr, timeNext, dt);
    pbIntegrate(P, out.x, out.y, v.x, v.y, out.z, dt);
  }
  prOut[s] = out;
  velOut[s] = v;
  absA[s] = F.fa;
  absR[s] = F.fr;
}

// Resident form for small simulations: ONE workgroup per simulation keeps its bots in registers
// (L lanes per bot) and the positions/velocities the neighbours read in LDS (ping-pong), and runs
// nsteps whole timesteps in one launch with one workgroup barrier per step.  A per-step launch of
// a few hundred bots spends ~12 us in dependent HBM round trips (kernel arguments -> own state ->
// cell table -> neighbours); here those become LDS reads.  The host launches it for the stretch of
// steps up to the next re-sort / phase update / caller boundary (stepMany).  Same device functions,
// same order of operations as k_state + k_force: bit-identical results.
template <bool PAYLOAD, bool FASTOK, int L, bool ASUM = true>
__global__ __launch_bounds__(1024) void k_resident(const PbDevParams *__restrict__ params, float4 *__restrict__ pr,
                                                   float2 *__restrict__ vel, const float *__restrict__ phase,
                                                   const int *__restrict__ dead, float *__restrict__ absA,
                                                   float *__restrict__ absR, const uint32_t *__restrict__ orig,
                                                   const uint32_t *__restrict__ cellSAll, uint32_t n, float dt,
                                                   float time0, int nsteps, int lightWave) {
  constexpr int CAP = 1024 / L;
  __shared__ float4 sPr[2][CAP + 1];  // +1: the sweep prefetches one slot past a range
  __shared__ float2 sVel[2][CAP + 1];
  constexpr bool REPLIST = !ASUM && L == 1;
  __shared__ float repLds[REPLIST ? (PB_REP_CAP + 1) * 1024 : 1];
  float *const repCol = &repLds[REPLIST ? threadIdx.x : 0];
  const PbDevParams &P = params[blockIdx.x];
  const uint32_t l = threadIdx.x / L, sub = threadIdx.x % L;
  const bool active = l < n;
  const uint32_t base = blockIdx.x * n;
  const uint32_t s = base + (active ? l : 0u);
  const uint32_t *__restrict__ cellS = cellSAll + (size_t)blockIdx.x * (P.numCells + 1u);

  float4 me = pr[s];
  float2 v = vel[s];
  const float ph = phase[s];
  const int dd = dead[s];
  float fa = absA[s], fr = absR[s];
  bool selfPayload = false;
  if (PAYLOAD) selfPayload = (orig[s] == P.nCells - 1u);
  const float att1 = selfPayload ? P.attractionFactor : 1.0f;

  PbSegCache segCache;
  segCache.gx = segCache.gy = (int)0x80000000;  // no cell yet
  segCache.m = 0;
  float t = time0;
  // radius actuation + integration of the first step (k_state)
  if (lightWave && t >= 0) me.z = pbActuate(P, me.z, ph, dd, fa, fr, t, dt);
  pbIntegrate(P, me.x, me.y, v.x, v.y, me.z, dt);
  if (active && sub == 0) {
    sPr[0][l] = me;
    sVel[0][l] = v;
  }
  __syncthreads();
  int cur = 0;
  for (int k = 0; k < nsteps; k++) {
    const float tNext = t + dt;
    if (active) {
      PbForce F;
      F.fx = 0.0f;
      F.fy = 0.0f;
      F.fa = 0.0f;
      F.fr = 0.0f * fr;  // impl.cuh:688
      const float4 *prIn = sPr[cur];
      const float2 *velIn = sVel[cur];
      using PR = const float4 *;
      using VL = const float2 *;
      if (FASTOK && __all(pbLaneFastMathOk(me.x, me.y)))
        pbSweepC<PAYLOAD, true, true, L, 1, (L > 1), PR, VL, uint32_t, ASUM, 1024>(P, prIn, velIn, cellS, base, l, sub, me,
                                                                                 v, att1, F, segCache, repCol);
      else
        pbSweepC<PAYLOAD, true, false, L, 1, (L > 1), PR, VL, uint32_t, ASUM, 1024>(P, prIn, velIn, cellS, base, l, sub,
                                                                                  me, v, att1, F, segCache, repCol);
      pbObstacles(P, me.x, me.y, v.x, v.y, me.z, F);
      pbFrictionAndKick(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);
      fa = F.fa;
      fr = F.fr;
      if (k + 1 < nsteps) {  // the next step's radius actuation + integration
        if (lightWave && tNext >= 0) me.z = pbActuate(P, me.z, ph, dd, fa, fr, tNext, dt);
        pbIntegrate(P, me.x, me.y, v.x, v.y, me.z, dt);
        if (sub == 0) {
          sPr[cur ^ 1][l] = me;
          sVel[cur ^ 1][l] = v;
        }
      }
    }
    t = tNext;
    cur ^= 1;
    __syncthreads();
  }
  if (active && sub == 0) {
    pr[s] = me;
    vel[s] = v;
    if (ASUM) absA[s] = fa;
    absR[s] = fr;
  }
}

// re-sort step 1: hash in ORIGINAL order (calcHashD, impl.cuh:446-465) + inverse permutation.
// Keys carry the simulation number above the cell hash so one stable sort keeps simulations apart.
__global__ __launch_bounds__(TILE) void k_hash(const PbDevParams *__restrict__ params, const float4 *__restrict__ pr,
                                               const uint32_t *__restrict__ orig, uint32_t *__restrict__ keys,
                                               uint32_t *__restrict__ vals, uint32_t *__restrict__ slotOf, uint32_t n) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t base = blockIdx.y * n, s = base + l;
  const float4 q = pr[s];
  const uint32_t o = base + orig[s];
  keys[o] = blockIdx.y * P.numCells + pbHash(P, pbCellX(P, q.x), pbCellY(P, q.y));
  vals[o] = o;
  slotOf[o] = s;
}

// re-sort step 3: move every per-bot array into the new slot order
__global__ __launch_bounds__(TILE) void k_permute(const uint32_t *__restrict__ newOrig,
                                                  const uint32_t *__restrict__ slotOf, const float4 *__restrict__ prIn,
                                                  const float2 *__restrict__ velIn, const float *__restrict__ phaseIn,
                                                  const int *__restrict__ deadIn, const float *__restrict__ absAIn,
                                                  const float *__restrict__ absRIn, float4 *__restrict__ prOut,
                                                  float2 *__restrict__ velOut, float *__restrict__ phaseOut,
                                                  int *__restrict__ deadOut, float *__restrict__ absAOut,
                                                  float *__restrict__ absROut, uint32_t *__restrict__ origOut,
                                                  uint32_t n) {
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t base = blockIdx.y * n, t = base + l;
  const uint32_t o = newOrig[t];  // global original index; stays inside this simulation's block
  const uint32_t src = slotOf[o];
  prOut[t] = prIn[src];
  velOut[t] = velIn[src];
  phaseOut[t] = phaseIn[src];
  deadOut[t] = deadIn[src];
  absAOut[t] = absAIn[src];
  absROut[t] = absRIn[src];
  origOut[t] = o - base;
}

// re-sort step 4: cellS[sim][c] = number of bots (all simulations before + this one) whose key is
// below sim*numCells + c: a lower bound in the sorted keys, as a GLOBAL slot index
__global__ __launch_bounds__(TILE) void k_cell_scan(const uint32_t *__restrict__ sortedKeys, uint32_t total,
                                                    uint32_t *__restrict__ cellS, uint32_t numCells) {
  const uint32_t c = blockIdx.x * TILE + threadIdx.x;
  if (c > numCells) return;
  const uint32_t want = blockIdx.y * numCells + c;
  uint32_t lo = 0, hi = total;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sortedKeys[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  cellS[(size_t)blockIdx.y * (numCells + 1u) + c] = lo;
}

// min over a simulation's bots of the squared distance to the light, as the host loop of
// particlebot.cpp:215-228 squares it: powf(light-x,2)+powf(light-y,2).  Non-negative floats order
// like their bit patterns, so an integer min is exact and order-independent.  Two levels: one value
// per workgroup (wave shuffles, then LDS), then one wave per simulation over those (k_min_final) --
// thousands of atomicMax/Min on one address took 46 us at 10^6 bots, this takes ~10.
//
// NaN semantics of the reference loop (a simulation that has blown up): `min_d = (min_d < dist ? min_d :
// dist)` in ORIGINAL bot order takes `dist` whenever the comparison is false, so a NaN distance
// replaces the running minimum and the next bot's distance replaces the NaN: the loop's result is the
// minimum over the bots AFTER the last NaN one (NaN itself if that is the last bot).  k_last_nan finds
// that index per simulation (-1: none), k_min_dist2 then only admits bots with a larger original
// index; both are two-level reductions.
__global__ __launch_bounds__(TILE) void k_last_nan(const PbDevParams *__restrict__ params,
                                                   const float4 *__restrict__ pr, const uint32_t *__restrict__ orig,
                                                   uint32_t n, int *__restrict__ partial) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  int last = -1;
  if (l < n) {
    const uint32_t s = blockIdx.y * n + l;
    const float4 q = pr[s];
    const float dx = P.light_x - q.x, dy = P.light_y - q.y;
    const float d2 = dx * dx + dy * dy;
    if (d2 != d2) last = (int)orig[s];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(last, d, 64);
    last = o > last ? o : last;
  }
  __shared__ int waveMax[TILE / 64];
  if ((threadIdx.x & 63u) == 0u) waveMax[threadIdx.x >> 6] = last;
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = waveMax[0];
#pragma unroll
    for (int w = 1; w < TILE / 64; w++) m = waveMax[w] > m ? waveMax[w] : m;
    partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
  }
}

__global__ __launch_bounds__(64) void k_last_nan_final(const int *__restrict__ partial, uint32_t nb,
                                                       int *__restrict__ out) {
  int last = -1;
  for (uint32_t b = threadIdx.x; b < nb; b += 64u) {
    const int o = partial[blockIdx.x * nb + b];
    last = o > last ? o : last;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(last, d, 64);
    last = o > last ? o : last;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = last;
}

__global__ __launch_bounds__(TILE) void k_min_dist2(const PbDevParams *__restrict__ params,
                                                    const float4 *__restrict__ pr, const uint32_t *__restrict__ orig,
                                                    const int *__restrict__ lastNan, uint32_t n,
                                                    uint32_t *__restrict__ partial) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  uint32_t bits = 0x7f800000u;  // +inf
  if (l < n) {
    const uint32_t s = blockIdx.y * n + l;
    const float4 q = pr[s];
    const float dx = P.light_x - q.x, dy = P.light_y - q.y;
    if ((int)orig[s] > lastNan[blockIdx.y]) bits = __float_as_uint(dx * dx + dy * dy);  // (never NaN: those are <= lastNan)
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(bits, d, 64);
    bits = o < bits ? o : bits;
  }
  __shared__ uint32_t waveMin[TILE / 64];
  if ((threadIdx.x & 63u) == 0u) waveMin[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = waveMin[0];
#pragma unroll
    for (int w = 1; w < TILE / 64; w++) m = waveMin[w] < m ? waveMin[w] : m;
    partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
  }
}

__global__ __launch_bounds__(64) void k_min_final(const uint32_t *__restrict__ partial, uint32_t nb,
                                                  uint32_t *__restrict__ outBits) {
  uint32_t bits = 0x7f800000u;
  for (uint32_t b = threadIdx.x; b < nb; b += 64u) {
    const uint32_t o = partial[blockIdx.x * nb + b];
    bits = o < bits ? o : bits;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(bits, d, 64);
    bits = o < bits ? o : bits;
  }
  if (threadIdx.x == 0) outBits[blockIdx.x] = bits;
}

// updatePhase (impl.cuh:264-290) + add_normal_noise (impl.cuh:43-51) in slot order.  rng: per-bot
// XORWOW states in ORIGINAL order (nullptr: the counter generator, which needs none).
__global__ __launch_bounds__(TILE) void k_phase(const PbDevParams *__restrict__ params, const float4 *__restrict__ pr,
                                                const uint32_t *__restrict__ orig, float *__restrict__ phase,
                                                uint32_t n, const float *__restrict__ minD, uint32_t draw,
                                                pbRngState *__restrict__ rng) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t s = blockIdx.y * n + l;
  const float4 q = pr[s];
  const float spacing = 2.0f * P.min_radius;  // particlebot.cpp:229
  float ph = pbPhase(P, q.x, q.y, spacing, minD[blockIdx.y], phase[s]);
  if (P.phase_std != 0.0f) {
    float z;
    if (rng) {
      pbRngState st = rng[blockIdx.y * n + orig[s]];
      z = pbXorwowNormal(st);
      rng[blockIdx.y * n + orig[s]] = st;
    } else {
      z = pbNormal(P.seed, orig[s], draw);
    }
    const float noise = P.phase_std * z;
    ph += noise;
  }
  phase[s] = ph;
}

// curand_setup_kernel (impl.cuh:36-41): state[i] = curand_init(seed of the simulation, i, 0), then
// `draws` normals already consumed (checkpoint resume)
__global__ __launch_bounds__(TILE) void k_rng_init(const PbDevParams *__restrict__ params,
                                                   pbRngState *__restrict__ rng, uint32_t n, int kind,
                                                   const uint32_t *__restrict__ jump, uint32_t draws) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t i = blockIdx.x * TILE + threadIdx.x;
  if (i >= n) return;
  pbRngState st;
  pbXorwowSeed(st, (uint64_t)P.seed, kind);
  pbXorwowSkipSubsequences(st, i, jump);
  if (P.phase_std != 0.0f)
    for (uint32_t k = 0; k < draws; k++) (void)pbXorwowNormal(st);
  rng[blockIdx.y * n + i] = st;
}

// host arrays of ONE simulation (original order, staged on the device) -> slot order.  The staged
// arrays hold the bots [start, start + count) of the original order; other bots keep their state.
__global__ __launch_bounds__(TILE) void k_set_state(const PbDevParams *__restrict__ params, uint32_t sim,
                                                    const uint32_t *__restrict__ orig, float4 *__restrict__ pr,
                                                    float2 *__restrict__ vel, float *__restrict__ phase,
                                                    int *__restrict__ dead, const float2 *__restrict__ inPos,
                                                    const float2 *__restrict__ inVel, const float *__restrict__ inRad,
                                                    const float *__restrict__ inPhase, const int *__restrict__ inDead,
                                                    uint32_t n, uint32_t start, uint32_t count) {
  const PbDevParams &P = params[sim];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t s = sim * n + l;
  const uint32_t o = orig[s];
  const uint32_t k = o - start;  // index into the staged arrays
  const bool mine = k < count;
  float4 q = pr[s];
  if (inPos && mine) {
    const float2 p = inPos[k];
    q.x = p.x;
    q.y = p.y;
  }
  if (inRad && mine) q.z = inRad[k];
  q.w = (P.nDead == -1 && o == P.nCells - 1u) ? P.attractionFactor : 1.0f;
  pr[s] = q;
  if (inVel && mine) vel[s] = inVel[k];
  if (inPhase && mine) phase[s] = inPhase[k];
  if (inDead && mine) dead[s] = inDead[k];
}

// slot order -> original order, ONE simulation
__global__ __launch_bounds__(TILE) void k_get_state(uint32_t sim, const uint32_t *__restrict__ orig,
                                                    const float4 *__restrict__ pr, const float2 *__restrict__ vel,
                                                    const float *__restrict__ phase, const int *__restrict__ dead,
                                                    const float *__restrict__ absA, const float *__restrict__ absR,
                                                    float2 *__restrict__ outPos, float2 *__restrict__ outVel,
                                                    float *__restrict__ outRad, float *__restrict__ outPhase,
                                                    int *__restrict__ outDead, float *__restrict__ outAbsA,
                                                    float *__restrict__ outAbsR, uint32_t n) {
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t s = sim * n + l;
  const uint32_t o = orig[s];
  const float4 q = pr[s];
  outPos[o] = make_float2(q.x, q.y);
  outRad[o] = q.z;
  outVel[o] = vel[s];
  outPhase[o] = phase[s];
  outDead[o] = dead[s];
  outAbsA[o] = absA[s];
  outAbsR[o] = absR[s];
}

// absForce_a / absForce_r of ONE simulation, original order -> slot order (checkpoint restore)
__global__ __launch_bounds__(TILE) void k_set_forces(uint32_t sim, const uint32_t *__restrict__ orig,
                                                     float *__restrict__ absA, float *__restrict__ absR,
                                                     const float *__restrict__ inA, const float *__restrict__ inR,
                                                     uint32_t n) {
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t s = sim * n + l;
  const uint32_t o = orig[s];
  absA[s] = inA[o];
  absR[s] = inR[o];
}

__global__ __launch_bounds__(TILE) void k_iota(uint32_t *__restrict__ a, uint32_t n) {
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l < n) a[blockIdx.y * n + l] = l;
}

// Centre of mass of every simulation in ORIGINAL index order with a fixed summation tree:
// positions scattered to original order, per-workgroup partial sums of 256 consecutive bots
// (double), then one wave per simulation adds its partials in order.
__global__ __launch_bounds__(TILE) void k_com_scatter(const uint32_t *__restrict__ orig,
                                                      const float4 *__restrict__ pr, float2 *__restrict__ posOrig,
                                                      uint32_t n) {
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  if (l >= n) return;
  const uint32_t s = blockIdx.y * n + l;
  const float4 q = pr[s];
  posOrig[blockIdx.y * n + orig[s]] = make_float2(q.x, q.y);
}

__global__ __launch_bounds__(TILE) void k_com_partial(const float2 *__restrict__ posOrig, uint32_t n,
                                                      double2 *__restrict__ partial) {
  __shared__ double2 sh[TILE];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  double2 v = make_double2(0.0, 0.0);
  if (l < n) {
    const float2 p = posOrig[blockIdx.y * n + l];
    v = make_double2((double)p.x, (double)p.y);
  }
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int w = TILE / 2; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sh[threadIdx.x].x += sh[threadIdx.x + w].x;
      sh[threadIdx.x].y += sh[threadIdx.x + w].y;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}

// one 64-lane wave per simulation: lane k sums partials k, k+64, ... in order, then a fixed
// shuffle tree combines the 64 lane sums
__global__ __launch_bounds__(64) void k_com_final(const double2 *__restrict__ partial, uint32_t nb, uint32_t n,
                                                  double2 *__restrict__ out) {
  double sx = 0.0, sy = 0.0;
  for (uint32_t b = threadIdx.x; b < nb; b += 64u) {
    const double2 p = partial[blockIdx.x * nb + b];
    sx += p.x;
    sy += p.y;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sx += __shfl_xor(sx, d, 64);
    sy += __shfl_xor(sy, d, 64);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = make_double2(sx / (double)n, sy / (double)n);
}

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

// ---- the object -----------------------------------------------------------------------------

struct pbSim {
  std::vector<PbDevParams> hP;  // one parameter block per simulation
  PbDevParams *dP = nullptr;
  SimParams host;  // schedule-relevant fields (shared by the batch): max_time, phase_update_interval, control
  uint32_t nsims = 1, n = 0, total = 0;
  int device = 0;  // the device the batch lives on; every entry point makes it the calling thread's current one
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;

  float4 *pr[2] = {nullptr, nullptr};
  float2 *vel[2] = {nullptr, nullptr};
  float *phase[2] = {nullptr, nullptr};
  int *dead[2] = {nullptr, nullptr};
  float *absA[2] = {nullptr, nullptr};
  float *absR[2] = {nullptr, nullptr};
  uint32_t *orig[2] = {nullptr, nullptr};  // LOCAL original index of each slot
  int cur = 0;                             // which copy of every array is live

  uint32_t *cellS = nullptr;  // nsims x (numCells+1), global slot indices
  uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *hist = nullptr, *slotOf = nullptr;
  uint32_t *sortedKeys = nullptr;  // keys[0] or keys[1]: composite keys of the slots, as of the last sort
  std::vector<uint32_t> layoutOrig, layoutKeys;  // host staging of pbSimSetLayoutOf
  std::vector<char> layoutGiven;
  pbRngState *rngState = nullptr;  // total, ORIGINAL order; only with an XORWOW generator (rng != 0)
  uint32_t *dMin = nullptr;  // nsims
  float *dMinD = nullptr;    // nsims
  uint32_t *hMin = nullptr;  // pinned, nsims
  float *hMinD = nullptr;    // pinned, nsims
  char *stage = nullptr;     // 36 n bytes: pos 8n | vel 8n | rad 4n | phase 4n | dead 4n | absA 4n | absR 4n
  float2 *comPos = nullptr;  // total
  double2 *comPartial = nullptr, *comOut = nullptr;
  double2 *hCom = nullptr;  // pinned, nsims

  float time = 0.0f;
  uint32_t phaseDraws = 0;
  bool haveCells = false;
  bool resortEveryStep = false;
  bool payload = false, fastOk = false;
  int variant = 2;  // force kernel: 0 reference-shaped branches, 1 branch-free, 2 (default) + fast exact math
  int resident = 0;     // 0 automatic, 1 never, 2 whenever the simulation fits one workgroup (n <= 1024)
  int lanesPerBot = 0;  // lanes per bot of the per-step force kernel: 0 automatic; 1 (throughput form), 2, 4, 8, 16
  bool debugForceBig = false;  // PB_DEBUG_FORCE_BIG under PB_ALLOW_ENV_OVERRIDES=1
  unsigned debugLdsBytes = 0;  // PB_DEBUG_LDS_BYTES under PB_ALLOW_ENV_OVERRIDES=1 (tools/occupancy_sweep.py --lds)
  int rng = 0;          // phase noise: 0 PB-RNG v1 (counter based), 1 cuRAND-compatible XORWOW (pb_xorwow.hpp)
  int forceSums = 0;    // 0: Sum|F_attr| only when a member reads it (constrained_contraction), 1: always
  bool anyConstrained = false;  // some member has constrained_contraction != 0
  pbSimStats stats{};
};

namespace {

inline bool gate(float t, float interval, float dt) {
  // the reference's fp32 schedule test (particlebot.cpp:207,212,256)
  return t - interval * floorf(t / interval) < dt;
}

inline dim3 gridOf(const pbSim *S) { return dim3(cdiv(S->n, TILE), S->nsims); }

// HIP's current device is per host thread (a new thread starts on device 0): a caller that drives
// several batches from several threads must not have to remember that
inline void useDevice(const pbSim *S) { (void)hipSetDevice(S->device); }

template <bool FUSE, bool PAYLOAD, bool FLAT, bool FASTOK, int L, int NB, bool BIG = false, bool ASUM = true>
void launchForceT(pbSim *S, int c, int o, float dt, float tNext, int doRadiusNext) {
  const uint32_t tiles = cdiv(S->n, TILE / L);
  // XCD-aware order only pays when a simulation spans many tiles
  const uint32_t perXcd = (L == 1 && tiles >= 64u) ? cdiv(tiles, 8u) : 0u;
  const dim3 grid(perXcd ? perXcd * 8u : tiles, S->nsims);
  // (debugLdsBytes: an occupancy experiment -- unused dynamic LDS that only limits workgroups per CU)
  hipLaunchKernelGGL((k_force<FUSE, PAYLOAD, FLAT, FASTOK, L, NB, BIG, ASUM>), grid, dim3(TILE), S->debugLdsBytes, S->stream, S->dP, S->pr[c],
                     S->vel[c], S->pr[o], S->vel[o], S->phase[c], S->dead[c], S->absA[c], S->absR[c], S->orig[c],
                     S->cellS, S->n, dt, tNext, doRadiusNext, perXcd);
}

// What a per-step force launch of this batch will be: the streamlined kernel or an exact one
// (kind 0 reference-shaped branches, 1 branch-free, 2 branch-free + fast exact math), and the lanes
// per bot of the exact branch-free kernels.  One place decides, so pbSimGetConfig reports what runs.
struct PbForcePlan {
  bool stream;
  int kind;
  int form;  // lanes per bot (1 = throughput form)
  bool asum; // the launch maintains absForce_a (false: dead-sum form of the throughput sweep)
};

// absForce_a has a reader (impl.cuh:167-169) or the caller asked for it (pbSimSetForceSums)
inline bool attractionSumsKept(const pbSim *S) { return S->forceSums != 0 || S->anyConstrained; }

PbForcePlan forcePlan(const pbSim *S) {
  PbForcePlan p{false, 0, 1, true};
  if (S->variant == 3 && S->total < (1u << 28) - 4u &&  // (32-bit byte offsets into posrad)
      (S->lanesPerBot == 1 || (S->lanesPerBot == 0 && S->total > 131072u))) {
    // streamlined arithmetic: throughput form only (smaller batches use the exact forms below)
    p.stream = true;
    p.kind = 3;
    return p;
  }
  // variant 0: reference-shaped branches; 1: branch-free; 2 (default): branch-free + fast exact math
  p.kind = S->variant == 0 ? 0 : (S->variant == 1 || !S->fastOk) ? 1 : 2;
  // A per-step launch of a small or medium batch is bound by one wave's serial neighbour loop, not
  // by VALU throughput, so bots get L = 8 or 4 lanes each while the chip has lanes to spare
  // (measured on MI355X, one simulation on the bench lattice, us/step for L = 1/2/4/8/16, dead-sum forms,
  //  profiles/r2_lanes_sweep.txt: 300 bots 21.2/13.0/8.8/6.6/5.7, 8192 bots 25.6/14.4/9.5/7.1/6.6,
  //  and for L = 16/32/64 on the final build: 100 bots 5.19/4.85/4.78, 1000 bots 5.39/5.06/4.83, 2000 bots
  //  5.42/5.09/5.20, 4000 bots 5.52/5.71/6.64,
  //  12000 bots 22.4/14.4/9.5/7.6/7.7, 3x10^4 22.1/14.6/10.6/10.5/12.8, 49152 22.1/16.4/13.1/14.0/17.6,
  //  10^5 28.3/21.7/21.5/23.7/30.6, 131072 27.9/24.6/25.0/28.1/37.2, 2x10^5 27.2/31.1/33.4/39.2/53.6).
  // Only the branch-free kernels have the multi-lane forms.
  if (p.kind != 0) {
    const int want = S->lanesPerBot;
    if (want == 64 || (want == 0 && S->total <= 1280u)) p.form = 64;
    else if (want == 32 || (want == 0 && S->total <= 2560u)) p.form = 32;
    else if (want == 16 || (want == 0 && S->total <= 8192u)) p.form = 16;
    else if (want == 8 || (want == 0 && S->total <= 40960u)) p.form = 8;
    else if (want == 4 || (want == 0 && S->total <= 131072u)) p.form = 4;
    else if (want == 2) p.form = 2;
    // the dead-sum forms exist for the branch-free kernels
    if (PB_THROUGHPUT_NB == 1 && !attractionSumsKept(S)) p.asum = false;
  }
  return p;
}

void launchForce(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext) {
  const bool payload = S->payload;
  const PbForcePlan plan = forcePlan(S);
  if (plan.stream) {
    const uint32_t tiles = cdiv(S->n, TILE);
    const uint32_t perXcd = tiles >= 64u ? cdiv(tiles, 8u) : 0u;
    const dim3 grid(perXcd ? perXcd * 8u : tiles, S->nsims);
#define PB_STREAM(F, PL)                                                                                       \
  hipLaunchKernelGGL((k_force_stream<F, PL>), grid, dim3(TILE), 0, S->stream, S->dP, S->pr[c], S->vel[c],     \
                     S->pr[o], S->vel[o], S->phase[c], S->dead[c], S->absA[c], S->absR[c], S->orig[c],        \
                     S->cellS, S->n, dt, tNext, doRadiusNext, perXcd)
    if (fuse && payload) PB_STREAM(true, true);
    else if (fuse) PB_STREAM(true, false);
    else if (payload) PB_STREAM(false, true);
    else PB_STREAM(false, false);
#undef PB_STREAM
    return;
  }
  const int kind = plan.kind, form = plan.form;
  // 32-bit byte offsets into posrad stop at 2^28 slots (debugForceBig: tests run the 64-bit form on small batches)
  const bool big = S->total >= (1u << 28) - 8u || S->debugForceBig;
#define PB_CASE(F, PL, K, FL, FA)                                                                        \
  if (fuse == F && payload == PL && kind == K) {                                                         \
    if (FL && !plan.asum) {                                                                                        \
      if (form == 64) return launchForceT<F, PL, FL, FA, (FL ? 64 : 1), 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext); \
      if (form == 32) return launchForceT<F, PL, FL, FA, (FL ? 32 : 1), 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext); \
      if (form == 16) return launchForceT<F, PL, FL, FA, (FL ? 16 : 1), 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext); \
      if (form == 8) return launchForceT<F, PL, FL, FA, (FL ? 8 : 1), 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext); \
      if (form == 4) return launchForceT<F, PL, FL, FA, (FL ? 4 : 1), 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext); \
      if (form == 2) return launchForceT<F, PL, FL, FA, (FL ? 2 : 1), 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext); \
    }                                                                                                              \
    if (FL && form == 64) return launchForceT<F, PL, FL, FA, (FL ? 64 : 1), 1>(S, c, o, dt, tNext, doRadiusNext); \
    if (FL && form == 32) return launchForceT<F, PL, FL, FA, (FL ? 32 : 1), 1>(S, c, o, dt, tNext, doRadiusNext); \
    if (FL && form == 16) return launchForceT<F, PL, FL, FA, (FL ? 16 : 1), 1>(S, c, o, dt, tNext, doRadiusNext); \
    if (FL && form == 8) return launchForceT<F, PL, FL, FA, (FL ? 8 : 1), 1>(S, c, o, dt, tNext, doRadiusNext); \
    if (FL && form == 4) return launchForceT<F, PL, FL, FA, (FL ? 4 : 1), 1>(S, c, o, dt, tNext, doRadiusNext); \
    if (FL && form == 2) return launchForceT<F, PL, FL, FA, (FL ? 2 : 1), 1>(S, c, o, dt, tNext, doRadiusNext); \
    if (FL && big && !plan.asum) return launchForceT<F, PL, FL, FA, 1, 1, FL, !FL>(S, c, o, dt, tNext, doRadiusNext);                \
    if (FL && big) return launchForceT<F, PL, FL, FA, 1, 1, FL>(S, c, o, dt, tNext, doRadiusNext);                                   \
    if (FL && !plan.asum) return launchForceT<F, PL, FL, FA, 1, 1, false, !FL>(S, c, o, dt, tNext, doRadiusNext);                    \
    return launchForceT<F, PL, FL, FA, 1, (FL ? PB_THROUGHPUT_NB : 1)>(S, c, o, dt, tNext, doRadiusNext);                            \
  }
  PB_CASE(true, true, 0, false, false)
  PB_CASE(true, true, 1, true, false)
  PB_CASE(true, true, 2, true, true)
  PB_CASE(true, false, 0, false, false)
  PB_CASE(true, false, 1, true, false)
  PB_CASE(true, false, 2, true, true)
  PB_CASE(false, true, 0, false, false)
  PB_CASE(false, true, 1, true, false)
  PB_CASE(false, true, 2, true, true)
  PB_CASE(false, false, 0, false, false)
  PB_CASE(false, false, 1, true, false)
  PB_CASE(false, false, 2, true, true)
#undef PB_CASE
}

// ---- resident form (k_resident) ----------------------------------------------------------------
// lanes per bot for a simulation of n bots held by one 1024-lane workgroup (0: does not fit)
inline int residentLanes(uint32_t n) { return n <= 128u ? 8 : n <= 256u ? 4 : n <= 512u ? 2 : n <= 1024u ? 1 : 0; }

bool residentWanted(const pbSim *S) {
  if (S->resident == 1 || S->variant == 0 || S->resortEveryStep || residentLanes(S->n) == 0) return false;
  if (S->lanesPerBot != 0 && S->resident != 2) return false;  // an explicit per-step form was asked for
  if (S->resident == 2) return true;
  // automatic: cost model fitted to MI355X measurements (microseconds per timestep of the whole batch,
  // dead-sum forms: profiles/r2_resident_sweep.txt, tools/resident_sweep.py).  One CU
  // per simulation costs the same however many simulations there are (up to one per CU): 5.3 us at 100
  // bots, 8.5 at 201, 11.2 at 300, 15.8 at 500, 25.3 at 1000 (the slope changes with the lanes per bot the
  // simulation's size allows); a per-step launch costs a ~5.5 us dependent-latency floor plus a term in
  // the TOTAL number of bots that depends on its lanes-per-bot form.  So the resident form wins for
  // ensembles of many small simulations, and loses for a lone simulation that per-step launches spread
  // over many CUs (at ~100 bots the two are equal).
  const double n = S->n, total = S->total;
  const double oneCu = n <= 128.0 ? 2.6 + 0.027 * n : n <= 256.0 ? 3.0 + 0.0275 * n
                       : n <= 512.0 ? 4.3 + 0.023 * n : 4.8 + 0.0205 * n;
  const double residentUs = oneCu * (S->nsims > 256u ? S->nsims / 256.0 : 1.0);
  const double perStepUs = total <= 2560.0 ? 4.75 + total / 10000.0
                           : total <= 8192.0 ? 5.4 + total / 6000.0
                           : total <= 40960.0 ? 5.6 + total / 7000.0
                           : total <= 131072.0 ? 6.0 + total / 7800.0 : 18.0 + total / 19000.0;
  return residentUs < perStepUs;
}

template <bool PAYLOAD, bool FASTOK, bool ASUM>
void launchResidentT(pbSim *S, float dt, float t0, int m, int lightWave) {
  const int c = S->cur;
  const int L = residentLanes(S->n);
  const dim3 grid(S->nsims), block(cdiv(S->n * (uint32_t)L, 64u) * 64u);
#define PB_RES(LL)                                                                                      \
  hipLaunchKernelGGL((k_resident<PAYLOAD, FASTOK, LL, ASUM>), grid, block, 0, S->stream, S->dP, S->pr[c], S->vel[c], \
                     S->phase[c], S->dead[c], S->absA[c], S->absR[c], S->orig[c], S->cellS, S->n, dt, t0, m,   \
                     lightWave)
  if (L == 8) PB_RES(8);
  else if (L == 4) PB_RES(4);
  else if (L == 2) PB_RES(2);
  else PB_RES(1);
#undef PB_RES
}

void launchResident(pbSim *S, float dt, float t0, int m, int lightWave) {
  const bool fast = S->variant >= 2 && S->fastOk;
  const bool asum = attractionSumsKept(S);
#define PB_RESL(PL, FA)                                                   \
  do {                                                                    \
    if (asum) launchResidentT<PL, FA, true>(S, dt, t0, m, lightWave);     \
    else launchResidentT<PL, FA, false>(S, dt, t0, m, lightWave);         \
  } while (0)
  if (S->payload) {
    if (fast) PB_RESL(true, true);
    else PB_RESL(true, false);
  } else {
    if (fast) PB_RESL(false, true);
    else PB_RESL(false, false);
  }
#undef PB_RESL
}

int resort(pbSim *S) {
  const uint32_t n = S->n;
  const int c = S->cur, o = c ^ 1;
  const dim3 g = gridOf(S), b(TILE);
  hipLaunchKernelGGL(k_hash, g, b, 0, S->stream, S->dP, S->pr[c], S->orig[c], S->keys[0], S->vals[0], S->slotOf, n);
  hipError_t e;
  const int bits = pbKeyBits(S->hP[0].numCells) + (S->nsims > 1 ? pbKeyBits(S->nsims) : 0);
  const int where =
      pbRadixSortPairs(S->keys[0], S->vals[0], S->keys[1], S->vals[1], S->hist, S->total, bits, S->stream, &e);
  if (where < 0) PB_TRY(e);
  hipLaunchKernelGGL(k_permute, g, b, 0, S->stream, S->vals[where], S->slotOf, S->pr[c], S->vel[c], S->phase[c],
                     S->dead[c], S->absA[c], S->absR[c], S->pr[o], S->vel[o], S->phase[o], S->dead[o], S->absA[o],
                     S->absR[o], S->orig[o], n);
  hipLaunchKernelGGL(k_cell_scan, dim3(cdiv(S->hP[0].numCells + 1u, TILE), S->nsims), b, 0, S->stream,
                     S->keys[where], S->total, S->cellS, S->hP[0].numCells);
  PB_TRY(hipGetLastError());
  S->cur = o;
  S->haveCells = true;
  S->sortedKeys = S->keys[where];
  S->stats.resorts++;
  return PB_OK;
}

int phaseUpdate(pbSim *S) {
  // particlebot.cpp:212-237.  Each simulation's min distance goes back to the host (4 bytes each)
  // because the reference takes its square root with glibc powf there (:219); max_d is unused.
  const uint32_t n = S->n;
  const int c = S->cur;
  const dim3 g = gridOf(S), b(TILE);
  // (the per-workgroup partial results borrow the centroid reduction's scratch: 16 bytes per workgroup
  //  there; dMin[nsims .. 2 nsims) holds the last-NaN indices)
  uint32_t *partial = (uint32_t *)S->comPartial;
  int *lastNan = (int *)(S->dMin + S->nsims);
  const uint32_t nb = cdiv(n, TILE);
  hipLaunchKernelGGL(k_last_nan, g, b, 0, S->stream, S->dP, S->pr[c], S->orig[c], n, (int *)partial);
  hipLaunchKernelGGL(k_last_nan_final, dim3(S->nsims), dim3(64), 0, S->stream, (const int *)partial, nb, lastNan);
  hipLaunchKernelGGL(k_min_dist2, g, b, 0, S->stream, S->dP, S->pr[c], S->orig[c], lastNan, n, partial);
  hipLaunchKernelGGL(k_min_final, dim3(S->nsims), dim3(64), 0, S->stream, partial, nb, S->dMin);
  PB_TRY(hipMemcpyAsync(S->hMin, S->dMin, sizeof(uint32_t) * 2 * S->nsims, hipMemcpyDeviceToHost, S->stream));
  PB_TRY(hipStreamSynchronize(S->stream));
  for (uint32_t k = 0; k < S->nsims; k++) {
    float minD2;
    memcpy(&minD2, &S->hMin[k], sizeof(float));
    S->hMinD[k] = powf(minD2, 0.5f);
    // the reference's loop ends on NaN when the LAST bot's distance is NaN (see k_last_nan)
    if ((int)S->hMin[S->nsims + k] == (int)n - 1) S->hMinD[k] = nanf("");
  }
  PB_TRY(hipMemcpyAsync(S->dMinD, S->hMinD, sizeof(float) * S->nsims, hipMemcpyHostToDevice, S->stream));
  hipLaunchKernelGGL(k_phase, g, b, 0, S->stream, S->dP, S->pr[c], S->orig[c], S->phase[c], n, S->dMinD,
                     S->phaseDraws, S->rng != PB_RNG_COUNTER ? S->rngState : (pbRngState *)nullptr);
  PB_TRY(hipGetLastError());
  // the draw counter advances when any simulation draws; simulations with phase_std == 0 skip it
  bool anyNoise = false;
  for (const PbDevParams &p : S->hP) anyNoise = anyNoise || p.phase_std != 0.0f;
  if (anyNoise) S->phaseDraws++;
  S->stats.phase_updates++;
  return PB_OK;
}

int stepMany(pbSim *S, float dt, float sortInterval, int nsteps, int *done) {
  const uint32_t n = S->n;
  const dim3 gA = gridOf(S), b(TILE);
  const float pui = S->host.phase_update_interval;
  const bool lightWave = (S->host.control == LIGHT_WAVE);
  bool ahead = false;  // true: radius+integration of the coming step are already applied
  const bool resident = residentWanted(S);
  int k = 0;
  for (; k < nsteps; k++) {
    const float t = S->time;
    if (t > S->host.max_time) break;  // particlebot.cpp:174-176 (the reference exits the process)
    if (!ahead) {
      if (lightWave && gate(t, pui, dt)) {
        const int rc = phaseUpdate(S);
        if (rc) return rc;
      }
      if (resident && S->haveCells && !gate(t, sortInterval, dt)) {
        // whole steps up to (not including) the next one that needs the host: a re-sort, a phase
        // update, the end of the run or of this call.  tt repeats the fp32 time accumulation.
        int m = 1;
        float tt = t + dt;
        while (k + m < nsteps && !(tt > S->host.max_time) && !(lightWave && gate(tt, pui, dt)) &&
               !gate(tt, sortInterval, dt)) {
          m++;
          tt += dt;
        }
        launchResident(S, dt, t, m, (int)lightWave);
        S->time = tt;
        S->stats.steps += m;
        S->stats.resident_launches++;
        k += m - 1;
        continue;
      }
      const int c = S->cur;
      hipLaunchKernelGGL(k_state, gA, b, 0, S->stream, S->dP, S->pr[c], S->vel[c], S->phase[c], S->dead[c],
                         S->absA[c], S->absR[c], n, t, dt, (int)(lightWave && t >= 0));
      S->stats.state_launches++;
    }
    if (S->resortEveryStep || !S->haveCells || gate(t, sortInterval, dt)) {
      const int rc = resort(S);
      if (rc) return rc;
    }
    const float tNext = t + dt;
    // Fuse the next step's radius+integration unless this is the last step of the batch or the
    // next step will not run.  A phase update due at the start of the next step only needs the
    // positions of THIS step's integration, which are final now, so it runs before the launch.
    // (resident form: never run ahead, so that the next step can start a resident stretch)
    const bool fuse = !resident && (k + 1 < nsteps) && !(tNext > S->host.max_time);
    if (fuse && lightWave && gate(tNext, pui, dt)) {
      const int rc = phaseUpdate(S);
      if (rc) return rc;
    }
    const int c = S->cur, o = c ^ 1;
    launchForce(S, fuse, c, o, dt, tNext, (int)(fuse && lightWave && tNext >= 0));
    if (fuse) S->stats.fused_launches++;
    else S->stats.plain_launches++;
    // pr/vel moved to the other copy; the remaining arrays did not.  Swap just those two.
    {
      float4 *tp = S->pr[c];
      S->pr[c] = S->pr[o];
      S->pr[o] = tp;
      float2 *tv = S->vel[c];
      S->vel[c] = S->vel[o];
      S->vel[o] = tv;
    }
    S->time = tNext;
    S->stats.steps++;
    ahead = fuse;
  }
  PB_TRY(hipGetLastError());
  if (done) *done = k;
  return PB_OK;
}

int gatherToStage(pbSim *S, uint32_t sim) {
  const size_t n = S->n;
  char *st = S->stage;
  const int c = S->cur;
  hipLaunchKernelGGL(k_get_state, dim3(cdiv(S->n, TILE)), dim3(TILE), 0, S->stream, sim, S->orig[c], S->pr[c],
                     S->vel[c], S->phase[c], S->dead[c], S->absA[c], S->absR[c], (float2 *)st, (float2 *)(st + 8 * n),
                     (float *)(st + 16 * n), (float *)(st + 20 * n), (int *)(st + 24 * n), (float *)(st + 28 * n),
                     (float *)(st + 32 * n), S->n);
  PB_TRY(hipGetLastError());
  return PB_OK;
}

}  // namespace

extern "C" {

const char *pbGetLastErrorString(void) { return g_lastError.c_str(); }

void pbSimDestroy(pbSim *S) {
  if (!S) return;
  useDevice(S);
  if (S->stream) (void)hipStreamSynchronize(S->stream);
  for (int i = 0; i < 2; i++) {
    (void)hipFree(S->pr[i]);
    (void)hipFree(S->vel[i]);
    (void)hipFree(S->phase[i]);
    (void)hipFree(S->dead[i]);
    (void)hipFree(S->absA[i]);
    (void)hipFree(S->absR[i]);
    (void)hipFree(S->orig[i]);
    (void)hipFree(S->keys[i]);
    (void)hipFree(S->vals[i]);
  }
  (void)hipFree(S->dP);
  (void)hipFree(S->cellS);
  (void)hipFree(S->hist);
  (void)hipFree(S->slotOf);
  (void)hipFree(S->rngState);
  (void)hipFree(S->dMin);
  (void)hipFree(S->dMinD);
  (void)hipFree(S->stage);
  (void)hipFree(S->comPos);
  (void)hipFree(S->comPartial);
  (void)hipFree(S->comOut);
  if (S->hMin) (void)hipHostFree(S->hMin);
  if (S->hMinD) (void)hipHostFree(S->hMinD);
  if (S->hCom) (void)hipHostFree(S->hCom);
  if (S->ev0) (void)hipEventDestroy(S->ev0);
  if (S->ev1) (void)hipEventDestroy(S->ev1);
  if (S->stream) (void)hipStreamDestroy(S->stream);
  delete S;
}

int pbSimCreateBatch(pbSim **out, const SimParams *params, int nsims, float wallHalf) {
  if (!out || !params || nsims < 1) {
    g_lastError = "pbSimCreateBatch: null argument or nsims < 1";
    return PB_ERR_ARG;
  }
  *out = nullptr;
  const uint32_t gx = params[0].gridSize.x, gy = params[0].gridSize.y;
  if (gx < 8 || gy < 8 || (gx & (gx - 1)) || (gy & (gy - 1)) || params[0].numCells != gx * gy) {
    g_lastError = "pbSimCreate: gridSize must be a power of two >= 8 per axis and numCells = x*y";
    return PB_ERR_ARG;
  }
  if (params[0].nCells == 0) {
    g_lastError = "pbSimCreate: nCells must be > 0";
    return PB_ERR_ARG;
  }
  for (int k = 1; k < nsims; k++) {
    const SimParams &a = params[0], &b = params[k];
    if (a.nCells != b.nCells || a.gridSize.x != b.gridSize.x || a.gridSize.y != b.gridSize.y ||
        a.numCells != b.numCells || a.max_time != b.max_time ||
        a.phase_update_interval != b.phase_update_interval || a.control != b.control ||
        (a.nDead == -1) != (b.nDead == -1)) {
      g_lastError = "pbSimCreateBatch: simulations of one batch must share nCells, grid, max_time, "
                    "phase_update_interval, control and payload mode";
      return PB_ERR_ARG;
    }
  }
  // (slots are 32-bit; below 2^28 bots the throughput sweep addresses posrad with 32-bit BYTE offsets,
  //  above it switches to 64-bit ones)
  if ((uint64_t)params[0].nCells * (uint64_t)nsims > 0xFFFFFFE0ull ||
      (uint64_t)params[0].numCells * (uint64_t)nsims > 0xFFFFFFF0ull) {
    g_lastError = "pbSimCreateBatch: batch too large (at most 2^32 bots and 2^32 cells in one batch)";
    return PB_ERR_ARG;
  }
  if (nsims > 65535) {  // members ride in gridDim.y
    g_lastError = "pbSimCreateBatch: at most 65535 simulations in one batch";
    return PB_ERR_ARG;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    g_lastError = "pbSimCreate: no HIP device visible";
    return PB_ERR_NO_DEVICE;
  }
  pbSim *S = new pbSim();
  (void)hipGetDevice(&S->device);
  S->host = params[0];
  S->host.x1obs = S->host.x2obs = S->host.y1obs = S->host.y2obs = nullptr;
  S->host.x_cir_obs = S->host.y_cir_obs = S->host.r_cir_obs = nullptr;
  S->nsims = (uint32_t)nsims;
  S->n = params[0].nCells;
  S->total = S->nsims * S->n;
  S->hP.resize(nsims);
  S->payload = params[0].nDead == -1;
  S->fastOk = true;
  for (int k = 0; k < nsims; k++) {
    pbFlattenParams(S->hP[k], params[k], wallHalf);
    S->fastOk = S->fastOk && pbFastMathAllowed(S->hP[k]);
    S->anyConstrained = S->anyConstrained || S->hP[k].constrained_contraction != 0u;
  }
  // A/B switches for tools/ab_bench.py: honoured only under PB_ALLOW_ENV_OVERRIDES=1 and through the
  // same range checks as the setters, so a stray variable cannot silently change what a caller runs
  if (const char *allow = getenv("PB_ALLOW_ENV_OVERRIDES"); allow && atoi(allow) == 1) {
    int rc = PB_OK;
    if (const char *v = getenv("PB_FORCE_VARIANT")) rc |= pbSimSetForceVariant(S, atoi(v));
    if (const char *v = getenv("PB_LANES_PER_BOT")) rc |= pbSimSetLanesPerBot(S, atoi(v));
    if (const char *v = getenv("PB_RESIDENT")) rc |= pbSimSetResident(S, atoi(v));
    if (const char *v = getenv("PB_FORCE_SUMS")) rc |= pbSimSetForceSums(S, atoi(v));
    if (const char *v = getenv("PB_DEBUG_LDS_BYTES")) S->debugLdsBytes = (unsigned)std::min(atol(v), 65536L);
    if (const char *v = getenv("PB_DEBUG_FORCE_BIG")) S->debugForceBig = atoi(v) != 0;
    if (rc != PB_OK) {
      g_lastError = "pbSimCreateBatch: PB_FORCE_VARIANT / PB_LANES_PER_BOT / PB_RESIDENT out of range";
      delete S;
      return PB_ERR_ARG;
    }
  }
  const size_t n = S->n, total = S->total, G1 = (size_t)S->hP[0].numCells + 1;
#define PB_TRY_NEW(expr)                                             \
  do {                                                               \
    hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) {                                          \
      g_lastError = std::string(hipGetErrorName(e_)) + " in " #expr; \
      pbSimDestroy(S);                                               \
      return PB_ERR_HIP;                                             \
    }                                                                \
  } while (0)
  PB_TRY_NEW(hipStreamCreateWithFlags(&S->stream, hipStreamNonBlocking));
  PB_TRY_NEW(hipEventCreate(&S->ev0));
  PB_TRY_NEW(hipEventCreate(&S->ev1));
  PB_TRY_NEW(hipMalloc((void **)&S->dP, sizeof(PbDevParams) * nsims));
  PB_TRY_NEW(hipMemcpyAsync(S->dP, S->hP.data(), sizeof(PbDevParams) * nsims, hipMemcpyHostToDevice, S->stream));
  for (int i = 0; i < 2; i++) {
    // spare elements: the neighbour sweeps prefetch up to three slots past the range they are walking
    PB_TRY_NEW(hipMalloc((void **)&S->pr[i], sizeof(float4) * (total + 4)));
    PB_TRY_NEW(hipMalloc((void **)&S->vel[i], sizeof(float2) * (total + 4)));
    PB_TRY_NEW(hipMalloc((void **)&S->phase[i], sizeof(float) * total));
    PB_TRY_NEW(hipMalloc((void **)&S->dead[i], sizeof(int) * total));
    PB_TRY_NEW(hipMalloc((void **)&S->absA[i], sizeof(float) * total));
    PB_TRY_NEW(hipMalloc((void **)&S->absR[i], sizeof(float) * total));
    PB_TRY_NEW(hipMalloc((void **)&S->orig[i], sizeof(uint32_t) * total));
    PB_TRY_NEW(hipMalloc((void **)&S->keys[i], sizeof(uint32_t) * total));
    PB_TRY_NEW(hipMalloc((void **)&S->vals[i], sizeof(uint32_t) * total));
    PB_TRY_NEW(hipMemsetAsync(S->pr[i], 0, sizeof(float4) * (total + 4), S->stream));
    PB_TRY_NEW(hipMemsetAsync(S->vel[i], 0, sizeof(float2) * (total + 4), S->stream));
    PB_TRY_NEW(hipMemsetAsync(S->phase[i], 0, sizeof(float) * total, S->stream));
    PB_TRY_NEW(hipMemsetAsync(S->dead[i], 0, sizeof(int) * total, S->stream));
    PB_TRY_NEW(hipMemsetAsync(S->absA[i], 0, sizeof(float) * total, S->stream));
    PB_TRY_NEW(hipMemsetAsync(S->absR[i], 0, sizeof(float) * total, S->stream));
  }
  PB_TRY_NEW(hipMalloc((void **)&S->cellS, sizeof(uint32_t) * G1 * nsims));
  PB_TRY_NEW(hipMalloc((void **)&S->hist, sizeof(uint32_t) * pbSortHistEntries(S->total)));
  PB_TRY_NEW(hipMalloc((void **)&S->slotOf, sizeof(uint32_t) * total));
  PB_TRY_NEW(hipMalloc((void **)&S->dMin, sizeof(uint32_t) * 2 * nsims));  // minima | last-NaN indices
  PB_TRY_NEW(hipMalloc((void **)&S->dMinD, sizeof(float) * nsims));
  PB_TRY_NEW(hipMalloc((void **)&S->stage, 36 * n));
  PB_TRY_NEW(hipMalloc((void **)&S->comPos, sizeof(float2) * total));
  PB_TRY_NEW(hipMalloc((void **)&S->comPartial, sizeof(double2) * cdiv(S->n, TILE) * nsims));
  PB_TRY_NEW(hipMalloc((void **)&S->comOut, sizeof(double2) * nsims));
  PB_TRY_NEW(hipHostMalloc((void **)&S->hMin, sizeof(uint32_t) * 2 * nsims));
  PB_TRY_NEW(hipHostMalloc((void **)&S->hMinD, sizeof(float) * nsims));
  PB_TRY_NEW(hipHostMalloc((void **)&S->hCom, sizeof(double2) * nsims));
  hipLaunchKernelGGL(k_iota, gridOf(S), dim3(TILE), 0, S->stream, S->orig[0], S->n);
  // attraction factor column and a defined radius for every slot
  for (uint32_t k = 0; k < S->nsims; k++)
    hipLaunchKernelGGL(k_set_state, dim3(cdiv(S->n, TILE)), dim3(TILE), 0, S->stream, S->dP, k, S->orig[0], S->pr[0],
                       S->vel[0], S->phase[0], S->dead[0], (const float2 *)nullptr, (const float2 *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const int *)nullptr, S->n, 0u, S->n);
  PB_TRY_NEW(hipGetLastError());
  PB_TRY_NEW(hipStreamSynchronize(S->stream));
#undef PB_TRY_NEW
  *out = S;
  return PB_OK;
}

int pbSimCreate(pbSim **out, const SimParams *params, float wallHalf) {
  return pbSimCreateBatch(out, params, 1, wallHalf);
}

int pbSimBatchSize(pbSim *S, unsigned *nsims, unsigned *nbots) {
  if (!S) return PB_ERR_ARG;
  if (nsims) *nsims = S->nsims;
  if (nbots) *nbots = S->n;
  return PB_OK;
}

int pbSimSetStateRangeOf(pbSim *S, unsigned sim, unsigned start, unsigned count, const float *pos,
                         const float *vel, const float *rad, const float *phase, const int *dead) {
  if (!S || sim >= S->nsims) return PB_ERR_ARG;
  useDevice(S);
  if (start > S->n || count > S->n - start) {
    g_lastError = "pbSimSetStateRangeOf: [start, start + count) must lie inside the simulation's bots";
    return PB_ERR_ARG;
  }
  if (count == 0) return PB_OK;
  const size_t n = S->n, m = count;
  char *st = S->stage;
  float2 *dPos = (float2 *)st, *dVel = (float2 *)(st + 8 * n);
  float *dRad = (float *)(st + 16 * n), *dPhase = (float *)(st + 20 * n);
  int *dDead = (int *)(st + 24 * n);
  if (pos) PB_TRY(hipMemcpyAsync(dPos, pos, 8 * m, hipMemcpyHostToDevice, S->stream));
  if (vel) PB_TRY(hipMemcpyAsync(dVel, vel, 8 * m, hipMemcpyHostToDevice, S->stream));
  if (rad) PB_TRY(hipMemcpyAsync(dRad, rad, 4 * m, hipMemcpyHostToDevice, S->stream));
  if (phase) PB_TRY(hipMemcpyAsync(dPhase, phase, 4 * m, hipMemcpyHostToDevice, S->stream));
  if (dead) PB_TRY(hipMemcpyAsync(dDead, dead, 4 * m, hipMemcpyHostToDevice, S->stream));
  const int c = S->cur;
  hipLaunchKernelGGL(k_set_state, dim3(cdiv(S->n, TILE)), dim3(TILE), 0, S->stream, S->dP, sim, S->orig[c],
                     S->pr[c], S->vel[c], S->phase[c], S->dead[c], pos ? dPos : nullptr, vel ? dVel : nullptr,
                     rad ? dRad : nullptr, phase ? dPhase : nullptr, dead ? dDead : nullptr, S->n, start, count);
  PB_TRY(hipGetLastError());
  PB_TRY(hipStreamSynchronize(S->stream));
  return PB_OK;
}

int pbSimSetStateOf(pbSim *S, unsigned sim, const float *pos, const float *vel, const float *rad,
                    const float *phase, const int *dead) {
  if (!S) return PB_ERR_ARG;
  return pbSimSetStateRangeOf(S, sim, 0u, S->n, pos, vel, rad, phase, dead);
}

int pbSimSetState(pbSim *S, const float *pos, const float *vel, const float *rad, const float *phase,
                  const int *dead) {
  return pbSimSetStateOf(S, 0, pos, vel, rad, phase, dead);
}

int pbSimGetStateOf(pbSim *S, unsigned sim, float *pos, float *vel, float *rad, float *phase, int *dead,
                    float *absForce_a, float *absForce_r) {
  if (!S || sim >= S->nsims) return PB_ERR_ARG;
  useDevice(S);
  const size_t n = S->n;
  char *st = S->stage;
  const int rc = gatherToStage(S, sim);
  if (rc) return rc;
  if (pos) PB_TRY(hipMemcpyAsync(pos, st, 8 * n, hipMemcpyDeviceToHost, S->stream));
  if (vel) PB_TRY(hipMemcpyAsync(vel, st + 8 * n, 8 * n, hipMemcpyDeviceToHost, S->stream));
  if (rad) PB_TRY(hipMemcpyAsync(rad, st + 16 * n, 4 * n, hipMemcpyDeviceToHost, S->stream));
  if (phase) PB_TRY(hipMemcpyAsync(phase, st + 20 * n, 4 * n, hipMemcpyDeviceToHost, S->stream));
  if (dead) PB_TRY(hipMemcpyAsync(dead, st + 24 * n, 4 * n, hipMemcpyDeviceToHost, S->stream));
  const bool haveA = attractionSumsKept(S);
  if (absForce_a && haveA) PB_TRY(hipMemcpyAsync(absForce_a, st + 28 * n, 4 * n, hipMemcpyDeviceToHost, S->stream));
  if (absForce_r) PB_TRY(hipMemcpyAsync(absForce_r, st + 32 * n, 4 * n, hipMemcpyDeviceToHost, S->stream));
  PB_TRY(hipStreamSynchronize(S->stream));
  // not maintained (no reader, see pbSimSetForceSums): say so instead of handing out stale numbers
  if (absForce_a && !haveA) std::fill(absForce_a, absForce_a + n, std::numeric_limits<float>::quiet_NaN());
  return PB_OK;
}

int pbSimGetState(pbSim *S, float *pos, float *vel, float *rad, float *phase, int *dead, float *absForce_a,
                  float *absForce_r) {
  return pbSimGetStateOf(S, 0, pos, vel, rad, phase, dead, absForce_a, absForce_r);
}

/* ---- layout (which bot sits in which slot, and under which stale cell) for exact checkpoints ---- */
int pbSimGetLayoutOf(pbSim *S, unsigned sim, unsigned *orig, unsigned *keys, int *sorted) {
  if (!S || sim >= S->nsims) return PB_ERR_ARG;
  useDevice(S);
  const size_t n = S->n;
  if (sorted) *sorted = S->haveCells ? 1 : 0;
  PB_TRY(hipStreamSynchronize(S->stream));
  if (orig) PB_TRY(hipMemcpy(orig, S->orig[S->cur] + (size_t)sim * n, 4 * n, hipMemcpyDeviceToHost));
  if (keys) {
    if (S->haveCells) {
      PB_TRY(hipMemcpy(keys, S->sortedKeys + (size_t)sim * n, 4 * n, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; i++) keys[i] -= sim * S->hP[0].numCells;  // strip the member number
    } else {
      memset(keys, 0, 4 * n);
    }
  }
  return PB_OK;
}

int pbSimSetLayoutOf(pbSim *S, unsigned sim, const unsigned *orig, const unsigned *keys) {
  if (!S || sim >= S->nsims || !orig || !keys) return PB_ERR_ARG;
  useDevice(S);
  const size_t n = S->n;
  if (S->layoutOrig.empty()) {
    S->layoutOrig.assign((size_t)S->total, 0);
    S->layoutKeys.assign((size_t)S->total, 0);
    S->layoutGiven.assign(S->nsims, 0);
  }
  std::vector<char> seen(n, 0);
  uint32_t prev = 0;
  for (size_t i = 0; i < n; i++) {
    if (orig[i] >= n || seen[orig[i]] || keys[i] >= S->hP[0].numCells || keys[i] < prev) {
      g_lastError = "pbSimSetLayoutOf: orig must be a permutation and keys ascending cell hashes";
      return PB_ERR_ARG;
    }
    seen[orig[i]] = 1;
    prev = keys[i];
    S->layoutOrig[sim * n + i] = orig[i];
    S->layoutKeys[sim * n + i] = sim * S->hP[0].numCells + keys[i];
  }
  S->layoutGiven[sim] = 1;
  for (char g : S->layoutGiven)
    if (!g) return PB_OK;  // wait for the other members
  // every member provided: install the slot order and rebuild the dense cell tables
  PB_TRY(hipMemcpyAsync(S->orig[S->cur], S->layoutOrig.data(), 4 * (size_t)S->total, hipMemcpyHostToDevice, S->stream));
  PB_TRY(hipMemcpyAsync(S->keys[0], S->layoutKeys.data(), 4 * (size_t)S->total, hipMemcpyHostToDevice, S->stream));
  hipLaunchKernelGGL(k_cell_scan, dim3(cdiv(S->hP[0].numCells + 1u, TILE), S->nsims), dim3(TILE), 0, S->stream,
                     S->keys[0], S->total, S->cellS, S->hP[0].numCells);
  PB_TRY(hipGetLastError());
  PB_TRY(hipStreamSynchronize(S->stream));
  S->sortedKeys = S->keys[0];
  S->haveCells = true;
  S->layoutOrig.clear();
  S->layoutKeys.clear();
  S->layoutGiven.clear();
  return PB_OK;
}

int pbSimSetForcesOf(pbSim *S, unsigned sim, const float *absForce_a, const float *absForce_r) {
  if (!S || sim >= S->nsims || !absForce_a || !absForce_r) return PB_ERR_ARG;
  useDevice(S);
  const size_t n = S->n;
  float *dA = (float *)(S->stage + 28 * n), *dR = (float *)(S->stage + 32 * n);
  PB_TRY(hipMemcpyAsync(dA, absForce_a, 4 * n, hipMemcpyHostToDevice, S->stream));
  PB_TRY(hipMemcpyAsync(dR, absForce_r, 4 * n, hipMemcpyHostToDevice, S->stream));
  const int c = S->cur;
  hipLaunchKernelGGL(k_set_forces, dim3(cdiv(S->n, TILE)), dim3(TILE), 0, S->stream, sim, S->orig[c], S->absA[c],
                     S->absR[c], dA, dR, S->n);
  PB_TRY(hipGetLastError());
  PB_TRY(hipStreamSynchronize(S->stream));
  return PB_OK;
}

int pbSimSetTime(pbSim *S, float time) {
  if (!S) return PB_ERR_ARG;
  S->time = time;
  return PB_OK;
}

int pbSimGetTime(pbSim *S, float *time) {
  if (!S || !time) return PB_ERR_ARG;
  *time = S->time;
  return PB_OK;
}

int pbSimGetPhaseDraws(pbSim *S, unsigned *draws) {
  if (!S || !draws) return PB_ERR_ARG;
  *draws = S->phaseDraws;
  return PB_OK;
}

namespace {
// (re)build every bot's XORWOW state: curand_init(seed, bot, 0) advanced by `draws` normals
int rngInit(pbSim *S, unsigned draws) {
  hipError_t e = hipSuccess;
  const uint32_t *jump = pbXorwowDeviceTable(&e);
  if (!jump) PB_TRY(e);
  if (!S->rngState) PB_TRY(hipMalloc((void **)&S->rngState, sizeof(pbRngState) * (size_t)S->total));
  hipLaunchKernelGGL(k_rng_init, gridOf(S), dim3(TILE), 0, S->stream, S->dP, S->rngState, S->n, S->rng, jump, draws);
  PB_TRY(hipGetLastError());
  PB_TRY(hipStreamSynchronize(S->stream));
  return PB_OK;
}
}  // namespace

int pbSimSetPhaseDraws(pbSim *S, unsigned draws) {
  if (!S) return PB_ERR_ARG;
  useDevice(S);
  S->phaseDraws = draws;
  if (S->rng != PB_RNG_COUNTER) return rngInit(S, draws);  // the states are a function of (seed, bot, draws)
  return PB_OK;
}

int pbSimSetRng(pbSim *S, int kind) {
  if (!S || (kind != PB_RNG_COUNTER && kind != PB_RNG_XORWOW_CURAND && kind != PB_RNG_XORWOW_ROCRAND)) {
    g_lastError = "pbSimSetRng: kind must be PB_RNG_COUNTER, PB_RNG_XORWOW_CURAND or PB_RNG_XORWOW_ROCRAND";
    return PB_ERR_ARG;
  }
  useDevice(S);
  S->rng = kind;
  S->phaseDraws = 0;
  if (kind == PB_RNG_COUNTER) return PB_OK;
  return rngInit(S, 0);
}

int pbSimGetRngStatesOf(pbSim *S, unsigned sim, pbRngState *states) {
  if (!S || sim >= S->nsims || !states || S->rng == PB_RNG_COUNTER || !S->rngState) return PB_ERR_ARG;
  useDevice(S);
  PB_TRY(hipStreamSynchronize(S->stream));
  PB_TRY(hipMemcpy(states, S->rngState + (size_t)sim * S->n, sizeof(pbRngState) * (size_t)S->n, hipMemcpyDeviceToHost));
  return PB_OK;
}

int pbSimStep(pbSim *S, float deltaTime, float sort_interval, int nsteps, int *steps_done) {
  if (!S || nsteps < 0) return PB_ERR_ARG;
  useDevice(S);
  if (steps_done) *steps_done = 0;
  return stepMany(S, deltaTime, sort_interval, nsteps, steps_done);
}

int pbSimStepTimed(pbSim *S, float deltaTime, float sort_interval, int nsteps, int *steps_done,
                   float *elapsed_ms) {
  if (!S || nsteps < 0) return PB_ERR_ARG;
  useDevice(S);
  if (steps_done) *steps_done = 0;
  PB_TRY(hipEventRecord(S->ev0, S->stream));
  const int rc = stepMany(S, deltaTime, sort_interval, nsteps, steps_done);
  if (rc) return rc;
  PB_TRY(hipEventRecord(S->ev1, S->stream));
  PB_TRY(hipEventSynchronize(S->ev1));
  float ms = 0.0f;
  PB_TRY(hipEventElapsedTime(&ms, S->ev0, S->ev1));
  if (elapsed_ms) *elapsed_ms = ms;
  return PB_OK;
}

int pbSimSynchronize(pbSim *S) {
  if (!S) return PB_ERR_ARG;
  useDevice(S);
  PB_TRY(hipStreamSynchronize(S->stream));
  return PB_OK;
}

int pbSimCentroids(pbSim *S, double *cxcy) {
  if (!S || !cxcy) return PB_ERR_ARG;
  useDevice(S);
  const uint32_t nb = cdiv(S->n, TILE);
  const int c = S->cur;
  hipLaunchKernelGGL(k_com_scatter, gridOf(S), dim3(TILE), 0, S->stream, S->orig[c], S->pr[c], S->comPos, S->n);
  hipLaunchKernelGGL(k_com_partial, dim3(nb, S->nsims), dim3(TILE), 0, S->stream, S->comPos, S->n, S->comPartial);
  hipLaunchKernelGGL(k_com_final, dim3(S->nsims), dim3(64), 0, S->stream, S->comPartial, nb, S->n, S->comOut);
  PB_TRY(hipGetLastError());
  PB_TRY(hipMemcpyAsync(S->hCom, S->comOut, sizeof(double2) * S->nsims, hipMemcpyDeviceToHost, S->stream));
  PB_TRY(hipStreamSynchronize(S->stream));
  for (uint32_t k = 0; k < S->nsims; k++) {
    cxcy[2 * k] = S->hCom[k].x;
    cxcy[2 * k + 1] = S->hCom[k].y;
  }
  return PB_OK;
}

int pbSimCentroid(pbSim *S, double *cx, double *cy) {
  if (!S) return PB_ERR_ARG;
  std::vector<double> v(2 * (size_t)S->nsims);
  const int rc = pbSimCentroids(S, v.data());
  if (rc) return rc;
  if (cx) *cx = v[0];
  if (cy) *cy = v[1];
  return PB_OK;
}

int pbSimGetStats(pbSim *S, pbSimStats *stats) {
  if (!S || !stats) return PB_ERR_ARG;
  *stats = S->stats;
  return PB_OK;
}

int pbSimSetForceVariant(pbSim *S, int variant) {
  if (!S || variant < 0 || variant > 3) return PB_ERR_ARG;
  S->variant = variant;
  return PB_OK;
}

int pbSimSetForceSums(pbSim *S, int mode) {
  if (!S || mode < 0 || mode > 1) return PB_ERR_ARG;
  S->forceSums = mode;
  return PB_OK;
}

int pbSimSetLanesPerBot(pbSim *S, int lanes) {
  if (!S || !(lanes == 0 || lanes == 1 || lanes == 2 || lanes == 4 || lanes == 8 || lanes == 16 || lanes == 32 ||
              lanes == 64))
    return PB_ERR_ARG;
  S->lanesPerBot = lanes;
  return PB_OK;
}

int pbSimSetResident(pbSim *S, int mode) {
  if (!S || mode < 0 || mode > 2) {
    g_lastError = "pbSimSetResident: mode must be 0 (automatic), 1 (never) or 2 (whenever it fits)";
    return PB_ERR_ARG;
  }
  S->resident = mode;
  return PB_OK;
}

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

int pbSimGetConfig(pbSim *S, pbSimConfig *cfg) {
  if (!S || !cfg) return PB_ERR_ARG;
  const PbForcePlan p = forcePlan(S);
  cfg->force_variant = S->variant;
  cfg->force_kind = p.kind;
  cfg->lanes_per_bot = p.form;
  cfg->resident = residentWanted(S) ? 1 : 0;
  cfg->fast_math_ok = S->fastOk ? 1 : 0;
  cfg->payload = S->payload ? 1 : 0;
  cfg->rng = S->rng;
  cfg->offsets64 = (p.form == 1 && p.kind >= 1 && !p.stream && (S->total >= (1u << 28) - 8u || S->debugForceBig)) ? 1 : 0;
  cfg->attraction_sums = attractionSumsKept(S) ? 1 : 0;
  cfg->dead_sum_form = (cfg->resident ? !attractionSumsKept(S) : (!p.stream && !p.asum)) ? 1 : 0;
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
    g_lastError = "pbClockSampleBegin: stream/buffer creation failed";
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
    g_lastError = "pbClockSampleEnd: sampler kernel failed";
    return PB_ERR_HIP;
  }
  if (mhz) *mhz = (double)v[0] / (double)v[1] * 100.0;
  if (seconds_sampled) *seconds_sampled = (double)v[1] * 1e-8;
  return PB_OK;
}

#ifdef PB_TIMELINE
int pbDebugSetTimeline(unsigned long long *deviceBuffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(pbTimelineBuf), &deviceBuffer, sizeof deviceBuffer) == hipSuccess ? PB_OK : PB_ERR_HIP;
}
#endif

int pbSimSetResortEveryStep(pbSim *S, int on) {
  if (!S) return PB_ERR_ARG;
  S->resortEveryStep = on != 0;
  return PB_OK;
}

}  // extern "C"
