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
//  * Three shapes of the same arithmetic, chosen per batch (pbForcePlan / pbResidentWanted), all
//    bit-identical: k_force with L = 1 (one bot per lane: throughput), k_force with L = 2/4/8 lanes
//    per bot (batches that cannot fill the chip), and k_resident (simulations of <= 1024 bots: one
//    workgroup per simulation, state in registers/LDS, many timesteps per launch).
//  * k_force_stream is the one kernel that is NOT bit-identical: the opt-in streamlined arithmetic
//    of force variant 3.
//
// This translation unit holds the object, the schedule and the re-sort / phase-update / I/O kernels; the
// force kernels live in pb_force.hip (exact, all forms), pb_stream.hip (streamlined) and pb_resident.hip.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "pb_engine.hpp"
#include "pb_xorwow.hpp"

std::string &pbLastError() {
  thread_local std::string text;
  return text;
}


namespace {

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
// that index per simulation, PLUS ONE (0: none; unsigned, so that a single arena of more than 2^31 bots
// works), k_min_dist2 then only admits bots with a larger original index; both are two-level reductions.
__global__ __launch_bounds__(TILE) void k_last_nan(const PbDevParams *__restrict__ params,
                                                   const float4 *__restrict__ pr, const uint32_t *__restrict__ orig,
                                                   uint32_t n, uint32_t *__restrict__ partial) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  uint32_t last = 0u;  // original index + 1 of the last NaN bot seen
  if (l < n) {
    const uint32_t s = blockIdx.y * n + l;
    const float4 q = pr[s];
    const float dx = P.light_x - q.x, dy = P.light_y - q.y;
    const float d2 = dx * dx + dy * dy;
    if (d2 != d2) last = orig[s] + 1u;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(last, d, 64);
    last = o > last ? o : last;
  }
  __shared__ uint32_t waveMax[TILE / 64];
  if ((threadIdx.x & 63u) == 0u) waveMax[threadIdx.x >> 6] = last;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t m = waveMax[0];
#pragma unroll
    for (int w = 1; w < TILE / 64; w++) m = waveMax[w] > m ? waveMax[w] : m;
    partial[blockIdx.y * gridDim.x + blockIdx.x] = m;
  }
}

__global__ __launch_bounds__(64) void k_last_nan_final(const uint32_t *__restrict__ partial, uint32_t nb,
                                                       uint32_t *__restrict__ out) {
  uint32_t last = 0u;
  for (uint32_t b = threadIdx.x; b < nb; b += 64u) {
    const uint32_t o = partial[blockIdx.x * nb + b];
    last = o > last ? o : last;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(last, d, 64);
    last = o > last ? o : last;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = last;
}

__global__ __launch_bounds__(TILE) void k_min_dist2(const PbDevParams *__restrict__ params,
                                                    const float4 *__restrict__ pr, const uint32_t *__restrict__ orig,
                                                    const uint32_t *__restrict__ lastNan, uint32_t n,
                                                    uint32_t *__restrict__ partial) {
  const PbDevParams &P = params[blockIdx.y];
  const uint32_t l = blockIdx.x * TILE + threadIdx.x;
  uint32_t bits = 0x7f800000u;  // +inf
  if (l < n) {
    const uint32_t s = blockIdx.y * n + l;
    const float4 q = pr[s];
    const float dx = P.light_x - q.x, dy = P.light_y - q.y;
    if (orig[s] + 1u > lastNan[blockIdx.y]) bits = __float_as_uint(dx * dx + dy * dy);  // (never NaN: those are <= lastNan)
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

// The reference's own centroid sums (particlebot.cpp:335-338: `sumX += hPos[i * 2]` over the bots in order, in fp32):
// ONE lane per simulation adds its positions serially -- the order IS the value (at 10^5 bots the fp32 running sum is
// good to ~5 digits, and the CSV prints 6) -- while the simulations of the batch run side by side.
__global__ __launch_bounds__(TILE) void k_com_serial(const float2 *__restrict__ posOrig, uint32_t n,
                                                     float2 *__restrict__ out) {
  // one workgroup per simulation: all lanes stage 2 048 positions at a time in LDS (coalesced), lane 0 adds them in
  // order (a lone lane reading global memory spends ~70 cycles per bot waiting; out of LDS ~6)
  constexpr uint32_t CH = 2048;
  __shared__ float2 sh[CH];
  const float2 *p = posOrig + (size_t)blockIdx.x * n;
  float sx = 0.0f, sy = 0.0f;
  for (uint32_t c0 = 0; c0 < n; c0 += CH) {
    const uint32_t cnt = n - c0 < CH ? n - c0 : CH;
    for (uint32_t j = threadIdx.x; j < cnt; j += TILE) sh[j] = p[c0 + j];
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t j = 0;
      for (; j + 8u <= cnt; j += 8u) {
        float2 q[8];
#pragma unroll
        for (int u = 0; u < 8; u++) q[u] = sh[j + u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          sx += q[u].x;
          sy += q[u].y;
        }
      }
      for (; j < cnt; j++) {
        sx += sh[j].x;
        sy += sh[j].y;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = make_float2(sx, sy);
}

}  // namespace

namespace {

inline bool gate(float t, float interval, float dt) {
  // the reference's fp32 schedule test (particlebot.cpp:207,212,256)
  return t - interval * floorf(t / interval) < dt;
}

// Trips per wave of the streamlined kernel's neighbour loop under its two walks (pb_stream.hip, WALK), summed over all
// waves: out[0] row by row (per stencil row the longest range of the wave's 64 lanes), out[1] flattened (the longest
// LIST of the wave's lanes).  Run at a re-sort; the host keeps the flattened walk when it saves enough trips.
__global__ __launch_bounds__(TILE) void k_walk_trips(const PbDevParams *__restrict__ params, const float4 *__restrict__ pr,
                                                     const uint32_t *__restrict__ cellSAll, uint32_t n,
                                                     unsigned long long *__restrict__ out) {
  const PbDevParams &P = params[blockIdx.y];
  uint32_t l = blockIdx.x * TILE + threadIdx.x;
  const bool alive = l < n;
  if (!alive) l = n - 1u;
  const uint32_t *__restrict__ cellS = cellSAll + (size_t)blockIdx.y * (P.numCells + 1u);
  const float4 me = pr[blockIdx.y * n + l];
  const int gx = pbCellX(P, me.x), gy = pbCellY(P, me.y);
  const uint32_t GX = P.gridX;
  const uint32_t mx0 = (uint32_t)(gx - 2) & (GX - 1u);
  const uint32_t first = (GX - mx0) < 5u ? (GX - mx0) : 5u;
  uint32_t rows = 0, list = 0;
  for (int r = 0; r < 5; r++) {
    const uint32_t row = ((uint32_t)(gy + r - 2) & (P.gridY - 1u)) << P.gridXLog2;
    uint32_t len = cellS[row + mx0 + first] - cellS[row + mx0];
    if (first < 5u) len += cellS[row + 5u - first] - cellS[row];  // the part of the row behind the x-wrap
    if (!alive) len = 0u;
    list += len;
    for (int d = 1; d < 64; d <<= 1) len = max(len, (uint32_t)__shfl_xor((int)len, d));
    rows += len;
  }
  for (int d = 1; d < 64; d <<= 1) list = max(list, (uint32_t)__shfl_xor((int)list, d));
  if ((threadIdx.x & 63u) == 0u) {
    atomicAdd(&out[0], (unsigned long long)rows);
    atomicAdd(&out[1], (unsigned long long)list);
  }
}

// The automatic choice of k_force_stream's walk for this batch, from its current cell lists: flattened when that
// saves at least 7 % of the trips (blobs: -13 ... -20 %; the bench lattice: ~0, where the row-by-row walk's better
// cache-line sharing wins).  Either walk gives the same bits.
int chooseStreamWalk(pbSim *S) {
  if (!S->haveCells) return PB_OK;
  if (!S->walkTrips) PB_TRY(hipMalloc((void **)&S->walkTrips, 2 * sizeof(unsigned long long)));
  PB_TRY(hipMemsetAsync(S->walkTrips, 0, 2 * sizeof(unsigned long long), S->stream));
  hipLaunchKernelGGL(k_walk_trips, gridOf(S), dim3(TILE), 0, S->stream, S->dP, S->pr[S->cur], S->cellS, S->n, S->walkTrips);
  PB_TRY(hipGetLastError());
  unsigned long long t[2] = {0, 0};
  PB_TRY(hipMemcpyAsync(t, S->walkTrips, sizeof t, hipMemcpyDeviceToHost, S->stream));
  PB_TRY(hipStreamSynchronize(S->stream));
  S->walkTripsHost[0] = t[0], S->walkTripsHost[1] = t[1];
  S->streamWalkAuto = t[1] * 100ull < t[0] * 93ull;
  return PB_OK;
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
  if (S->variant == 3 && S->streamWalk < 0) return chooseStreamWalk(S);
  return PB_OK;
}

// process-wide default of pbSimSetMinDistanceMode for batches created from now on (pbSetMinDistanceMode).  -1: not
// decided yet -- the first use seeds it from PB_MIN_DISTANCE_MODE (0 or 1; anything else: 0), a default that CHILD
// processes inherit (tests/conftest.py exports it when this host's libm fails the check, so that the binaries the
// tests spawn take the reference's host loop too).  An explicit pbSetMinDistanceMode always wins over the environment.
// (atomic: the two lanes of an ensemble pipeline create their batches on two threads at once, and an explicit
//  pbSetMinDistanceMode may arrive from a third; the compare-exchange lets a concurrent explicit setting win)
std::atomic<int> g_minDistanceMode{-1};
static int minDistanceDefault() {
  int cur = g_minDistanceMode.load(std::memory_order_acquire);
  if (cur < 0) {
    int m = 0;
    if (const char *v = getenv("PB_MIN_DISTANCE_MODE"))
      if ((v[0] == '0' || v[0] == '1') && v[1] == 0) m = v[0] - '0';
    if (g_minDistanceMode.compare_exchange_strong(cur, m, std::memory_order_acq_rel)) cur = m;
  }
  return cur;
}

int phaseUpdate(pbSim *S) {
  // particlebot.cpp:212-237.  The reference copies every position to the host and takes
  // min_i powf(powf(lx - x_i, 2) + powf(ly - y_i, 2), 0.5f) there (:214-228); max_d is unused.
  //  mode 0 (default): the device reduces min_i (dx*dx + dy*dy) -- 4 bytes back per simulation -- and the host
  //    takes the root with glibc powf.  Equal to the reference's value iff powf(x, 2) == x*x for every float and
  //    powf(., 0.5f) is non-decreasing: properties of the host's libm, checked exhaustively by pbHostLibmCheck
  //    (tests/test_libm_pin.py; 0 mismatches, 0 inversions on glibc 2.35).
  //  mode 1: the reference's own loop on the host over the positions in original order (8 n bytes back per
  //    simulation): no assumption about libm.  For hosts where the check fails.
  const uint32_t n = S->n;
  const int c = S->cur;
  const dim3 g = gridOf(S), b(TILE);
  if (S->minDistanceMode == 1) {
    if (S->hostPos.size() < 2 * (size_t)n) S->hostPos.resize(2 * (size_t)n);
    for (uint32_t k = 0; k < S->nsims; k++) {
      hipLaunchKernelGGL(k_com_scatter, dim3(cdiv(n, TILE), 1), b, 0, S->stream, S->orig[c] + (size_t)k * n,
                         S->pr[c] + (size_t)k * n, S->comPos, n);
      PB_TRY(hipMemcpyAsync(S->hostPos.data(), S->comPos, 8 * (size_t)n, hipMemcpyDeviceToHost, S->stream));
      PB_TRY(hipStreamSynchronize(S->stream));
      const float lx = S->hP[k].light_x, ly = S->hP[k].light_y;
      float minD = 0.0f;
      for (uint32_t i = 0; i < n; i++) {
        const float d = powf(powf(lx - S->hostPos[2 * (size_t)i], 2) + powf(ly - S->hostPos[2 * (size_t)i + 1], 2), 0.5f);
        minD = (i == 0) ? d : (minD < d ? minD : d);
      }
      S->hMinD[k] = minD;
    }
  } else {
    // (the per-workgroup partial results borrow the centroid reduction's scratch: 16 bytes per workgroup
    //  there; dMin[nsims .. 2 nsims) holds the last-NaN indices)
    uint32_t *partial = (uint32_t *)S->comPartial;
    uint32_t *lastNan = S->dMin + S->nsims;
    const uint32_t nb = cdiv(n, TILE);
    hipLaunchKernelGGL(k_last_nan, g, b, 0, S->stream, S->dP, S->pr[c], S->orig[c], n, partial);
    hipLaunchKernelGGL(k_last_nan_final, dim3(S->nsims), dim3(64), 0, S->stream, partial, nb, lastNan);
    hipLaunchKernelGGL(k_min_dist2, g, b, 0, S->stream, S->dP, S->pr[c], S->orig[c], lastNan, n, partial);
    hipLaunchKernelGGL(k_min_final, dim3(S->nsims), dim3(64), 0, S->stream, partial, nb, S->dMin);
    PB_TRY(hipMemcpyAsync(S->hMin, S->dMin, sizeof(uint32_t) * 2 * S->nsims, hipMemcpyDeviceToHost, S->stream));
    PB_TRY(hipStreamSynchronize(S->stream));
    for (uint32_t k = 0; k < S->nsims; k++) {
      float minD2;
      memcpy(&minD2, &S->hMin[k], sizeof(float));
      S->hMinD[k] = powf(minD2, 0.5f);
      // the reference's loop ends on NaN when the LAST bot's distance is NaN (see k_last_nan)
      if (S->hMin[S->nsims + k] == n) S->hMinD[k] = nanf("");  // (index + 1 of the last NaN bot)
    }
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
  const bool resident = pbResidentWanted(S);
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
        pbLaunchResident(S, dt, t, m, (int)lightWave);
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
    pbLaunchForce(S, fuse, c, o, dt, tNext, (int)(fuse && lightWave && tNext >= 0));
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

const char *pbGetLastErrorString(void) { return pbLastError().c_str(); }

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
  (void)hipFree(S->walkTrips);
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
    pbLastError() = "pbSimCreateBatch: null argument or nsims < 1";
    return PB_ERR_ARG;
  }
  *out = nullptr;
  const uint32_t gx = params[0].gridSize.x, gy = params[0].gridSize.y;
  if (gx < 8 || gy < 8 || (gx & (gx - 1)) || (gy & (gy - 1)) || params[0].numCells != gx * gy) {
    pbLastError() = "pbSimCreate: gridSize must be a power of two >= 8 per axis and numCells = x*y";
    return PB_ERR_ARG;
  }
  if (params[0].nCells == 0) {
    pbLastError() = "pbSimCreate: nCells must be > 0";
    return PB_ERR_ARG;
  }
  for (int k = 1; k < nsims; k++) {
    const SimParams &a = params[0], &b = params[k];
    if (a.nCells != b.nCells || a.gridSize.x != b.gridSize.x || a.gridSize.y != b.gridSize.y ||
        a.numCells != b.numCells || a.max_time != b.max_time ||
        a.phase_update_interval != b.phase_update_interval || a.control != b.control ||
        (a.nDead == -1) != (b.nDead == -1)) {
      pbLastError() = "pbSimCreateBatch: simulations of one batch must share nCells, grid, max_time, "
                    "phase_update_interval, control and payload mode";
      return PB_ERR_ARG;
    }
  }
  // (slots are 32-bit; below 2^28 bots the throughput sweep addresses posrad with 32-bit BYTE offsets,
  //  above it switches to 64-bit ones)
  if ((uint64_t)params[0].nCells * (uint64_t)nsims > 0xFFFFFFE0ull ||
      (uint64_t)params[0].numCells * (uint64_t)nsims > 0xFFFFFFF0ull) {
    pbLastError() = "pbSimCreateBatch: batch too large (at most 2^32 bots and 2^32 cells in one batch)";
    return PB_ERR_ARG;
  }
  if (nsims > 65535) {  // members ride in gridDim.y
    pbLastError() = "pbSimCreateBatch: at most 65535 simulations in one batch";
    return PB_ERR_ARG;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    pbLastError() = "pbSimCreate: no HIP device visible";
    return PB_ERR_NO_DEVICE;
  }
  pbSim *S = new pbSim();
  (void)hipGetDevice(&S->device);
  S->minDistanceMode = minDistanceDefault();
  S->host = params[0];
  S->host.x1obs = S->host.x2obs = S->host.y1obs = S->host.y2obs = nullptr;
  S->host.x_cir_obs = S->host.y_cir_obs = S->host.r_cir_obs = nullptr;
  S->nsims = (uint32_t)nsims;
  S->n = params[0].nCells;
  S->total = S->nsims * S->n;
  S->hP.resize(nsims);
  S->payload = params[0].nDead == -1;
  S->fastOk = true;
  S->magOk = true;
  for (int k = 0; k < nsims; k++) {
    pbFlattenParams(S->hP[k], params[k], wallHalf);
    S->fastOk = S->fastOk && pbFastMathAllowed(S->hP[k]);
    S->magOk = S->magOk && pbAttractionMagnitudeSafe(S->hP[k]);
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
    if (const char *v = getenv("PB_DEBUG_FORCE_BIG")) S->wideOffsets = atoi(v) != 0;
    if (const char *v = getenv("PB_XCD_MEMBERS")) S->xcdMembers = atoi(v) != 0;
    if (const char *v = getenv("PB_XCD_TILES_ALL")) S->xcdTilesAll = atoi(v) != 0;
    if (const char *v = getenv("PB_STREAM_WALK")) rc |= pbSimSetStreamWalk(S, atoi(v));
    if (rc != PB_OK) {
      pbLastError() = "pbSimCreateBatch: PB_FORCE_VARIANT / PB_LANES_PER_BOT / PB_RESIDENT out of range";
      delete S;
      return PB_ERR_ARG;
    }
  }
  const size_t n = S->n, total = S->total, G1 = (size_t)S->hP[0].numCells + 1;
#define PB_TRY_NEW(expr)                                             \
  do {                                                               \
    hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) {                                          \
      pbLastError() = std::string(hipGetErrorName(e_)) + " in " #expr; \
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
    pbLastError() = "pbSimSetStateRangeOf: [start, start + count) must lie inside the simulation's bots";
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
      pbLastError() = "pbSimSetLayoutOf: orig must be a permutation and keys ascending cell hashes";
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
    pbLastError() = "pbSimSetRng: kind must be PB_RNG_COUNTER, PB_RNG_XORWOW_CURAND or PB_RNG_XORWOW_ROCRAND";
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
  return pbSimStepTimedWall(S, deltaTime, sort_interval, nsteps, steps_done, elapsed_ms, nullptr);
}

int pbSimStepTimedWall(pbSim *S, float deltaTime, float sort_interval, int nsteps, int *steps_done,
                       float *elapsed_ms, double *wall_ms) {
  if (!S || nsteps < 0) return PB_ERR_ARG;
  // PB_TIMED_TRACE=1 (diagnostic): where the host's time around a timed region goes, in microseconds from entry
  static const bool trace = [] { const char *v = getenv("PB_TIMED_TRACE"); return v && v[0] == '1'; }();
  const auto h0 = std::chrono::steady_clock::now();
  auto us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count(); };
  useDevice(S);
  if (steps_done) *steps_done = 0;
  PB_TRY(hipEventRecord(S->ev0, S->stream));
  const double tRec0 = us();
  const int rc = stepMany(S, deltaTime, sort_interval, nsteps, steps_done);
  if (rc) return rc;
  const double tLaunched = us();
  PB_TRY(hipEventRecord(S->ev1, S->stream));
  const double tRec1 = us();
  // A measurement API: the caller's wall clock around this call should see the device time, not the wake-up latency of
  // an interrupt-driven wait.  So the host polls the event (about a microsecond per query) for as long as a short
  // region lasts, and only then falls back to the blocking wait.
  {
    const auto spin0 = std::chrono::steady_clock::now();
    hipError_t q;
    while ((q = hipEventQuery(S->ev1)) == hipErrorNotReady)
      if (std::chrono::steady_clock::now() - spin0 > std::chrono::milliseconds(250)) break;
    if (q != hipSuccess && q != hipErrorNotReady) PB_TRY(q);
  }
  const double tDone = us();
  PB_TRY(hipEventSynchronize(S->ev1));
  if (wall_ms) {
    // the host's clock over the same region: entry (the stream idle, the caller having synchronised) -> every launch
    // issued -> the stream drained.  ev1 was recorded behind the last launch and has completed, so the stream IS
    // drained: hipStreamQuery confirms it (a hipStreamSynchronize on an idle stream costs ~17 us here, 1 % of a 20-step
    // region, for no information) and only an unexpected "not ready" falls back to the blocking call.
    const hipError_t q = hipStreamQuery(S->stream);
    if (q == hipErrorNotReady) PB_TRY(hipStreamSynchronize(S->stream));
    else PB_TRY(q);
    *wall_ms = us() * 1e-3;
  }
  float ms = 0.0f;
  PB_TRY(hipEventElapsedTime(&ms, S->ev0, S->ev1));
  if (elapsed_ms) *elapsed_ms = ms;
  if (trace)
    fprintf(stderr, "pbSimStepTimed: %d steps: ev0 recorded %.1f us, launches issued %.1f, ev1 recorded %.1f, event complete "
            "%.1f, return %.1f; device %.1f us\n", nsteps, tRec0, tLaunched, tRec1, tDone, us(), ms * 1e3);
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

int pbSimCentroidSums(pbSim *S, float *sumxy) {
  if (!S || !sumxy) return PB_ERR_ARG;
  useDevice(S);
  const int c = S->cur;
  float2 *const out = (float2 *)S->comOut;  // (8 of the 16 bytes per simulation the mean uses)
  hipLaunchKernelGGL(k_com_scatter, gridOf(S), dim3(TILE), 0, S->stream, S->orig[c], S->pr[c], S->comPos, S->n);
  hipLaunchKernelGGL(k_com_serial, dim3(S->nsims), dim3(TILE), 0, S->stream, S->comPos, S->n, out);
  PB_TRY(hipGetLastError());
  PB_TRY(hipMemcpyAsync(S->hCom, out, sizeof(float2) * S->nsims, hipMemcpyDeviceToHost, S->stream));
  PB_TRY(hipStreamSynchronize(S->stream));
  memcpy(sumxy, S->hCom, sizeof(float2) * S->nsims);
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
  const bool becomes3 = variant == 3 && S->variant != 3;
  S->variant = variant;
  // (a batch that already has its cell lists: the walk of the streamlined kernel is chosen now, not at the next re-sort)
  if (becomes3 && S->streamWalk < 0 && S->haveCells) {
    useDevice(S);
    return chooseStreamWalk(S);
  }
  return PB_OK;
}

int pbSimSetStreamWalk(pbSim *S, int mode) {
  if (!S || mode < -1 || mode > 1) return PB_ERR_ARG;
  S->streamWalk = mode;
  if (mode < 0 && S->variant == 3 && S->haveCells) {
    useDevice(S);
    return chooseStreamWalk(S);
  }
  return PB_OK;
}

int pbSimGetStreamWalkTrips(pbSim *S, unsigned long long *rowByRow, unsigned long long *flattened) {
  if (!S) return PB_ERR_ARG;
  if (rowByRow) *rowByRow = S->walkTripsHost[0];
  if (flattened) *flattened = S->walkTripsHost[1];
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
    pbLastError() = "pbSimSetResident: mode must be 0 (automatic), 1 (never) or 2 (whenever it fits)";
    return PB_ERR_ARG;
  }
  S->resident = mode;
  return PB_OK;
}

int pbSimGetConfig(pbSim *S, pbSimConfig *cfg) {
  if (!S || !cfg) return PB_ERR_ARG;
  const PbForcePlan p = pbForcePlan(S);
  cfg->force_variant = S->variant;
  cfg->force_kind = p.kind;
  cfg->lanes_per_bot = p.form;
  cfg->resident = pbResidentWanted(S) ? 1 : 0;
  cfg->fast_math_ok = S->fastOk ? 1 : 0;
  cfg->payload = S->payload ? 1 : 0;
  cfg->rng = S->rng;
  cfg->offsets64 = (!p.stream && p.big) ? 1 : 0;
  cfg->attraction_sums = attractionSumsKept(S) ? 1 : 0;
  cfg->dead_sum_form = ((cfg->resident || p.stream) ? !attractionSumsKept(S) : !p.asum) ? 1 : 0;
  cfg->stream_walk = (p.stream && pbStreamWalk(S)) ? 1 : 0;
  return PB_OK;
}

int pbGetDevice(int *device) {
  if (!device) return PB_ERR_ARG;
  PB_TRY(hipGetDevice(device));
  return PB_OK;
}

int pbSetDevice(int device) {
  PB_TRY(hipSetDevice(device));
  return PB_OK;
}

int pbDevicePciBusId(int device, char *out, int cap) {
  if (!out || cap < 13) return PB_ERR_ARG;
  PB_TRY(hipDeviceGetPCIBusId(out, cap, device));
  return PB_OK;
}

int pbSimSetMinDistanceMode(pbSim *S, int mode) {
  if (!S || mode < 0 || mode > 1) return PB_ERR_ARG;
  S->minDistanceMode = mode;
  return PB_OK;
}

int pbSetMinDistanceMode(int mode) {
  if (mode < 0 || mode > 1) return PB_ERR_ARG;
  g_minDistanceMode.store(mode, std::memory_order_release);
  return PB_OK;
}

int pbGetMinDistanceMode(void) { return minDistanceDefault(); }

float pbHostSqrtThreshold(float c) { return pbSqrtThreshold(c); }

int pbSimSetResortEveryStep(pbSim *S, int on) {
  if (!S) return PB_ERR_ARG;
  S->resortEveryStep = on != 0;
  return PB_OK;
}

}  // extern "C"
