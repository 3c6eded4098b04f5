// pb_resident.hip -- k_resident: the resident multi-step form for simulations of at most 1024 bots (one
// workgroup per simulation, state in registers/LDS, every timestep up to the next host event in ONE
// launch), its cost model and its launch.  Same device functions, in the same order, as k_state +
// k_force (pb_engine.hip, pb_force.hip): bit-identical results.
#include "pb_engine.hpp"
#include "pb_sweep.hpp"

namespace {

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
  constexpr bool REPLIST = L == 1 && (!ASUM || PB_ASUM_XY);  // (as k_force: the one-lane-per-bot sweep parks contact magnitudes)
  __shared__ float repLds[REPLIST ? (PB_REP_CAP + 1) * 1024 : 1];
  // L == 1 (members of 513 ... 1024 bots): 32.8 + 16.4 + 36.9 = 86 KB of the CU's 160 KB, i.e. ONE workgroup per CU.
  // Registers allow a second one only below 768 bots (>= 71 VGPRs: 6 waves per SIMD) and only a batch of more than
  // 256 such members would use it; tests/test_code_objects.py pins the figure.
  static_assert(sizeof(sPr) + sizeof(sVel) + sizeof(repLds) <= 160 * 1024, "k_resident: LDS beyond one CU");
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

// ---- resident form (k_resident) ----------------------------------------------------------------
// lanes per bot for a simulation of n bots held by one 1024-lane workgroup (0: does not fit)
inline int residentLanes(uint32_t n) { return n <= 128u ? 8 : n <= 256u ? 4 : n <= 512u ? 2 : n <= 1024u ? 1 : 0; }

}  // namespace

bool pbResidentWanted(const pbSim *S) {
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

namespace {

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

}  // namespace

void pbLaunchResident(pbSim *S, float dt, float t0, int m, int lightWave) {
  const bool asum = attractionSumsKept(S);
  // (with both sums kept the ONE-lane-per-bot sweep roots its attraction magnitudes without a domain check,
  //  pbAttractionMagnitudeSafe; the multi-lane forms do not and keep the fast path whatever magOk says -- as k_force)
  const bool magNeeded = asum && residentLanes(S->n) == 1 && PB_ASUM_XY;
  const bool fast = S->variant >= 2 && S->fastOk && (!magNeeded || S->magOk);
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
