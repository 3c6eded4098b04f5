// pb_force.hip -- k_force: the exact per-step force kernel of the pbSim engine in all its forms, the
// table of those forms (shared by the dispatch, pbSimGetConfig and the parity tests), and the launch.
//
// Reference: collideD + collideCell + collideSpheres (particlebot_kernel_impl.cuh:541-831) and, fused,
// the next step's updateRad_light_wave (:124-181) and integrate_functor (:53-103).
#include <cxxabi.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <typeinfo>

#include "pb_engine.hpp"

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
#endif

#include "pb_sweep.hpp"

namespace {

// Forces + kick of step n (impl.cuh:657-831); with `fuse` also radius + integration of step n+1.
// PAYLOAD: object-transport mode (nDead == -1), per-pair attraction factors.  FLAT: branch-free
// pair evaluation instead of the reference-shaped branches (pbPair).
// fuse and fastOk are RUNTIME flags (round 3; they were template parameters): both are tested once
// per bot, outside the pair loop, so one instantiation serves what took four (the matrix was 128
// kernels).  fastOk: the simulation passed pbFastMathAllowed, so waves whose lanes all pass
// pbLaneFastMathOk may use the exact fast sqrt/division forms.
// BIG: 64-bit byte offsets in the neighbour sweep (batches of 2^28 bots and more, throughput form only).
// ASUM: maintain absForce_a.  false (throughput form, batches without constrained contraction):
// the attraction magnitudes are dead values and are neither computed nor stored (pbPairEvalXY).
template <bool PAYLOAD, bool FLAT, int L, int NB, bool BIG, bool ASUM>
__global__ __launch_bounds__(TILE, (NB == 2 ? PB_NB2_WAVES : PB_FORCE_WAVES)) void k_force(const PbDevParams *__restrict__ params,
                                                const float4 *__restrict__ prIn, const float2 *__restrict__ velIn,
                                                const uint32_t *__restrict__ cellSAll, float *__restrict__ absR,
                                                uint32_t n, uint32_t perXcd, uint32_t memberTiles, uint32_t nsims,
                                                float dt, int fastOk,
                                                // -- up to here (14 dwords: 16 user SGPRs, two of them the kernarg
                                                // pointer) the arguments arrive in SGPRs with the wave (kernarg preload):
                                                // the tile decode and the own-state loads need nothing else, so a launch
                                                // does not start with a scalar round trip for its arguments (fastOk in
                                                // there instead of absR: measured equal); what follows is used later --
                                                float4 *__restrict__ prOut, float2 *__restrict__ velOut,
                                                const float *__restrict__ phase, const int *__restrict__ dead,
                                                float *__restrict__ absA, const uint32_t *__restrict__ orig,
                                                float timeNext, int doRadiusNext, int fuse) {
  // memberTiles != 0 (round 5: batches of >= 8 small simulations, 1-D grid): workgroups b, b+8, b+16, ... share an XCD
  // (round-robin dispatch) and every XCD has its own L2: ALL tiles of a member go to one XCD, so that the neighbour
  // reads of a member's tiles -- the same few KB -- meet in one L2 instead of missing in eight.  BASELINE configs[3] on
  // one GPU: 32 + 32 members 7.75 -> 7.31 us per step, 64 + 64 members 10.22 -> 9.40, 8 + 8 unchanged
  // (tools/experiments/ab_xcd_members.sh); the results do not depend on the mapping.
  uint32_t member = blockIdx.y, tileX = blockIdx.x;
  if (memberTiles) {
    const uint32_t r = blockIdx.x >> 3;
    member = (r / memberTiles) * 8u + (blockIdx.x & 7u);
    tileX = r % memberTiles;
    if (member >= nsims) return;
  }
  const PbDevParams &P = params[member];
  // XCD-aware tile order (large simulations): workgroups b, b+8, b+16, ... share an XCD
  // (round-robin dispatch); give each XCD one contiguous eighth of the tiles (gridDim.x = 8*perXcd).
  // perXcd == 0: plain order (small simulations, a handful of tiles each).
  const uint32_t tile = perXcd ? (tileX & 7u) * perXcd + (tileX >> 3) : tileX;
  const uint32_t l = tile * (TILE / L) + threadIdx.x / L;  // all L lanes of a group share the bot
  const uint32_t sub = threadIdx.x % L;
#ifdef PB_TIMELINE
  // (stored at once: a start stamp kept in registers to the end cost the kernel a wave per SIMD)
  PB_TL_STAMP(0);
#endif
  if (l >= n) return;
  const uint32_t s = member * n + l;  // global slot; the cell table holds global slots too
  const uint32_t *__restrict__ cellS = cellSAll + (size_t)member * (P.numCells + 1u);

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
  // (L > 1: magnitudes are rooted inside the contact block; the both-sums throughput form has the list since round 5)
  constexpr bool REPLIST = L == 1 && FLAT && NB == 1 && (!ASUM || PB_ASUM_XY);
  __shared__ float repLds[REPLIST ? (PB_REP_CAP + 1) * TILE : 1];
  float *const repCol = &repLds[REPLIST ? threadIdx.x : 0];
  if (FLAT && fastOk && __all(pbLaneFastMathOk(me.x, me.y)))
    pbSweep<PAYLOAD, FLAT, true, L, NB, OffT, ASUM>(P, prIn, velIn, cellS, 0u, s, sub, me, v, att1, F, repCol);
  else
    pbSweep<PAYLOAD, FLAT, false, L, NB, OffT, ASUM>(P, prIn, velIn, cellS, 0u, s, sub, me, v, att1, F, repCol);
  pbObstacles(P, me.x, me.y, v.x, v.y, me.z, F);
  pbFrictionAndKick(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);

  float4 out = me;
  if (fuse) {
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

template <bool PAYLOAD, bool FLAT, int L, bool BIG, bool ASUM>
void launchForceT(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext) {
  const uint32_t tiles = cdiv(S->n, TILE / L);
  // XCD-aware order only pays when a simulation spans many tiles.  (Round 5: for the multi-lane forms too -- one
  // simulation of 4 000 ... 10^5 bots steps 1-4 % faster in its automatic form, profiles/r5_xcd_tiles_all.txt.)
  const uint32_t perXcd = ((L == 1 || S->xcdTilesAll) && tiles >= 64u) ? cdiv(tiles, 8u) : 0u;
  dim3 grid(perXcd ? perXcd * 8u : tiles, S->nsims);
  uint32_t memberTiles = 0u;
  if (S->xcdMembers && !perXcd && S->nsims >= 8u) {
    memberTiles = tiles;
    grid = dim3(cdiv(S->nsims, 8u) * 8u * tiles, 1u);
  }
  constexpr int NB = (FLAT && L == 1 && !BIG && ASUM) ? PB_THROUGHPUT_NB : 1;
  // (the both-sums throughput form roots its attraction magnitudes without a domain check: pbAttractionMagnitudeSafe)
  const bool magNeeded = ASUM && FLAT && L == 1 && PB_ASUM_XY;
  const int fastOk = (S->variant >= 2 && S->fastOk && (!magNeeded || S->magOk)) ? 1 : 0;
  // (debugLdsBytes: an occupancy experiment -- unused dynamic LDS that only limits workgroups per CU)
  hipLaunchKernelGGL((k_force<PAYLOAD, FLAT, L, NB, BIG, ASUM>), grid, dim3(TILE), S->debugLdsBytes, S->stream, S->dP,
                     S->pr[c], S->vel[c], S->cellS, S->absR[c], S->n, perXcd, memberTiles, S->nsims, dt, fastOk, S->pr[o],
                     S->vel[o], S->phase[c], S->dead[c], S->absA[c], S->orig[c], tNext, doRadiusNext, (int)fuse);
}

// The kernel's name as a profiler prints it -- "k_force<false, true, 1, 1, false, true>(PbDevParams const*, ...)" --
// built from the instantiation's own type, so that a committed rocprofv3 summary can be matched against the
// LOADED library (bench.py drops a profile's counters when the signatures differ: an argument reorder or a new
// template parameter makes the counters another kernel's).
template <bool PAYLOAD, bool FLAT, int L, bool BIG, bool ASUM>
std::string forceNameT() {
  constexpr int NB = (FLAT && L == 1 && !BIG && ASUM) ? PB_THROUGHPUT_NB : 1;
  auto b = [](bool v) { return std::string(v ? "true" : "false"); };
  return "k_force<" + b(PAYLOAD) + ", " + b(FLAT) + ", " + std::to_string(L) + ", " + std::to_string(NB) + ", " + b(BIG) +
         ", " + b(ASUM) + ">" + pbKernelArgs(typeid(&k_force<PAYLOAD, FLAT, L, NB, BIG, ASUM>).name());
}

// ---- the forms table ---------------------------------------------------------------------------
// Every shape of the exact kernel that can run, once.  The dispatch below looks a launch up here, the
// C-ABI hands the rows to the parity tests (pbForceFormCount / pbForceFormGet / pbSimSelectForceForm),
// so a form cannot exist without being enumerable.  Each row is instantiated for both payload modes.
using LaunchFn = void (*)(pbSim *, bool, int, int, float, float, int);
using NameFn = std::string (*)();
struct FormRow {
  pbForceForm form;
  LaunchFn launch[2];  // [payload]
  NameFn name[2];
};
#define PB_FORM(FL, LL, AS, BG)                                                        \
  {{FL, LL, AS, BG}, {launchForceT<false, (FL) != 0, LL, (BG) != 0, (AS) != 0>,        \
                      launchForceT<true, (FL) != 0, LL, (BG) != 0, (AS) != 0>},        \
                     {forceNameT<false, (FL) != 0, LL, (BG) != 0, (AS) != 0>,          \
                      forceNameT<true, (FL) != 0, LL, (BG) != 0, (AS) != 0>}}
const FormRow kForms[] = {
    PB_FORM(0, 1, 1, 0),  // reference-shaped branches (force variant 0)
    PB_FORM(1, 1, 1, 0),  PB_FORM(1, 1, 0, 0),   // throughput form: one bot per lane
    PB_FORM(1, 1, 1, 1),  PB_FORM(1, 1, 0, 1),   //   ... with 64-bit byte offsets (>= 2^28 bots)
    PB_FORM(1, 2, 1, 0),  PB_FORM(1, 2, 0, 0),   // L lanes per bot, ordered group sum
    PB_FORM(1, 4, 1, 0),  PB_FORM(1, 4, 0, 0),
    PB_FORM(1, 8, 1, 0),  PB_FORM(1, 8, 0, 0),
    PB_FORM(1, 16, 1, 0), PB_FORM(1, 16, 0, 0),
    PB_FORM(1, 32, 1, 0), PB_FORM(1, 32, 0, 0),
    PB_FORM(1, 64, 1, 0), PB_FORM(1, 64, 0, 0),
};
#undef PB_FORM
constexpr int kNumForms = (int)(sizeof(kForms) / sizeof(kForms[0]));

}  // namespace

PbForcePlan pbForcePlan(const pbSim *S) {
  PbForcePlan p{false, 0, 1, true, false};
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
    // 32-bit byte offsets into posrad stop at 2^28 slots (wideOffsets: tests run the 64-bit form on small
    // batches); the wide form is a throughput form
    p.big = S->total >= (1u << 28) - 8u || S->wideOffsets;
    const int want = p.big ? 1 : S->lanesPerBot;
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

std::string pbKernelArgs(const char *mangledPointerType) {
  // "void (*)(PbDevParams const*, ...)" -> "(PbDevParams const*, ...)"
  int status = 0;
  char *d = abi::__cxa_demangle(mangledPointerType, nullptr, nullptr, &status);
  std::string t = (status == 0 && d) ? d : mangledPointerType;
  free(d);
  const size_t at = t.find("(*)");
  return at == std::string::npos ? t : t.substr(at + 3);
}

void pbLaunchForce(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext) {
  const PbForcePlan plan = pbForcePlan(S);
  if (plan.stream) return pbLaunchForceStream(S, fuse, c, o, dt, tNext, doRadiusNext);
  for (const FormRow &r : kForms)
    if (r.form.flat == (plan.kind != 0) && r.form.lanes_per_bot == plan.form && (r.form.attraction_sums != 0) == plan.asum &&
        (r.form.offsets64 != 0) == plan.big)
      return r.launch[S->payload ? 1 : 0](S, fuse, c, o, dt, tNext, doRadiusNext);
  // unreachable: pbForcePlan only produces rows of the table (tests/test_capi_symbols.py walks it)
  pbLastError() = "pbLaunchForce: no kernel form for this plan";
}

extern "C" {

int pbForceFormCount(void) { return kNumForms; }

int pbForceFormGet(int index, pbForceForm *form) {
  if (index < 0 || index >= kNumForms || !form) return PB_ERR_ARG;
  *form = kForms[index].form;
  return PB_OK;
}

int pbForceFormKernelName(int index, int payload, char *buf, size_t cap) {
  if (index < 0 || index >= kNumForms || !buf) return PB_ERR_ARG;
  const std::string name = kForms[index].name[payload ? 1 : 0]();
  if (name.size() + 1 > cap) return PB_ERR_ARG;
  memcpy(buf, name.c_str(), name.size() + 1);
  return PB_OK;
}

int pbSimForceKernelName(pbSim *S, char *buf, size_t cap) {
  if (!S || !buf || cap == 0) return PB_ERR_ARG;
  const PbForcePlan plan = pbForcePlan(S);
  std::string name;
  if (plan.stream) name = pbForceStreamName(S);
  for (const FormRow &r : kForms)
    if (!plan.stream && r.form.flat == (plan.kind != 0) && r.form.lanes_per_bot == plan.form &&
        (r.form.attraction_sums != 0) == plan.asum && (r.form.offsets64 != 0) == plan.big)
      name = r.name[S->payload ? 1 : 0]();
  if (name.empty() || name.size() + 1 > cap) return PB_ERR_ARG;
  memcpy(buf, name.c_str(), name.size() + 1);
  return PB_OK;
}

int pbSimSelectForceForm(pbSim *S, int index) {
  if (!S || index < -1 || index >= kNumForms) return PB_ERR_ARG;
  if (index < 0) {  // back to the automatic choice
    S->variant = 2;
    S->lanesPerBot = 0;
    S->forceSums = 0;
    S->wideOffsets = false;
    S->resident = 0;
    return PB_OK;
  }
  const pbForceForm &f = kForms[index].form;
  if (!f.attraction_sums && S->anyConstrained) {
    pbLastError() = "pbSimSelectForceForm: a member reads absForce_a (constrained_contraction); the dead-sum forms cannot run";
    return PB_ERR_ARG;
  }
  S->variant = f.flat ? 2 : 0;
  S->lanesPerBot = f.lanes_per_bot;
  S->forceSums = f.attraction_sums;
  S->wideOffsets = f.offsets64 != 0;
  S->resident = 1;  // a per-step form was asked for
  return PB_OK;
}

#ifdef PB_TIMELINE
int pbDebugSetTimeline(unsigned long long *deviceBuffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(pbTimelineBuf), &deviceBuffer, sizeof deviceBuffer) == hipSuccess ? PB_OK : PB_ERR_HIP;
}
#endif

}  // extern "C"
