// pb_engine.hpp -- what the translation units of the pbSim engine share (not part of the public C-ABI):
// the batch object, the launch plan, and the launchers each kernel family exports.
//
//   pb_engine.hip    the object, the schedule (stepMany), re-sort / phase update / I/O kernels, the C-ABI
//   pb_force.hip     k_force: the exact per-step force kernel in all its forms + the forms table
//   pb_stream.hip    k_force_stream: the opt-in streamlined (tolerance) force kernel
//   pb_resident.hip  k_resident: the multi-step one-workgroup-per-simulation kernel
//   pb_selftest.hip  exhaustive / sampled on-device proofs of the fast exact math, shader-clock sampler
//   pb_sweep.hpp     the neighbour sweep (device code shared by k_force and k_resident)
#pragma once

#include <string>
#include <vector>

#include "particlebot_hip.h"
#include "pb_device.hpp"
#include "pb_internal.hpp"

// thread-local text behind pbGetLastErrorString (defined in pb_engine.hip)
std::string &pbLastError();

#define PB_TRY(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      pbLastError() = std::string(hipGetErrorName(e_)) + " at " + __FILE__ + ":" + std::to_string(__LINE__) + \
                      " in " #expr;                                                                   \
      return PB_ERR_HIP;                                                                              \
    }                                                                                                 \
  } while (0)

#ifndef PB_TILE
#define PB_TILE 256
#endif
constexpr int TILE = PB_TILE;

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

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
  bool xcdMembers = true;   // batches of >= 8 small members: all tiles of a member on one XCD (PB_XCD_MEMBERS=0 under PB_ALLOW_ENV_OVERRIDES: the plain (tile, member) grid, for A/B)
  bool xcdTilesAll = true;   // the XCD-contiguous tile order for the multi-lane forms too (PB_XCD_TILES_ALL=0 under PB_ALLOW_ENV_OVERRIDES: only for one bot per lane, as rounds 1-4)
  bool magOk = false;   // every member passes pbAttractionMagnitudeSafe (fast path of the both-sums throughput form)
  int variant = 2;  // force kernel: 0 reference-shaped branches, 1 branch-free, 2 (default) + fast exact math
  int resident = 0;     // 0 automatic, 1 never, 2 whenever the simulation fits one workgroup (n <= 1024)
  int lanesPerBot = 0;  // lanes per bot of the per-step force kernel: 0 automatic; 1 (throughput form), 2, 4, 8, 16
  // k_force_stream's neighbour walk (pb_stream.hip, WALK): -1 chosen at every re-sort from the batch's cell lists
  // (chooseStreamWalk), 0 row by row, 1 flattened (pbSimSetStreamWalk).  Bit-identical either way.
  int streamWalk = -1;
  bool streamWalkAuto = false;
  unsigned long long *walkTrips = nullptr, walkTripsHost[2] = {0, 0};
  bool wideOffsets = false;  // run the 64-bit-offset throughput sweep on a batch below 2^28 bots (pbSimSelectForceForm, tests)
  unsigned debugLdsBytes = 0;  // PB_DEBUG_LDS_BYTES under PB_ALLOW_ENV_OVERRIDES=1 (tools/occupancy_sweep.py --lds)
  int rng = 0;          // phase noise: 0 PB-RNG v1 (counter based), 1 cuRAND-compatible XORWOW (pb_xorwow.hpp)
  int minDistanceMode = 0;  // phase update: 0 device min of squares + host root, 1 the reference's host loop (pbSimSetMinDistanceMode)
  std::vector<float> hostPos;  // mode 1: one simulation's positions, original order
  int forceSums = 0;    // 0: Sum|F_attr| only when a member reads it (constrained_contraction), 1: always
  bool anyConstrained = false;  // some member has constrained_contraction != 0
  pbSimStats stats{};
};

static inline dim3 gridOf(const pbSim *S) { return dim3(cdiv(S->n, TILE), S->nsims); }

// HIP's current device is per host thread (a new thread starts on device 0): a caller that drives
// several batches from several threads must not have to remember that
static inline void useDevice(const pbSim *S) { (void)hipSetDevice(S->device); }

// absForce_a has a reader (impl.cuh:167-169) or the caller asked for it (pbSimSetForceSums)
static inline bool pbStreamWalk(const pbSim *S) { return S->streamWalk >= 0 ? S->streamWalk == 1 : S->streamWalkAuto; }
static inline bool attractionSumsKept(const pbSim *S) { return S->forceSums != 0 || S->anyConstrained; }

// What a per-step force launch of this batch will be: the streamlined kernel or an exact one
// (kind 0 reference-shaped branches, 1 branch-free, 2 branch-free + fast exact math), and the lanes
// per bot of the exact branch-free kernels.  One place decides (pbForcePlan, pb_force.hip), so
// pbSimGetConfig reports what runs.
struct PbForcePlan {
  bool stream;
  int kind;
  int form;   // lanes per bot (1 = throughput form)
  bool asum;  // the launch maintains absForce_a (false: dead-sum form)
  bool big;   // 64-bit byte offsets in the throughput sweep
};
PbForcePlan pbForcePlan(const pbSim *S);

// step n's forces + kick into the other copy of posrad/vel; fuse: also step n+1's radius + integration
void pbLaunchForce(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext);          // pb_force.hip
void pbLaunchForceStream(pbSim *S, bool fuse, int c, int o, float dt, float tNext, int doRadiusNext);    // pb_stream.hip
// profiler-style names of the kernels above ("k_force<...>(argument types)"): pbSimForceKernelName
std::string pbKernelArgs(const char *mangledPointerType);                                                // pb_force.hip
std::string pbForceStreamName(const pbSim *S);                                                           // pb_stream.hip
// m whole timesteps from time t0 in one launch (simulations of <= 1024 bots)
bool pbResidentWanted(const pbSim *S);                                                                    // pb_resident.hip
void pbLaunchResident(pbSim *S, float dt, float t0, int m, int lightWave);
