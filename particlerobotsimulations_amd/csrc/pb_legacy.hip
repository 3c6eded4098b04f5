// pb_legacy.hip -- the reference's `extern "C"` device boundary on gfx950.
//
// Every entry point of particlebot.cuh:15-121 with the same argument meaning, one global parameter
// block, default stream, caller-owned buffers, abort-on-error.  One hand-written
// kernel per reference kernel; the per-bot arithmetic lives in pb_device.hpp and is shared with
// the fused engine (pb_engine.hip), so both paths are bit-identical to the oracle.
//
// This is the compatibility seam.  The fast path is pbSim* (pb_engine.hip).
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "particlebot_hip.h"
#include "pb_device.hpp"
#include "pb_internal.hpp"
#include "pb_xorwow.hpp"

const uint32_t *pbXorwowDeviceTable(hipError_t *err) {
  // one copy per DEVICE (a process may drive batches on several GPUs, pbSim::device): keyed by the
  // calling thread's current device, which every pbSim entry point has set (useDevice)
  static std::mutex mu;
  static std::map<int, uint32_t *> perDevice;
  static std::vector<uint32_t> table;  // built once on the host
  std::lock_guard<std::mutex> lock(mu);
  if (err) *err = hipSuccess;
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) {
    if (err) *err = e;
    return nullptr;
  }
  auto it = perDevice.find(device);
  if (it != perDevice.end()) return it->second;
  if (table.empty()) {
    table.resize(PB_XW_TABLE_WORDS);
    pbXorwowBuildJumpTable(table.data());
  }
  uint32_t *d = nullptr;
  e = hipMalloc((void **)&d, sizeof(uint32_t) * PB_XW_TABLE_WORDS);
  if (e == hipSuccess) e = hipMemcpy(d, table.data(), sizeof(uint32_t) * PB_XW_TABLE_WORDS, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (err) *err = e;
    if (d) (void)hipFree(d);
    return nullptr;
  }
  perDevice[device] = d;
  return d;
}

namespace {

PbDevParams g_P;  // the reference's single `__constant__ SimParams params` (impl.cuh:27)
SimParams g_hostParams;
bool g_haveParams = false;
int g_rngKind = PB_RNG_COUNTER;  // generator curand_setup() initialises (pbSetRngKind)
float g_wallHalf = 64.0f;

struct Buffer {
  void *dev;
  size_t size;
};
std::map<uint, Buffer> g_buffers;  // headless "GL buffer objects"
uint g_nextBuffer = 1;
std::mutex g_mu;

// grow-only scratch for sortParticlebots
uint32_t *g_sortKeys = nullptr, *g_sortVals = nullptr, *g_sortHist = nullptr;
size_t g_sortCap = 0, g_sortHistCap = 0;

inline dim3 gridFor(uint32_t n, int block) { return dim3((n + block - 1) / block); }

// ---- kernels ----------------------------------------------------------------------------------

// impl.cuh:53-103 via thrust::for_each (particlebot_cuda.cu:145-160)
__global__ __launch_bounds__(256) void k_integrate(PbDevParams P, float2 *__restrict__ pos,
                                                   float2 *__restrict__ vel, const float *__restrict__ rad,
                                                   float dt, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  float2 p = pos[i], v = vel[i];
  pbIntegrate(P, p.x, p.y, v.x, v.y, rad[i], dt);
  pos[i] = p;
  vel[i] = v;
}

// impl.cuh:446-465
__global__ __launch_bounds__(256) void k_calc_hash(PbDevParams P, uint32_t *__restrict__ hash,
                                                   uint32_t *__restrict__ index, const float2 *__restrict__ pos,
                                                   uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float2 p = pos[i];
  hash[i] = pbHash(P, pbCellX(P, p.x), pbCellY(P, p.y));
  index[i] = i;
}

// impl.cuh:469-538.  The reference stages the previous thread's hash through shared memory; on a
// 64-wide wave the neighbour's hash comes from a lane shuffle, and only lane 0 of each wave reads
// hash[i-1] from memory.
__global__ __launch_bounds__(256) void k_reorder(uint32_t *__restrict__ cellStart, uint32_t *__restrict__ cellEnd,
                                                 float2 *__restrict__ sortedPos, float2 *__restrict__ sortedVel,
                                                 float *__restrict__ sortedRad, const uint32_t *__restrict__ hash,
                                                 const uint32_t *__restrict__ index, const float2 *__restrict__ oldPos,
                                                 const float2 *__restrict__ oldVel, const float *__restrict__ oldRad,
                                                 uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool ok = i < n;
  const uint32_t h = ok ? hash[i] : 0u;
  uint32_t prev = __shfl_up(h, 1, 64);
  if ((threadIdx.x & 63u) == 0u && ok && i > 0) prev = hash[i - 1];
  if (!ok) return;
  if (i == 0 || h != prev) {
    cellStart[h] = i;
    if (i > 0) cellEnd[prev] = i;
  }
  if (i == n - 1) cellEnd[h] = i + 1;
  const uint32_t src = index[i];
  sortedRad[i] = oldRad[src];
  sortedPos[i] = oldPos[src];
  sortedVel[i] = oldVel[src];
}

// impl.cuh:124-181
__global__ __launch_bounds__(256) void k_update_rad(PbDevParams P, const float *__restrict__ absA,
                                                    const float *__restrict__ absR, float *__restrict__ rad,
                                                    const float *__restrict__ phase, float time, float dt,
                                                    const int *__restrict__ dead, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  rad[i] = pbActuate(P, rad[i], phase[i], dead[i], absA[i], absR[i], time, dt);
}

// impl.cuh:264-290
__global__ __launch_bounds__(256) void k_update_phase(PbDevParams P, const float2 *__restrict__ pos,
                                                      float *__restrict__ phase, float spacing, float min_d,
                                                      uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float2 p = pos[i];
  phase[i] = pbPhase(P, p.x, p.y, spacing, min_d, phase[i]);
}

// impl.cuh:36-41.  Counter generator: state = (seed, draws so far).  XORWOW kinds: curand_init(seed, i, 0).
__global__ __launch_bounds__(256) void k_rng_setup(pbRngState *__restrict__ st, uint32_t seed, uint32_t n, int kind,
                                                   const uint32_t *__restrict__ jump) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  pbRngState s;
  if (kind == PB_RNG_COUNTER) {
    s.d = seed;
    s.v[0] = s.v[1] = s.v[2] = s.v[3] = s.v[4] = 0u;
    s.boxmuller_flag = 0;
    s.kind = PB_RNG_COUNTER;
    s.boxmuller_extra = 0.0f;
    s.reserved[0] = s.reserved[1] = s.reserved[2] = 0.0f;
  } else {
    pbXorwowSeed(s, (uint64_t)seed, kind);
    pbXorwowSkipSubsequences(s, i, jump);
  }
  st[i] = s;
}

// impl.cuh:43-51
__global__ __launch_bounds__(256) void k_add_noise(pbRngState *__restrict__ st, float *__restrict__ val, float std,
                                                   uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  pbRngState s = st[i];
  float noise;
  if (s.kind == PB_RNG_COUNTER) {
    noise = std * pbNormal(s.d, i, s.v[0]);
    s.v[0] += 1;
  } else {
    noise = std * pbXorwowNormal(s);
  }
  val[i] += noise;
  st[i] = s;
}

// impl.cuh:657-831 with collideCell (:597-653) inlined.  One bot per lane over the sorted arrays,
// 25 cells in the reference's order (y outer, x inner), bots of a cell in ascending sorted index.
// The pair arithmetic is the engine's branch-free form (pbPairEvalK), with the fast exact
// sqrt/division forms where their domain holds (FASTOK from pbFastMathAllowed at setParameters time,
// and every lane of the wave away from the axes): bit-identical to the reference-shaped pbPair,
// about 1.5x faster.
template <bool FASTOK>
__global__ __launch_bounds__(256) void k_collide(PbDevParams P, float2 *__restrict__ newVel,
                                                 float *__restrict__ absA, float *__restrict__ absR,
                                                 const float2 *__restrict__ sPos, const float2 *__restrict__ sVel,
                                                 const float *__restrict__ sRad, const uint32_t *__restrict__ index,
                                                 const uint32_t *__restrict__ cellStart,
                                                 const uint32_t *__restrict__ cellEnd, uint32_t n, float dt) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float2 p = sPos[i];
  float2 v = sVel[i];
  const float rad = sRad[i];
  const int gx = pbCellX(P, p.x), gy = pbCellY(P, p.y);
  const uint32_t orig = index[i];
  const bool payloadMode = (P.nDead == -1);
  const uint32_t payloadIdx = P.nCells - 1u;
  const bool selfPayload = payloadMode && orig == payloadIdx;
  const float att1 = selfPayload ? P.attractionFactor : 1.0f;
  const float slope0 = pbBandSlope(P.attraction);
  const PbContactK CK{P.spring, P.damping, P.shear};
  PbForce F;
  F.fx = 0.0f;
  F.fy = 0.0f;
  F.fa = 0.0f;
  F.fr = 0.0f * absR[orig];  // impl.cuh:688
  auto sweep = [&](auto fastTag) {
    constexpr bool FAST = decltype(fastTag)::value;
    for (int y = -2; y <= 2; y++) {
      for (int x = -2; x <= 2; x++) {
        const uint32_t h = pbHash(P, gx + x, gy + y);
        const uint32_t start = cellStart[h];
        if (start == 0xffffffffu) continue;
        const uint32_t end = cellEnd[h];
        for (uint32_t j = start; j < end; j++) {
          float att2 = 1.0f;
          if (payloadMode && index[j] == payloadIdx) att2 = P.attractionFactor;
          const float2 q = sPos[j];
          const bool live[1] = {j != i};
          const float bx[1] = {q.x}, by[1] = {q.y}, rb[1] = {sRad[j]};
          const float A[1] = {P.attraction * att2 * att1};
          const float K[1] = {payloadMode ? pbBandSlope(A[0]) : slope0};
          PbPairTerm t[1];
          pbPairEvalK<FAST, 1>(CK, live, p.x, p.y, v.x, v.y, rad, bx, by, rb, A, K, [&](int) { return sVel[j]; }, t);
          pbPairAdd(live[0], t[0], F);
        }
      }
    }
  };
  if (FASTOK && __all(pbLaneFastMathOk(p.x, p.y))) sweep(std::true_type{});
  else sweep(std::false_type{});
  pbObstacles(P, p.x, p.y, v.x, v.y, rad, F);
  pbFrictionAndKick(P, selfPayload, F.fx, F.fy, dt, v.x, v.y);
  newVel[orig] = v;
  absA[orig] = F.fa;
  absR[orig] = F.fr;
}

void requireParams(const char *who) {
  if (!g_haveParams) {
    fprintf(stderr, "%s: setParameters() has not been called\n", who);
    exit(EXIT_FAILURE);
  }
}

}  // namespace

extern "C" {

void cudaInit(int, char **) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    printf("No HIP capable devices found, exiting...\n");
    exit(EXIT_SUCCESS);  // particlebot_cuda.cu:36-40 exits with EXIT_SUCCESS here
  }
  PB_CHECK_ABORT(hipSetDevice(0));
}

void cudaGLInit(int argc, char **argv) { cudaInit(argc, argv); }

void allocateArray(void **devPtr, size_t size) { PB_CHECK_ABORT(hipMalloc(devPtr, size ? size : 1)); }

void freeArray(void *devPtr) { PB_CHECK_ABORT(hipFree(devPtr)); }

void threadSync(void) { PB_CHECK_ABORT(hipDeviceSynchronize()); }

void copyArrayToDevice(void *device, const void *host, int offset, int size) {
  PB_CHECK_ABORT(hipMemcpy((char *)device + offset, host, (size_t)size, hipMemcpyHostToDevice));
}

struct pbGraphicsResource {
  uint vbo;
};

void registerGLBufferObject(uint vbo, struct pbGraphicsResource **res) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_buffers.count(vbo)) {
    fprintf(stderr, "registerGLBufferObject: unknown buffer object %u\n", vbo);
    exit(EXIT_FAILURE);
  }
  *res = new pbGraphicsResource{vbo};
}

void unregisterGLBufferObject(struct pbGraphicsResource *res) { delete res; }

void *mapGLBufferObject(struct pbGraphicsResource **res) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_buffers.find((*res)->vbo);
  if (it == g_buffers.end()) {
    fprintf(stderr, "mapGLBufferObject: buffer object %u was deleted\n", (*res)->vbo);
    exit(EXIT_FAILURE);
  }
  return it->second.dev;
}

void unmapGLBufferObject(struct pbGraphicsResource *) {}

void copyArrayFromDevice(void *host, const void *device, struct pbGraphicsResource **res, int size) {
  if (res) device = mapGLBufferObject(res);
  PB_CHECK_ABORT(hipMemcpy(host, device, (size_t)size, hipMemcpyDeviceToHost));
  if (res) unmapGLBufferObject(*res);
}

uint pbCreateBuffer(size_t size) {
  void *d = nullptr;
  PB_CHECK_ABORT(hipMalloc(&d, size ? size : 1));
  PB_CHECK_ABORT(hipMemset(d, 0, size ? size : 1));
  std::lock_guard<std::mutex> lk(g_mu);
  const uint id = g_nextBuffer++;
  g_buffers[id] = Buffer{d, size};
  return id;
}

void pbBufferSubData(uint vbo, size_t offset, size_t size, const void *data) {
  Buffer b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_buffers.find(vbo);
    if (it == g_buffers.end() || offset + size > it->second.size) {
      fprintf(stderr, "pbBufferSubData: bad buffer object %u or range\n", vbo);
      exit(EXIT_FAILURE);
    }
    b = it->second;
  }
  PB_CHECK_ABORT(hipMemcpy((char *)b.dev + offset, data, size, hipMemcpyHostToDevice));
}

void pbDeleteBuffer(uint vbo) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_buffers.find(vbo);
  if (it == g_buffers.end()) return;
  PB_CHECK_ABORT(hipFree(it->second.dev));
  g_buffers.erase(it);
}

void setParameters(SimParams *hostParams) {
  g_hostParams = *hostParams;
  pbFlattenParams(g_P, *hostParams, g_wallHalf);
  g_haveParams = true;
}

void pbSetWallHalfExtent(float half) {
  g_wallHalf = half > 0.0f ? half : 64.0f;
  if (g_haveParams) g_P.wallHalf = g_wallHalf;
}

void integrateSystem(float *pos, float *vel, float *rad, float deltaTime, uint nCells, float) {
  requireParams("integrateSystem");
  if (!nCells) return;
  hipLaunchKernelGGL(k_integrate, gridFor(nCells, 256), dim3(256), 0, 0, g_P, (float2 *)pos, (float2 *)vel, rad,
                     deltaTime, nCells);
  PB_CHECK_ABORT(hipGetLastError());
}

void calcHash(uint *hash, uint *index, float *pos, int nCells) {
  requireParams("calcHash");
  if (nCells <= 0) return;
  hipLaunchKernelGGL(k_calc_hash, gridFor(nCells, 256), dim3(256), 0, 0, g_P, hash, index, (const float2 *)pos,
                     (uint32_t)nCells);
  PB_CHECK_ABORT(hipGetLastError());
}

void reorderDataAndFindCellStart(uint *cellStart, uint *cellEnd, float *sortedPos, float *sortedVel,
                                 float *sortedRad, uint *hash, uint *index, float *oldPos, float *oldVel,
                                 float *oldRad, uint nCells, uint numCells) {
  // particlebot_cuda.cu:301: every cell empty; cellEnd is deliberately left alone
  PB_CHECK_ABORT(hipMemsetAsync(cellStart, 0xff, (size_t)numCells * sizeof(uint), 0));
  if (!nCells) return;
  hipLaunchKernelGGL(k_reorder, gridFor(nCells, 256), dim3(256), 0, 0, cellStart, cellEnd, (float2 *)sortedPos,
                     (float2 *)sortedVel, sortedRad, hash, index, (const float2 *)oldPos, (const float2 *)oldVel,
                     oldRad, nCells);
  PB_CHECK_ABORT(hipGetLastError());
}

void updateRad_light_wave(float *, float *absForce_a, float *absForce_r, float *rad, float *phase, float time,
                          float deltaTime, int *dead, int nCells) {
  requireParams("updateRad_light_wave");
  if (nCells <= 0) return;
  hipLaunchKernelGGL(k_update_rad, gridFor(nCells, 256), dim3(256), 0, 0, g_P, absForce_a, absForce_r, rad, phase,
                     time, deltaTime, dead, (uint32_t)nCells);
  PB_CHECK_ABORT(hipGetLastError());
}

int pbSetRngKind(int kind) {
  if (kind != PB_RNG_COUNTER && kind != PB_RNG_XORWOW_CURAND && kind != PB_RNG_XORWOW_ROCRAND) return 1;
  g_rngKind = kind;
  return 0;
}

int pbGetRngKind(void) { return g_rngKind; }

void curand_setup(pbRngState *state, int N) {
  requireParams("curand_setup");
  if (N <= 0) return;
  const uint32_t *jump = nullptr;
  if (g_rngKind != PB_RNG_COUNTER) {
    hipError_t e = hipSuccess;
    jump = pbXorwowDeviceTable(&e);
    PB_CHECK_ABORT(e);
  }
  hipLaunchKernelGGL(k_rng_setup, gridFor(N, 256), dim3(256), 0, 0, state, g_P.seed, (uint32_t)N, g_rngKind, jump);
  PB_CHECK_ABORT(hipGetLastError());
}

void add_normal_noise(pbRngState *state, float *val, float std, int N) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_add_noise, gridFor(N, 256), dim3(256), 0, 0, state, val, std, (uint32_t)N);
  PB_CHECK_ABORT(hipGetLastError());
}

void updatePhase(float *pos, float *phase, float spacing, float, float min_d, int nCells) {
  requireParams("updatePhase");
  if (nCells <= 0) return;
  hipLaunchKernelGGL(k_update_phase, gridFor(nCells, 256), dim3(256), 0, 0, g_P, (const float2 *)pos, phase, spacing,
                     min_d, (uint32_t)nCells);
  PB_CHECK_ABORT(hipGetLastError());
}

void updateCol(float *, float *, int, float *, float *, int *) {}  // display only

void collide(float *newVel, float *absForce_a, float *absForce_r, float *sortedPos, float *sortedVel,
             float *sortedRad, uint *index, uint *cellStart, uint *cellEnd, uint nCells, uint, float deltaTime) {
  requireParams("collide");
  if (!nCells) return;
  if (pbFastMathAllowed(g_P))
    hipLaunchKernelGGL(k_collide<true>, gridFor(nCells, 256), dim3(256), 0, 0, g_P, (float2 *)newVel, absForce_a,
                       absForce_r, (const float2 *)sortedPos, (const float2 *)sortedVel, sortedRad, index,
                       cellStart, cellEnd, nCells, deltaTime);
  else
    hipLaunchKernelGGL(k_collide<false>, gridFor(nCells, 256), dim3(256), 0, 0, g_P, (float2 *)newVel, absForce_a,
                       absForce_r, (const float2 *)sortedPos, (const float2 *)sortedVel, sortedRad, index,
                       cellStart, cellEnd, nCells, deltaTime);
  PB_CHECK_ABORT(hipGetLastError());
}

void calcCOG(float *, float *, float *, int, float, int, float) {}  // display only

void sortParticlebots(uint *hash, uint *index, uint nCells) {
  if (!nCells) return;
  if (nCells > g_sortCap) {
    if (g_sortKeys) PB_CHECK_ABORT(hipFree(g_sortKeys));
    if (g_sortVals) PB_CHECK_ABORT(hipFree(g_sortVals));
    PB_CHECK_ABORT(hipMalloc((void **)&g_sortKeys, sizeof(uint32_t) * nCells));
    PB_CHECK_ABORT(hipMalloc((void **)&g_sortVals, sizeof(uint32_t) * nCells));
    g_sortCap = nCells;
  }
  const size_t he = pbSortHistEntries(nCells);
  if (he > g_sortHistCap) {
    if (g_sortHist) PB_CHECK_ABORT(hipFree(g_sortHist));
    PB_CHECK_ABORT(hipMalloc((void **)&g_sortHist, sizeof(uint32_t) * he));
    g_sortHistCap = he;
  }
  // keys are cell hashes < numCells when parameters are known; otherwise sort all 32 bits
  const int bits = g_haveParams ? pbKeyBits(g_P.numCells) : 32;
  hipError_t e;
  const int where = pbRadixSortPairs(hash, index, g_sortKeys, g_sortVals, g_sortHist, nCells, bits, 0, &e);
  if (where < 0) PB_CHECK_ABORT(e);
  if (where == 1) {
    PB_CHECK_ABORT(hipMemcpyAsync(hash, g_sortKeys, sizeof(uint32_t) * nCells, hipMemcpyDeviceToDevice, 0));
    PB_CHECK_ABORT(hipMemcpyAsync(index, g_sortVals, sizeof(uint32_t) * nCells, hipMemcpyDeviceToDevice, 0));
  }
}

}  // extern "C"
