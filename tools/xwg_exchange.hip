// xwg_exchange.hip -- what would one timestep's EXCHANGE cost if a small simulation were spread over G workgroups of a
// persistent kernel ("a resident kernel that gives one simulation 2-4 workgroups of one XCD with a
// flag barrier per step")?  Per step every workgroup publishes its share of the member's posrad + velocity (24 B per
// bot) to global memory with sc1 stores, arrives at a per-member monotonic counter (agent-scope atomic add after a
// release fence), polls it with sc1 loads until all G have arrived, and reads the whole member (n x 24 B) back with
// sc1 loads into LDS.  No physics at all: this is the floor such a kernel would pay per step ON TOP of 1/G of the
// one-workgroup resident kernel's compute.  Members are placed so that a member's G workgroups have equal
// blockIdx % 8 (one XCD under round-robin dispatch).
//   hipcc --offload-arch=gfx950 -O3 tools/xwg_exchange.hip -o tools/xwg_exchange && tools/xwg_exchange
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <int G>
__global__ __launch_bounds__(256) void k_exchange(float4 *__restrict__ pr, float2 *__restrict__ vel,
                                                  unsigned *__restrict__ counters, int n, int steps, int members,
                                                  float *__restrict__ sink) {
  __shared__ float4 sPr[1024];
  __shared__ float2 sVel[1024];
  // member m's workgroups are blocks m*8*? ... : give the G workgroups of a member equal blockIdx % 8
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;  // idx-th workgroup of this XCD
  const int member = (idx / G) * 8 + xcd, part = idx % G;
  if (member >= members) return;
  float4 *P = pr + (size_t)member * n;
  float2 *V = vel + (size_t)member * n;
  unsigned *ctr = counters + member * 32;  // (own 128-byte line)
  const int per = (n + G - 1) / G, lo = part * per, hi = min(n, lo + per);
  float acc = 0.0f;
  for (int s = 1; s <= steps; s++) {
    // publish my share (sc1: write through to L2)
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      __builtin_nontemporal_store(make_float4(acc + i, (float)s, 0.1f, 1.0f).x, &P[i].x);  // keep it simple: 4 + 2 dwords
      __hip_atomic_store(&P[i].y, (float)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&V[i].x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * s)) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // read the whole member back (sc1 loads: L2-served)
    for (int i = threadIdx.x; i < n; i += 256) {
      float4 q;
      q.x = __hip_atomic_load(&P[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q.y = __hip_atomic_load(&P[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q.z = __hip_atomic_load(&P[i].z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q.w = 1.0f;
      sPr[i] = q;
      sVel[i] = make_float2(__hip_atomic_load(&V[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.0f);
    }
    __syncthreads();
    acc += sPr[(threadIdx.x * 7) % n].y + sVel[threadIdx.x % n].x * 1e-9f;
    __syncthreads();  // (everybody done reading before the next step's stores overwrite the arrays)
    // a second barrier would be needed in a real kernel (ping-pong buffers avoid it); not counted here
  }
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = acc;
}

template <int G>
int run(int members, int n, int steps) {
  float4 *pr;
  float2 *vel;
  unsigned *ctr;
  float *sink;
  const int blocks = ((members + 7) / 8) * 8 * G;
  CHECK(hipMalloc((void **)&pr, sizeof(float4) * members * n));
  CHECK(hipMalloc((void **)&vel, sizeof(float2) * members * n));
  CHECK(hipMalloc((void **)&ctr, 128 * members));
  CHECK(hipMalloc((void **)&sink, sizeof(float) * blocks));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CHECK(hipMemset(ctr, 0, 128 * members));
    CHECK(hipMemset(pr, 0, sizeof(float4) * members * n));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_exchange<G>, dim3(blocks), dim3(256), 0, 0, pr, vel, ctr, n, steps, members, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;
  }
  printf("G = %d workgroups per member, %3d members x %4d bots: %.2f us per step (exchange only, %d workgroups)\n", G, members,
         n, best * 1e3f / steps, blocks);
  (void)hipFree(pr), (void)hipFree(vel), (void)hipFree(ctr), (void)hipFree(sink);
  return 0;
}

int main() {
  const int steps = 4000;
  for (int members : {1, 32, 64}) {
    for (int n : {201, 500, 1000}) {
      if (run<1>(members, n, steps)) return 1;
      if (run<2>(members, n, steps)) return 1;
      if (run<4>(members, n, steps)) return 1;
    }
  }
  return 0;
}
