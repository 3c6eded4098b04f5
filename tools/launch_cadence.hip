// launch_cadence.hip -- what a dependent kernel boundary costs on this GPU, eager against hipGraph replay.
// The per-step forms of small ensembles (BASELINE configs[3] at 32 + 32 members per GPU: one ~5 us kernel per step, each
// needing the one before it) are bound by this cadence; DESIGN.md section 6 / HISTORY.md "No hipGraph" argue that a
// replayed graph cannot shorten it.  This measures it.
//   hipcc --offload-arch=gfx950 -O3 -o launch_cadence tools/launch_cadence.hip && ./launch_cadence
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(e)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (e);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s at line %d\n", hipGetErrorString(e_), __LINE__);             \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

// one workgroup of 256 lanes per "member", `spin` dependent FMAs per lane, reads what the previous launch wrote
__global__ void k_step(const float *__restrict__ in, float *__restrict__ out, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = in[i];
  for (int k = 0; k < spin; k++) v = __builtin_fmaf(v, 1.0000001f, 1e-7f);
  out[i] = v;
}

int main() {
  const int members = 64, n = members * 256, launches = 4000, perGraph = 200;
  float *a, *b;
  CHECK(hipMalloc(&a, n * sizeof(float)));
  CHECK(hipMalloc(&b, n * sizeof(float)));
  CHECK(hipMemset(a, 0, n * sizeof(float)));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%d workgroups of 256 lanes per launch, %d dependent launches (ping-pong buffers)\n", members, launches);
  printf("%8s %14s %14s %14s\n", "spin", "kernel alone", "eager cadence", "graph cadence");
  for (int spin : {0, 200, 1000, 2500, 5000}) {
    // the kernel's own duration: one launch between events, best of 20
    float alone = 1e9f;
    for (int r = 0; r < 20; r++) {
      CHECK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_step, dim3(members), dim3(256), 0, s, a, b, spin);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < alone) alone = ms;
    }
    // eager: back-to-back launches on one stream
    float eager = 1e9f;
    for (int r = 0; r < 3; r++) {
      CHECK(hipEventRecord(e0, s));
      for (int l = 0; l < launches; l++)
        hipLaunchKernelGGL(k_step, dim3(members), dim3(256), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, spin);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < eager) eager = ms;
    }
    // graph: perGraph launches captured once, replayed launches / perGraph times
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int l = 0; l < perGraph; l++)
      hipLaunchKernelGGL(k_step, dim3(members), dim3(256), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, spin);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    float graph = 1e9f;
    for (int r = 0; r < 3; r++) {
      CHECK(hipEventRecord(e0, s));
      for (int l = 0; l < launches / perGraph; l++) CHECK(hipGraphLaunch(ge, s));
      CHECK(hipEventRecord(e1, s));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < graph) graph = ms;
    }
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    printf("%8d %11.2f us %11.2f us %11.2f us\n", spin, alone * 1e3f, eager * 1e3f / launches, graph * 1e3f / launches);
  }
  return 0;
}
