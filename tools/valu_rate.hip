// VALU issue-rate microbenchmark for gfx950: nanoseconds of SIMD time per wave64 instruction for
// simple fp32 ops (fma, mul, compare/select) and for the transcendentals the force kernels use
// (v_rcp_f32, v_sqrt_f32, v_rsq_f32), at 1, 2, 4 and 8 waves per SIMD with eight independent chains
// per lane.  The 8-waves/SIMD figures are the "VALU roofline" DESIGN.md section 5 prices the force
// kernel's instruction mix against.  Prints one JSON line at the end.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0000001f, 1e-7f);
      if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
      if (OP == 3) a[i] = __builtin_amdgcn_rsqf(a[i]);
      if (OP == 4) a[i] = a[i] * 1.0000001f;
      if (OP == 5) a[i] = a[i] > 1.5f ? a[i] - 0.5f : a[i] + 0.25f;  // cmp + 2 alu + cndmask
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static double g_last_ns = 0;
template <int OP>
void run(const char *name, float *d, int blocks, int per_iter_instr) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // waves per SIMD = blocks*4/(256*4); instr per SIMD = waves_per_simd * iters * per_iter_instr
  const double wavesPerSimd = blocks * 4.0 / 1024.0;
  const double instr = wavesPerSimd * iters * (double)per_iter_instr;
  g_last_ns = ms * 1e6 / instr;
  printf("%-10s blocks %5d (%.0f waves/SIMD): %.3f ms -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, blocks, wavesPerSimd, ms, ms * 1e6 / instr);
}
int main() {
  float *d; hipMalloc(&d, 4096 * 256 * 4);
  double fma8 = 0, mul8 = 0, rcp8 = 0, sqrt8 = 0, rsq8 = 0, sel8 = 0;
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0>("fma", d, blocks, 8), fma8 = g_last_ns;
    run<4>("mul", d, blocks, 8), mul8 = g_last_ns;
    run<1>("rcp", d, blocks, 8), rcp8 = g_last_ns;
    run<2>("sqrt", d, blocks, 8), sqrt8 = g_last_ns;
    run<3>("rsq", d, blocks, 8), rsq8 = g_last_ns;
    run<5>("cmp/sel", d, blocks, 32), sel8 = g_last_ns;
  }
  // last pass = 8 waves per SIMD
  printf("{\"waves_per_simd\": 8, \"ns_per_wave_instr\": {\"fma\": %.4f, \"mul\": %.4f, \"cmp_sel_mix\": %.4f, "
         "\"rcp\": %.4f, \"sqrt\": %.4f, \"rsq\": %.4f}}\n", fma8, mul8, sel8, rcp8, sqrt8, rsq8);
  return 0;
}
