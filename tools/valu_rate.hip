// VALU issue-rate microbenchmark for gfx950: nanoseconds of SIMD time per wave64 instruction for
// simple fp32 ops (fma, mul, compare/select) and for the transcendentals the force kernels use
// (v_rcp_f32, v_sqrt_f32, v_rsq_f32), at 1, 2, 4 and 8 waves per SIMD with eight independent chains
// per lane.  The 8-waves/SIMD figures are the "VALU roofline" DESIGN.md section 3 prices the force
// kernel's instruction mix against.  Prints one JSON line at the end.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed, unsigned long long *stamps) {
  // shader cycles (s_memtime) and 100 MHz real time (s_memrealtime) around the loop of one wave: the
  // clock the chip holds under THIS load = dcycles / dreal * 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
  const bool stamper = stamps && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long c0 = 0, r0 = 0;
  if (stamper) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  float a[8];
  // multiplier and addend in registers, as the force kernel's operands are (a literal-carrying
  // v_fmaak/v_fmamk encoding measured 3.4 cycles instead of ~2.3)
  const float mulc = 1.0f + seed * 1e-7f, addc = seed * 1e-7f;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = __builtin_fmaf(a[i], mulc, addc);
      if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
      if (OP == 3) a[i] = __builtin_amdgcn_rsqf(a[i]);
      if (OP == 4) a[i] = a[i] * mulc;
      if (OP == 5) a[i] = a[i] > 1.5f ? a[i] - 0.5f : a[i] + 0.25f;  // cmp + 2 alu + cndmask
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (stamper) {
    stamps[0] = __builtin_amdgcn_s_memtime() - c0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}
static double g_last_ns = 0, g_last_mhz = 0;
static unsigned long long *g_stamps = nullptr;
template <int OP>
void run(const char *name, float *d, int blocks, int per_iter_instr) {
  const int iters = 200000;  // ~12 ms per run at 8 waves/SIMD: long enough for the clock to settle
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, (unsigned long long *)nullptr);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, g_stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long st[2] = {0, 1};
  hipMemcpy(st, g_stamps, sizeof st, hipMemcpyDeviceToHost);
  g_last_mhz = (double)st[0] / (double)st[1] * 100.0;
  // waves per SIMD = blocks*4/(256*4); instr per SIMD = waves_per_simd * iters * per_iter_instr
  const double wavesPerSimd = blocks * 4.0 / 1024.0;
  const double instr = wavesPerSimd * iters * (double)per_iter_instr;
  g_last_ns = ms * 1e6 / instr;
  printf("%-10s blocks %5d (%.0f waves/SIMD): %.3f ms -> %.2f ns per wave-instruction per SIMD at %.0f MHz in-kernel = %.2f cycles\n",
         name, blocks, wavesPerSimd, ms, g_last_ns, g_last_mhz, g_last_ns * g_last_mhz * 1e-3);
}
int main() {
  float *d; hipMalloc(&d, 4096 * 256 * 4);
  hipMalloc(&g_stamps, 2 * sizeof(unsigned long long));
  double fma8 = 0, mul8 = 0, rcp8 = 0, sqrt8 = 0, rsq8 = 0, sel8 = 0;
  double fmaM = 0, mulM = 0, rcpM = 0, sqrtM = 0, rsqM = 0, selM = 0;
  for (int blocks : {256, 512, 1024, 2048}) {
    run<0>("fma", d, blocks, 8), fma8 = g_last_ns, fmaM = g_last_mhz;
    run<4>("mul", d, blocks, 8), mul8 = g_last_ns, mulM = g_last_mhz;
    run<1>("rcp", d, blocks, 8), rcp8 = g_last_ns, rcpM = g_last_mhz;
    run<2>("sqrt", d, blocks, 8), sqrt8 = g_last_ns, sqrtM = g_last_mhz;
    run<3>("rsq", d, blocks, 8), rsq8 = g_last_ns, rsqM = g_last_mhz;
    run<5>("cmp/sel", d, blocks, 32), sel8 = g_last_ns, selM = g_last_mhz;
  }
  // last pass = 8 waves per SIMD
  printf("{\"waves_per_simd\": 8, \"ns_per_wave_instr\": {\"fma\": %.4f, \"mul\": %.4f, \"cmp_sel_mix\": %.4f, "
         "\"rcp\": %.4f, \"sqrt\": %.4f, \"rsq\": %.4f}, "
         "\"in_kernel_mhz\": {\"fma\": %.1f, \"mul\": %.1f, \"cmp_sel_mix\": %.1f, \"rcp\": %.1f, \"sqrt\": %.1f, "
         "\"rsq\": %.1f}, "
         "\"cycles_per_wave_instr\": {\"fma\": %.3f, \"mul\": %.3f, \"cmp_sel_mix\": %.3f, \"rcp\": %.3f, "
         "\"sqrt\": %.3f, \"rsq\": %.3f}}\n",
         fma8, mul8, sel8, rcp8, sqrt8, rsq8, fmaM, mulM, selM, rcpM, sqrtM, rsqM, fma8 * fmaM * 1e-3,
         mul8 * mulM * 1e-3, sel8 * selM * 1e-3, rcp8 * rcpM * 1e-3, sqrt8 * sqrtM * 1e-3, rsq8 * rsqM * 1e-3);
  return 0;
}
