// pb_sort.hip -- stable LSD radix sort of (cell hash, bot index) pairs for gfx950.
//
// Replaces thrust::sort_by_key (particlebot_cuda.cu:377-382).  Stability matters: inside a cell
// the bots must stay in ascending original index because that fixes the fp32 summation order of
// the force loop.
//
// One pass = three kernels over tiles of PB_SORT_TILE pairs:
//   histogram : per-workgroup 256-bin digit histogram in LDS            -> hist[digit][block]
//   scan      : exclusive prefix sum over that digit-major table (3 small launches: chunk sums,
//               scan of the sums, rescan of each chunk)                  -> global base per (digit, block)
//   scatter   : each wave walks its 64-pair chunks in order; lanes holding the same digit find each
//               other with 8 wave ballots, rank themselves with a popcount of the lower lanes, and
//               a per-wave running counter in LDS carries the order from chunk to chunk and (after
//               an LDS prefix over the 4 waves) from wave to wave.  No global atomics, so the
//               output order is a pure function of the input.
#include "pb_internal.hpp"

namespace {

constexpr int WAVES = PB_SORT_THREADS / 64;
constexpr int CHUNKS = PB_SORT_ITEMS;  // 64-pair chunks per wave

__global__ __launch_bounds__(PB_SORT_THREADS) void k_sort_hist(const uint32_t *__restrict__ keys,
                                                               uint32_t *__restrict__ hist, uint32_t n,
                                                               int shift, uint32_t nblocks) {
  __shared__ uint32_t bins[256];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * PB_SORT_TILE;
#pragma unroll
  for (int k = 0; k < PB_SORT_ITEMS; k++) {
    const uint32_t i = base + k * PB_SORT_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&bins[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = bins[threadIdx.x];
}

// Exclusive scan of the digit-major histogram table (m entries) in three small launches: per-chunk
// sums, a one-workgroup scan of the chunk sums, then each chunk rescanned onto its base.  Inside a
// workgroup a thread owns SCAN_PER consecutive entries; thread sums are scanned with wave shuffles
// and the 4 wave totals go through LDS.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER;

// exclusive prefix of `sum` over the workgroup; *total receives the workgroup's total
__device__ __forceinline__ uint32_t blockExclusive(uint32_t sum, uint32_t *total) {
  __shared__ uint32_t waveSum[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(inc, d, 64);
    if (lane >= d) inc += up;
  }
  if (lane == 63) waveSum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    if (w < wave) base += waveSum[w];
    all += waveSum[w];
  }
  if (total) *total = all;
  __syncthreads();
  return base + inc - sum;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chunk_sums(const uint32_t *__restrict__ a, uint32_t m,
                                                                  uint32_t *__restrict__ chunkSum) {
  const uint32_t lo = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_PER;
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++)
    if (lo + k < m) sum += a[lo + k];
  uint32_t total;
  (void)blockExclusive(sum, &total);
  if (threadIdx.x == 0) chunkSum[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the nchunks chunk sums, in place
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_top(uint32_t *__restrict__ chunkSum, uint32_t nchunks) {
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nchunks; base += SCAN_THREADS) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nchunks ? chunkSum[i] : 0u;
    uint32_t total;
    const uint32_t ex = blockExclusive(v, &total);
    if (i < nchunks) chunkSum[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t *__restrict__ a, uint32_t m,
                                                             const uint32_t *__restrict__ chunkBase) {
  const uint32_t lo = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_PER;
  uint32_t v[SCAN_PER];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    v[k] = lo + k < m ? a[lo + k] : 0u;
    sum += v[k];
  }
  uint32_t run = chunkBase[blockIdx.x] + blockExclusive(sum, nullptr);
#pragma unroll
  for (int k = 0; k < SCAN_PER; k++) {
    if (lo + k < m) a[lo + k] = run;
    run += v[k];
  }
}

__global__ __launch_bounds__(PB_SORT_THREADS) void k_sort_scatter(
    const uint32_t *__restrict__ keysIn, const uint32_t *__restrict__ valsIn, uint32_t *__restrict__ keysOut,
    uint32_t *__restrict__ valsOut, const uint32_t *__restrict__ hist, uint32_t n, int shift, uint32_t nblocks) {
  __shared__ uint32_t cnt[WAVES][256];  // per-wave digit counts, then per-wave running bases
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = 0; w < WAVES; w++) cnt[w][threadIdx.x] = 0;
  __syncthreads();

  const uint32_t waveBase = blockIdx.x * PB_SORT_TILE + wave * (CHUNKS * 64);
  uint32_t key[CHUNKS], val[CHUNKS];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint32_t i = waveBase + c * 64 + lane;
    const bool ok = i < n;
    key[c] = ok ? keysIn[i] : 0u;
    val[c] = ok ? valsIn[i] : 0u;
    if (ok) atomicAdd(&cnt[wave][(key[c] >> shift) & 255u], 1u);
  }
  __syncthreads();
  {
    // thread d turns the 4 per-wave counts of digit d into 4 global write bases
    const uint32_t d = threadIdx.x;
    uint32_t run = hist[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
      const uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();

  volatile uint32_t *myBase = cnt[wave];
  const uint64_t lowerLanes = (1ull << lane) - 1ull;
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint32_t i = waveBase + c * 64 + lane;
    const bool ok = i < n;
    const uint32_t d = (key[c] >> shift) & 255u;
    uint64_t same = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint64_t m = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t rank = __popcll(same & lowerLanes);
    const uint32_t group = __popcll(same);
    uint32_t dst = 0;
    if (ok) dst = myBase[d] + rank;
    // every lane has read its base before the group leader advances it (one wave, in-order LDS)
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) myBase[d] = myBase[d] + group;
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      keysOut[dst] = key[c];
      valsOut[dst] = val[c];
    }
  }
}

}  // namespace

int pbRadixSortPairs(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                     uint32_t *hist, uint32_t n, int bits, hipStream_t stream, hipError_t *err) {
  if (err) *err = hipSuccess;
  if (n == 0) return 0;
  const uint32_t nblocks = pbSortBlocks(n);
  uint32_t *kin = keys, *vin = vals, *kout = keys_tmp, *vout = vals_tmp;
  int where = 0;
  for (int shift = 0; shift < bits; shift += 8) {
    hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(PB_SORT_THREADS), 0, stream, kin, hist, n, shift, nblocks);
    const uint32_t m = 256u * nblocks, nchunks = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint32_t *chunkSum = hist + m;  // the workspace has room for the chunk sums behind the table
    hipLaunchKernelGGL(k_scan_chunk_sums, dim3(nchunks), dim3(SCAN_THREADS), 0, stream, hist, m, chunkSum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_THREADS), 0, stream, chunkSum, nchunks);
    hipLaunchKernelGGL(k_scan_apply, dim3(nchunks), dim3(SCAN_THREADS), 0, stream, hist, m, chunkSum);
    hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(PB_SORT_THREADS), 0, stream, kin, vin, kout, vout,
                       hist, n, shift, nblocks);
    uint32_t *t = kin;
    kin = kout;
    kout = t;
    t = vin;
    vin = vout;
    vout = t;
    where ^= 1;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (err) *err = e;
    return -1;
  }
  return where;
}
