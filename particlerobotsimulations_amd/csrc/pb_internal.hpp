// pb_internal.hpp -- host-side declarations shared between the translation units of
// libparticlebot_hip.so (not part of the public C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- stable LSD radix sort of (key,value) pairs, 8 bits per pass (pb_sort.hip) ----------------
// Tile geometry of one sort workgroup.
constexpr int PB_SORT_THREADS = 256;
constexpr int PB_SORT_ITEMS = 8;
constexpr int PB_SORT_TILE = PB_SORT_THREADS * PB_SORT_ITEMS;

static inline uint32_t pbSortBlocks(uint32_t n) { return (n + PB_SORT_TILE - 1) / PB_SORT_TILE; }
// number of uint32 entries the histogram workspace needs for n pairs
// (the 256 x nblocks digit table, plus one sum per 2048-entry chunk of it for the scan)
static inline size_t pbSortHistEntries(uint32_t n) {
  const size_t table = (size_t)256 * (pbSortBlocks(n) ? pbSortBlocks(n) : 1);
  return table + (table + 2047) / 2048 + 1;
}

// Sorts n pairs by the low `bits` bits of the key; equal keys keep their input order (this is what
// thrust::sort_by_key gives the reference, particlebot_cuda.cu:377-382).  Buffers ping-pong; the
// return value is 0 when the result is in (keys, vals) and 1 when it is in (keys_tmp, vals_tmp).
// Returns -1 after recording a HIP error.
int pbRadixSortPairs(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                     uint32_t *hist, uint32_t n, int bits, hipStream_t stream, hipError_t *err);

static inline int pbKeyBits(uint32_t numKeys) {
  int b = 0;
  while (b < 32 && (numKeys > (1u << b))) b++;
  return b < 1 ? 1 : b;
}

// ---- XORWOW jump table (pb_xorwow.hpp) on the CURRENT device: built on the host once per process,
// uploaded once per device; returns nullptr and sets *err after a HIP error (pb_legacy.hip)
const uint32_t *pbXorwowDeviceTable(hipError_t *err);
