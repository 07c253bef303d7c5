// Shared device/host helpers for libcnrma_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cnrma.h"

#define CNRMA_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return -(int)e__;      \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Byte fill as a plain kernel (replaces hipMemsetAsync: inside a captured launch sequence its memset nodes did not
// reliably re-execute on replay with this runtime -- stale hash tables on the second replay --, and a kernel node costs the
// same).  p 4-byte aligned, n a multiple of 4 bytes.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void cnrma_fill_kernel(uint32_t* __restrict__ p, uint32_t word, size_t n_words) {
  const size_t n4 = (((uintptr_t)p & 15) == 0) ? n_words / 4 : 0;
  const uint4 w4 = make_uint4(word, word, word, word);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<uint4*>(p)[i] = w4;
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
    p[i] = word;
}
static inline hipError_t cnrma_fill_bytes(void* p, int byte, size_t n_bytes, hipStream_t st) {
  if (n_bytes == 0) return hipSuccess;
  if ((n_bytes & 3) != 0 || ((uintptr_t)p & 3) != 0) return hipMemsetAsync(p, byte, n_bytes, st);
  const uint32_t b = (uint32_t)(byte & 0xFF), word = b | (b << 8) | (b << 16) | (b << 24);
  size_t blocks = (n_bytes / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(cnrma_fill_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), word,
                     n_bytes / 4);
  return hipGetLastError();
}

// Up to three byte fills in ONE launch (hash keys + values + a status word of a coordinate-set build: three launches of
// ~5 us each, 21 times per scene).  Regions 4-byte aligned, sizes multiples of 4 bytes; blocks are dealt to the regions in
// proportion to their sizes.
struct FillRegion { uint32_t* p; uint32_t word; size_t n_words; unsigned first_block, n_blocks; };
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void cnrma_fill3_kernel(FillRegion a, FillRegion b, FillRegion c) {
  const FillRegion& r = blockIdx.x >= c.first_block ? c : (blockIdx.x >= b.first_block ? b : a);
  if (r.n_blocks == 0) return;
  const unsigned lb = blockIdx.x - r.first_block;
  const size_t n4 = (((uintptr_t)r.p & 15) == 0) ? r.n_words / 4 : 0;
  const uint4 w4 = make_uint4(r.word, r.word, r.word, r.word);
  for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < n4; i += (size_t)r.n_blocks * blockDim.x)
    reinterpret_cast<uint4*>(r.p)[i] = w4;
  for (size_t i = 4 * n4 + (size_t)lb * blockDim.x + threadIdx.x; i < r.n_words; i += (size_t)r.n_blocks * blockDim.x)
    r.p[i] = r.word;
}
static inline hipError_t cnrma_fill_bytes3(void* p1, int b1, size_t n1, void* p2, int b2, size_t n2, void* p3, int b3, size_t n3,
                                           hipStream_t st) {
  void* ps[3] = {p1, p2, p3};
  const int bs[3] = {b1, b2, b3};
  const size_t ns[3] = {n1, n2, n3};
  FillRegion r[3];
  unsigned total = 0;
  for (int i = 0; i < 3; ++i) {
    if ((ns[i] & 3) != 0 || ((uintptr_t)ps[i] & 3) != 0) {          // odd shapes: the plain path, one launch each
      hipError_t e = cnrma_fill_bytes(p1, b1, n1, st);
      if (e == hipSuccess) e = cnrma_fill_bytes(p2, b2, n2, st);
      if (e == hipSuccess) e = cnrma_fill_bytes(p3, b3, n3, st);
      return e;
    }
    const uint32_t b = (uint32_t)(bs[i] & 0xFF);
    size_t blocks = ns[i] ? (ns[i] / 16 + 255) / 256 : 0;
    blocks = ns[i] && blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    r[i] = FillRegion{reinterpret_cast<uint32_t*>(ps[i]), b | (b << 8) | (b << 16) | (b << 24), ns[i] / 4, total, (unsigned)blocks};
    total += (unsigned)blocks;
  }
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(cnrma_fill3_kernel<0>, dim3(total), dim3(256), 0, st, r[0], r[1], r[2]);
  return hipGetLastError();
}

// live row count of a device-counted tensor: min(capacity, *n_dev) (n_dev may be NULL)
__device__ __forceinline__ int64_t live_rows(int64_t cap, const int32_t* n_dev) {
  if (n_dev == nullptr) return cap;
  int64_t n = (int64_t)__builtin_nontemporal_load(n_dev);
  return n < cap ? n : cap;
}

// ---- wave / block scans (wave64) --------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v += t;
  }
  return v;
}

// exclusive scan over a block of BLOCK threads (BLOCK multiple of 64, <= 1024); returns exclusive prefix and
// writes the block total to *total. smem: BLOCK/64 ints.
template <int BLOCK>
__device__ __forceinline__ int block_excl_scan(int v, int* smem, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  if (lane == 63) smem[wid] = incl;
  __syncthreads();
  if (wid == 0) {
    int w = lane < BLOCK / 64 ? smem[lane] : 0;
    int wi = wave_incl_scan(w);
    if (lane < BLOCK / 64) smem[lane] = wi - w;  // exclusive wave offsets
    if (lane == BLOCK / 64 - 1) smem[BLOCK / 64] = wi;
  }
  __syncthreads();
  int res = incl - v + smem[wid];
  *total = smem[BLOCK / 64];
  __syncthreads();
  return res;
}

// ---- 64-bit coordinate key: (b, x, y, z) with 16 / 16 / 16 / 16 bits, biased so that negatives order correctly
__device__ __forceinline__ uint64_t coord_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)(x + 32768) << 32) |
         ((uint64_t)(uint16_t)(y + 32768) << 16) | (uint64_t)(uint16_t)(z + 32768);
}
#define CNRMA_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint32_t hash_u64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

// find slot of key (or the empty slot where it would go). cap is a power of two.
__device__ __forceinline__ int64_t hash_find(const uint64_t* keys, int64_t cap, uint64_t key) {
  int64_t slot = hash_u64(key) & (cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    uint64_t k = keys[slot];
    if (k == key) return slot;
    if (k == CNRMA_EMPTY_KEY) return -1;
    slot = (slot + 1) & (cap - 1);
  }
  return -1;
}

// insert key; returns its slot (claims an empty slot with CAS when absent)
__device__ __forceinline__ int64_t hash_insert(uint64_t* keys, int64_t cap, uint64_t key) {
  int64_t slot = hash_u64(key) & (cap - 1);
  for (int64_t probe = 0; probe < cap; ++probe) {
    uint64_t k = keys[slot];
    if (k == key) return slot;
    if (k == CNRMA_EMPTY_KEY) {
      unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[slot]),
                                          (unsigned long long)CNRMA_EMPTY_KEY, (unsigned long long)key);
      if (prev == CNRMA_EMPTY_KEY || prev == key) return slot;
    }
    slot = (slot + 1) & (cap - 1);
  }
  return -1;
}
