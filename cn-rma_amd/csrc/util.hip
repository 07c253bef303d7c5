// Device-side utilities of the hot path: exclusive scan, reductions, mask -> index, NCHW -> NHWC.
// All stream-ordered, no host synchronisation.
#include "common.h"

extern "C" int cnrma_abi_version(void) { return CNRMA_ABI_VERSION; }

extern "C" int cnrma_fill_bytes_u8(void* dst, int byte, size_t n_bytes, void* stream) {
  if (dst == nullptr || (n_bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(dst) & 3) != 0) return CNRMA_EINVAL;
  if (n_bytes == 0) return 0;
  const hipError_t e = cnrma_fill_bytes(dst, byte, n_bytes, reinterpret_cast<hipStream_t>(stream));
  return e == hipSuccess ? 0 : -(int)e;
}

__global__ __launch_bounds__(256) void range_violations_kernel(const int32_t* __restrict__ v, const int32_t* __restrict__ lo,
                                                               const int32_t* __restrict__ hi, int n, int32_t* __restrict__ out) {
  __shared__ int sm[4];
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) c += (v[i] < lo[i] || v[i] > hi[i]) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sm[0] + sm[1] + sm[2] + sm[3];
}

extern "C" int cnrma_range_violations_i32(const int32_t* values, const int32_t* lo, const int32_t* hi, int n, int32_t* out,
                                          void* stream) {
  if (values == nullptr || lo == nullptr || hi == nullptr || out == nullptr || n <= 0 || n > 4096) return CNRMA_EINVAL;
  hipLaunchKernelGGL(range_violations_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), values, lo, hi, n, out);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

namespace {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                       // items per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;  // 2048 items per block

template <typename T>
__device__ __forceinline__ int scan_load(const T* in, int64_t i, int64_t n) {
  return i < n ? (int)in[i] : 0;
}

// phase A: per-tile sums
template <typename T>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_sums(const T* __restrict__ in, int32_t* __restrict__ tile_sum,
                                                             int64_t n) {
  __shared__ int smem[SCAN_BLOCK / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) s += scan_load(in, base + j, n);
  int total;
  block_excl_scan<SCAN_BLOCK>(s, smem, &total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

// phase B: one block scans the tile sums in place (exclusive), writes the grand total to tile_sum[n_tiles].
// Every thread owns a contiguous run of ceil(n_tiles / 1024) sums: ONE block-wide scan (three barriers) whatever the
// length -- the former loop of 1024-wide scans took 41 rounds of barriers at the north-star shape (87 M rows).
__global__ __launch_bounds__(1024) void scan_tile_offsets(int32_t* __restrict__ tile_sum, int64_t n_tiles) {
  __shared__ int smem[1024 / 64 + 1];
  const int64_t per = (n_tiles + 1023) / 1024;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < n_tiles ? lo + per : n_tiles;
  int s = 0;
  for (int64_t i = lo; i < hi; ++i) s += tile_sum[i];
  int total;
  int run = block_excl_scan<1024>(s, smem, &total);
  for (int64_t i = lo; i < hi; ++i) {
    const int v = tile_sum[i];
    tile_sum[i] = run;
    run += v;
  }
  if (threadIdx.x == 0) tile_sum[n_tiles] = total;
}

// phase C: rescan each tile with its offset.  MODE 0: out[i] = exclusive sum (and out[n] = total);
// MODE 1 (mask -> index): out[i] = in[i] ? rank : -1, n_sel[0] = total.
// OWN_OFFSET (n_tiles <= SCAN_TILE, i.e. up to 4 M items): there is no phase B -- every block adds up the sums of the tiles
// in front of it itself (<= 8 KB from L2), block 0 the grand total as well: two launches per scan instead of three.
template <typename T, int MODE, bool OWN_OFFSET>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply(const T* __restrict__ in, const int32_t* __restrict__ tile_off,
                                                         int32_t* __restrict__ out, int32_t* __restrict__ total_out,
                                                         int64_t n, int64_t n_tiles) {
  __shared__ int smem[SCAN_BLOCK / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = scan_load(in, base + j, n);
    s += v[j];
  }
  int offset, grand = 0;
  if constexpr (OWN_OFFSET) {
    const int upto = blockIdx.x == 0 ? (int)n_tiles : (int)blockIdx.x;   // block 0 needs the total (its own offset is 0)
    int part = 0;
    for (int i = threadIdx.x; i < upto; i += SCAN_BLOCK) part += tile_off[i];
    int sum;
    block_excl_scan<SCAN_BLOCK>(part, smem, &sum);
    __syncthreads();                                                  // smem is reused below
    offset = blockIdx.x == 0 ? 0 : sum;
    grand = sum;
  } else {
    offset = tile_off[blockIdx.x];
    if (blockIdx.x == 0) grand = tile_off[n_tiles];
  }
  int total;
  int run = block_excl_scan<SCAN_BLOCK>(s, smem, &total) + offset;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = (MODE == 0) ? run : (v[j] ? run : -1);
    run += v[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (MODE == 0) out[n] = grand;
    if (total_out) total_out[0] = grand;
  }
}

// short inputs (n <= SCAN_SINGLE_MAX: the coarse levels' strided sets, the neck's unions): ONE block, ONE launch -- every thread
// owns a contiguous run of ceil(n / 1024) items (round 6: a scene spends ~300 launches, two thirds of them this small)
constexpr int SCAN_SINGLE_MAX = 32768;
template <typename T, int MODE>
__global__ __launch_bounds__(1024) void scan_single_kernel(const T* __restrict__ in, int32_t* __restrict__ out,
                                                           int32_t* __restrict__ total_out, int n) {
  __shared__ int smem[1024 / 64 + 1];
  const int per = (n + 1023) / 1024;                       // <= 32
  const int lo = threadIdx.x * per, hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += (int)in[i];
  int total;
  int run = block_excl_scan<1024>(s, smem, &total);
  for (int i = lo; i < hi; ++i) {
    const int v = (int)in[i];
    out[i] = (MODE == 0) ? run : (v ? run : -1);
    run += v;
  }
  if (threadIdx.x == 0) {
    if (MODE == 0) out[n] = total;
    if (total_out) total_out[0] = total;
  }
}

template <typename T, int MODE>
int run_scan(const T* in, int32_t* out, int32_t* total_out, int64_t n, void* workspace, hipStream_t st) {
  if (n < 0) return CNRMA_EINVAL;
  if (n <= SCAN_SINGLE_MAX) {
    hipLaunchKernelGGL((scan_single_kernel<T, MODE>), dim3(1), dim3(1024), 0, st, in, out, total_out, (int)n);
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
  int64_t n_tiles = ceil_div(n > 0 ? n : 1, SCAN_TILE);
  int32_t* tile = reinterpret_cast<int32_t*>(workspace);
  hipLaunchKernelGGL((scan_tile_sums<T>), dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, st, in, tile, n);
  if (n_tiles <= SCAN_TILE) {
    hipLaunchKernelGGL((scan_apply<T, MODE, true>), dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, st, in, tile, out, total_out,
                       n, n_tiles);
  } else {
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(1024), 0, st, tile, n_tiles);
    hipLaunchKernelGGL((scan_apply<T, MODE, false>), dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, st, in, tile, out, total_out,
                       n, n_tiles);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// ---- fp64 sum ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_f64_partial(const double* __restrict__ in, double* __restrict__ part,
                                                       int64_t n) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += in[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void sum_f64_final(const double* __restrict__ part, int nparts,
                                                     double* __restrict__ out) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---- NCHW -> NHWC -------------------------------------------------------------------------------------------
// per view a [C][HW] -> [HW][C] transpose through a padded LDS tile (32 channels x 64 pixels)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int C, int64_t HW) {
  __shared__ float tile[32][65];
  const int64_t view = blockIdx.z;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const float* s = src + view * C * HW;
  float* d = dst + view * C * HW;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int c = c0 + ty + 4 * j;
    int64_t p = p0 + tx;
    tile[ty + 4 * j][tx] = (c < C && p < HW) ? s[(int64_t)c * HW + p] : 0.0f;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int c = c0 + cx;
    int64_t p = p0 + py + 8 * j;
    if (c < C && p < HW) d[p * C + c] = tile[cx][py + 8 * j];
  }
}

// 64 channels x 64 pixels per block with 16-byte accesses on both sides: a wave instruction reads four 256-byte runs of
// channel rows and writes four 256-byte runs of pixel rows (the dword version above moves a quarter of that per
// instruction).  Needs HW % 4 == 0 and C % 4 == 0.
__global__ __launch_bounds__(256) void nchw_to_nhwc_v4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                              int64_t HW) {
  __shared__ float tile[64][65];
  const int64_t view = blockIdx.z;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const float* s = src + view * C * HW;
  float* d = dst + view * C * HW;
  const int q = threadIdx.x & 15, r = threadIdx.x >> 4;     // 16 quads x 16 rows
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = r + 16 * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + c < C && p0 + 4 * q < HW) v = *reinterpret_cast<const float4*>(s + (int64_t)(c0 + c) * HW + p0 + 4 * q);
    tile[c][4 * q + 0] = v.x; tile[c][4 * q + 1] = v.y; tile[c][4 * q + 2] = v.z; tile[c][4 * q + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = r + 16 * j;
    if (c0 + 4 * q < C && p0 + p < HW) {
      const float4 o = make_float4(tile[4 * q + 0][p], tile[4 * q + 1][p], tile[4 * q + 2][p], tile[4 * q + 3][p]);
      *reinterpret_cast<float4*>(d + (p0 + p) * C + c0 + 4 * q) = o;
    }
  }
}

// ---- exact-count uniform random subset (device replacement of np.random.choice(M, n_keep, replace=False)) ----------
// every row gets a 32-bit hash key of (seed, index); the n_keep smallest keys are kept (radix select through two
// 16-bit histograms); ties on the threshold key are broken by index, so the result is deterministic for a seed.
__device__ __forceinline__ uint32_t row_hash(uint32_t seed, uint32_t i) {
  uint32_t x = i * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// selection key of row i: a hash (random subset) or the order-preserving image of a float score, inverted so that the
// LARGEST scores get the SMALLEST keys (top-k selection); ties are broken by index in both cases
__device__ __forceinline__ uint32_t select_key(const float* __restrict__ scores, uint32_t seed, uint32_t i) {
  if (scores == nullptr) return row_hash(seed, i);
  const uint32_t u = __float_as_uint(scores[i]);
  const uint32_t ordered = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending in the float order
  return ~ordered;
}

struct SampleWs {          // sampler workspace (int32 words)
  int32_t hist[3][2048];   // digit histograms: bits 31..21, 20..10, 9..0 of the key (within the selected prefix)
  int32_t prefix[3];       // selected digit per pass
  int32_t need[4];         // need[d] = rows still to take inside the prefix after pass d-1 (need[0] = n_keep)
  int32_t tie_count, pad;
  int32_t tie_idx[256];
  // more than 256 rows share the threshold key (e.g. thousands of exactly equal scores): a second select over the ROW
  // INDEX of the tie rows finds the need[3]-th smallest one exactly (sample_tie_select_kernel)
  int32_t prefix2[3];
  int32_t pad3;
  int32_t list_count, pad2;     // top-k indices: the survivors, appended in arrival order ...
  // ---- everything above is zeroed before a select; the list below is not (list_count bounds what is read)
  uint64_t list[1024];          // ... as (selection key << 32 | row): ascending = descending score, ties by row
};
constexpr size_t SAMPLE_WS_ZEROED = offsetof(SampleWs, list);

__device__ __forceinline__ int key_digit(uint32_t k, int pass) {
  return pass == 0 ? (int)(k >> 21) : (pass == 1 ? (int)((k >> 10) & 2047u) : (int)(k & 1023u));
}
__device__ __forceinline__ bool key_matches(uint32_t k, int pass, const int32_t* prefix) {
  if (pass >= 1 && (int)(k >> 21) != prefix[0]) return false;
  if (pass >= 2 && (int)((k >> 10) & 2047u) != prefix[1]) return false;
  return true;
}

// Every block resolves the digits of the finished passes itself from the complete global histograms (a 2048-bin scan by
// 256 threads) instead of a separate one-block "find" launch per pass: first bin whose cumulative count reaches `need`.
// All threads of the block call it; returns through *digit / *need_next (same values in every thread).
__device__ __forceinline__ void resolve_digit(const int32_t* __restrict__ hist, int need, int* digit, int* need_next,
                                              int* smem /* 256/64 + 1 + 2 ints */) {
  int bins[8], local = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { bins[j] = hist[8 * threadIdx.x + j]; local += bins[j]; }
  int* out = smem + 256 / 64 + 1;
  if (threadIdx.x == 0) { out[0] = 0; out[1] = 0; }       // need > total (n_keep >= M): the mask kernel keeps every row
  int total;
  int ex = block_excl_scan<256>(local, smem, &total);
  if (ex < need && ex + local >= need) {          // exactly one thread: the crossing lies in its 8 bins
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ex < need && ex + bins[j] >= need) { out[0] = 8 * threadIdx.x + j; out[1] = need - ex; }
      ex += bins[j];
    }
  }
  __syncthreads();
  *digit = out[0];
  *need_next = out[1];
  __syncthreads();
}

// block-private LDS histogram of one digit, flushed with (contiguous) global atomics
// live row count of a select: min(m_dev[0], capacity); the seed may carry a device-side increment (graph replays draw a
// fresh subset without a new kernel argument)
__device__ __forceinline__ int64_t select_rows(const int32_t* __restrict__ m_dev, int64_t m_cap) {
  const int64_t m = m_dev[0];
  return m < m_cap ? m : m_cap;
}
__device__ __forceinline__ uint32_t select_seed(uint32_t seed, const uint32_t* __restrict__ seed_dev) {
  return seed_dev ? seed + 0x9E3779B9u * seed_dev[0] : seed;
}

__global__ __launch_bounds__(256) void sample_hist_kernel(const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                          const float* __restrict__ scores, uint32_t seed,
                                                          const uint32_t* __restrict__ seed_dev, int pass, int n_keep,
                                                          SampleWs* __restrict__ ws) {
  seed = select_seed(seed, seed_dev);
  __shared__ int h[2048];
  __shared__ int rs[256 / 64 + 1 + 2];
  int32_t prefix[3] = {0, 0, 0};
  int need = n_keep;
  for (int q = 0; q < pass; ++q) {
    int nn;
    resolve_digit(ws->hist[q], need, &prefix[q], &nn, rs);
    need = nn;
  }
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  const int64_t M = select_rows(m_dev, m_cap);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = select_key(scores, seed, (uint32_t)i);
    if (key_matches(k, pass, prefix)) atomicAdd(&h[key_digit(k, pass)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (h[i]) atomicAdd(&ws->hist[pass][i], h[i]);
}

__global__ __launch_bounds__(256) void sample_ties_kernel(const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                          const float* __restrict__ scores, uint32_t seed,
                                                          const uint32_t* __restrict__ seed_dev, int n_keep,
                                                          SampleWs* __restrict__ ws) {
  seed = select_seed(seed, seed_dev);
  __shared__ int rs[256 / 64 + 1 + 2];
  int32_t prefix[3];
  int need = n_keep;
  for (int q = 0; q < 3; ++q) {
    int nn;
    resolve_digit(ws->hist[q], need, &prefix[q], &nn, rs);
    need = nn;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {      // the later kernels read the threshold key and the tie quota from here
    ws->prefix[0] = prefix[0]; ws->prefix[1] = prefix[1]; ws->prefix[2] = prefix[2];
    ws->need[0] = n_keep; ws->need[3] = need;
  }
  const int64_t M = select_rows(m_dev, m_cap);
  const uint32_t key = ((uint32_t)prefix[0] << 21) | ((uint32_t)prefix[1] << 10) | (uint32_t)prefix[2];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    if (select_key(scores, seed, (uint32_t)i) == key) {
      const int slot = atomicAdd(&ws->tie_count, 1);
      if (slot < 256) ws->tie_idx[slot] = (int32_t)i;
    }
  }
}

// second-level select, only active when the tie list overflowed (more than 256 rows carry the threshold key: thousands of
// exactly equal scores): the need[3]-th smallest ROW INDEX among the tie rows, by the same 3-digit radix select -- ONE
// workgroup runs the three passes itself (it used to be three grid-wide launches that were no-ops in all but that case;
// the case is cheap for score selects -- <= 0.5 M rows -- and does not occur for hashed keys)
__global__ __launch_bounds__(1024) void sample_tie_select_kernel(const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                                const float* __restrict__ scores, uint32_t seed,
                                                                const uint32_t* __restrict__ seed_dev,
                                                                SampleWs* __restrict__ ws) {
  if (ws->tie_count <= 256) return;
  seed = select_seed(seed, seed_dev);
  __shared__ int h[2048];
  __shared__ int sh_digit, sh_need;
  const int64_t M = select_rows(m_dev, m_cap);
  const uint32_t key = ((uint32_t)ws->prefix[0] << 21) | ((uint32_t)ws->prefix[1] << 10) | (uint32_t)ws->prefix[2];
  int32_t prefix2[3] = {0, 0, 0};
  int need = ws->need[3];
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = threadIdx.x; i < 2048; i += 1024) h[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < M; i += 1024) {
      if (select_key(scores, seed, (uint32_t)i) != key) continue;
      const uint32_t k2 = (uint32_t)i;                         // smaller index = kept first
      if (key_matches(k2, pass, prefix2)) atomicAdd(&h[key_digit(k2, pass)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                    // first bin whose cumulative count reaches `need`
      int cum = 0, d = 0, nn = 0;
      for (int bin = 0; bin < 2048; ++bin) {
        if (cum < need && cum + h[bin] >= need) { d = bin; nn = need - cum; break; }
        cum += h[bin];
      }
      sh_digit = d; sh_need = nn;
    }
    __syncthreads();
    prefix2[pass] = sh_digit;
    need = sh_need;
    __syncthreads();
  }
  if (threadIdx.x == 0) { ws->prefix2[0] = prefix2[0]; ws->prefix2[1] = prefix2[1]; ws->prefix2[2] = prefix2[2]; }
}

// index bound of the rows that carry the threshold key: kept when their index <= bound (block-uniform; all threads call)
__device__ __forceinline__ int32_t select_tie_bound(const SampleWs* __restrict__ ws, int32_t* sh_bound, int* rs) {
  (void)rs;
  if (threadIdx.x == 0) {
    // the need smallest indices among the (normally 1, at most 256) rows that carry the threshold key
    const int cnt = min(ws->tie_count, 256), need = ws->need[3];
    int32_t bound = -1;
    if (ws->tie_count > 256) {         // exact: the need-th smallest tie index (sample_tie_select_kernel)
      bound = (int32_t)(((uint32_t)ws->prefix2[0] << 21) | ((uint32_t)ws->prefix2[1] << 10) | (uint32_t)ws->prefix2[2]);
    } else {
      for (int r = 0; r < need; ++r) {
        int32_t best = 0x7FFFFFFF;
        for (int q = 0; q < cnt; ++q) { const int32_t v = ws->tie_idx[q]; if (v > bound && v < best) best = v; }
        bound = best;
      }
    }
    *sh_bound = bound;
  }
  __syncthreads();
  return *sh_bound;
}

__global__ __launch_bounds__(256) void sample_mask_kernel(const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                          const float* __restrict__ scores, uint32_t seed,
                                                          const uint32_t* __restrict__ seed_dev,
                                                          const SampleWs* __restrict__ ws, int n_keep,
                                                          uint8_t* __restrict__ mask) {
  seed = select_seed(seed, seed_dev);
  const int64_t M = select_rows(m_dev, m_cap);
  __shared__ int32_t sh_bound;      // rows with the threshold key are kept when their index <= tie_bound
  __shared__ int rs[256 / 64 + 1 + 2];
  const int32_t tie_bound = select_tie_bound(ws, &sh_bound, rs);
  const bool all = M <= n_keep;
  const uint32_t key = ((uint32_t)ws->prefix[0] << 21) | ((uint32_t)ws->prefix[1] << 10) | (uint32_t)ws->prefix[2];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = select_key(scores, seed, (uint32_t)i);
    mask[i] = (all || k < key || (k == key && (int32_t)i <= tie_bound)) ? 1 : 0;
  }
  // rows between the live count and the capacity are not part of the tensor: never kept
  for (int64_t i = M + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m_cap; i += (int64_t)gridDim.x * blockDim.x)
    mask[i] = 0;
}

// top-k INDICES: the kept rows go to a list instead of a mask ...
__global__ __launch_bounds__(256) void sample_collect_kernel(const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                             const float* __restrict__ scores, SampleWs* __restrict__ ws,
                                                             int n_keep) {
  const int64_t M = select_rows(m_dev, m_cap);
  __shared__ int32_t sh_bound;
  __shared__ int rs[256 / 64 + 1 + 2];
  const int32_t tie_bound = select_tie_bound(ws, &sh_bound, rs);
  const bool all = M <= n_keep;
  const uint32_t key = ((uint32_t)ws->prefix[0] << 21) | ((uint32_t)ws->prefix[1] << 10) | (uint32_t)ws->prefix[2];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t k = select_key(scores, 0u, (uint32_t)i);
    if (all || k < key || (k == key && (int32_t)i <= tie_bound)) {
      const int slot = atomicAdd(&ws->list_count, 1);
      if (slot < 1024) ws->list[slot] = ((uint64_t)k << 32) | (uint32_t)i;
    }
  }
}

// ... which one workgroup sorts (bitonic, 1024 packed keys in LDS): out[0..min(M, k)) = the rows in descending score order,
// ties by smaller row -- torch.topk(scores, k)[1]; the slots behind them repeat row 0
__global__ __launch_bounds__(1024) void topk_sort_kernel(const SampleWs* __restrict__ ws, int k, int64_t* __restrict__ out) {
  __shared__ uint64_t a[1024];
  const int t = threadIdx.x;
  const int n = min(min(ws->list_count, k), 1024);
  a[t] = t < n ? ws->list[t] : ~0ull;
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = t ^ stride;
      if (partner > t) {
        const bool up = (t & size) == 0;
        const uint64_t x = a[t], y = a[partner];
        if ((x > y) == up) { a[t] = y; a[partner] = x; }
      }
      __syncthreads();
    }
  }
  if (t < k) out[t] = t < n ? (int64_t)(uint32_t)a[t] : 0;
}

// ---- the random subset straight from the per-ray sample records (no 72-M-row mask, no 72-M-row index) ----------------
// Rows are the kept samples of the rays in ray order: ray r owns rows [off[r], off[r+1]).  The keep predicate of the select is
// a pure function of the row number, so a ray counts its kept rows itself, a scan over the RAYS (12 M, not 72 M rows) places
// them, and the records are written in the same order the mask + index pair produced.
__global__ __launch_bounds__(256) void select_count_segments_kernel(int64_t R, const int32_t* __restrict__ off,
                                                                    const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                                    uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                                    const SampleWs* __restrict__ ws, int n_keep,
                                                                    int32_t* __restrict__ counts) {
  seed = select_seed(seed, seed_dev);
  const int64_t M = select_rows(m_dev, m_cap);
  __shared__ int32_t sh_bound;
  __shared__ int rs[256 / 64 + 1 + 2];
  const int32_t tie_bound = select_tie_bound(ws, &sh_bound, rs);
  const bool all = M <= n_keep;
  const uint32_t key = ((uint32_t)ws->prefix[0] << 21) | ((uint32_t)ws->prefix[1] << 10) | (uint32_t)ws->prefix[2];
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  int64_t m0 = off[r], m1 = off[r + 1];
  if (m1 > M) m1 = M;
  int c = 0;
  for (int64_t i = m0; i < m1; ++i) {
    const uint32_t k = row_hash(seed, (uint32_t)i);
    c += (all || k < key || (k == key && (int32_t)i <= tie_bound)) ? 1 : 0;
  }
  counts[r] = c;
}

__global__ __launch_bounds__(256) void select_scatter_records_kernel(int64_t R, const int32_t* __restrict__ off,
                                                                     const int32_t* __restrict__ sel_off,
                                                                     const int2* __restrict__ kept, int cap,
                                                                     const int32_t* __restrict__ m_dev, int64_t m_cap,
                                                                     uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                                     const SampleWs* __restrict__ ws, int n_keep,
                                                                     int64_t rec_cap, int4* __restrict__ rec) {
  seed = select_seed(seed, seed_dev);
  const int64_t M = select_rows(m_dev, m_cap);
  __shared__ int32_t sh_bound;
  __shared__ int rs[256 / 64 + 1 + 2];
  const int32_t tie_bound = select_tie_bound(ws, &sh_bound, rs);
  const bool all = M <= n_keep;
  const uint32_t key = ((uint32_t)ws->prefix[0] << 21) | ((uint32_t)ws->prefix[1] << 10) | (uint32_t)ws->prefix[2];
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int64_t m0 = off[r];
  int64_t m1 = off[r + 1];
  if (m1 > M) m1 = M;
  int64_t j = sel_off[r];
  for (int64_t i = m0; i < m1; ++i) {
    const uint32_t k = row_hash(seed, (uint32_t)i);
    if (all || k < key || (k == key && (int32_t)i <= tie_bound)) {
      if (j < rec_cap && i - m0 < cap) {
        const int2 kv = kept[r * cap + (i - m0)];
        rec[j] = make_int4((int)r, kv.y, kv.x, 0);
      }
      ++j;
    }
  }
}

}  // namespace

extern "C" size_t cnrma_sample_workspace_bytes(void) { return sizeof(SampleWs); }

// mask != nullptr: keep-mask; else out_idx: the kept rows in selection order (n_keep <= 1024)
static int run_select(const int32_t* m_dev, const float* scores, int64_t m_cap, int n_keep, uint32_t seed,
                      const uint32_t* seed_dev, uint8_t* mask, void* workspace, hipStream_t st, int64_t* out_idx = nullptr) {
  if (m_cap <= 0 || n_keep <= 0 || m_cap >= ((int64_t)1 << 31)) return CNRMA_EINVAL;
  SampleWs* ws = reinterpret_cast<SampleWs*>(workspace);
  hipError_t e = cnrma_fill_bytes(ws, 0, SAMPLE_WS_ZEROED, st);
  if (e != hipSuccess) return -(int)e;
  int blocks = (int)(m_cap / 2048 + 1);
  if (blocks > 1024) blocks = 1024;
  for (int pass = 0; pass < 3; ++pass)
    hipLaunchKernelGGL(sample_hist_kernel, dim3(blocks), dim3(256), 0, st, m_dev, m_cap, scores, seed, seed_dev, pass,
                       n_keep, ws);
  hipLaunchKernelGGL(sample_ties_kernel, dim3(blocks), dim3(256), 0, st, m_dev, m_cap, scores, seed, seed_dev, n_keep, ws);
  hipLaunchKernelGGL(sample_tie_select_kernel, dim3(1), dim3(1024), 0, st, m_dev, m_cap, scores, seed, seed_dev, ws);   // a no-op
                                                                     // unless more than 256 rows carry the threshold key
  if (mask != nullptr) {
    hipLaunchKernelGGL(sample_mask_kernel, dim3(blocks), dim3(256), 0, st, m_dev, m_cap, scores, seed, seed_dev, ws, n_keep,
                       mask);
  } else if (out_idx == nullptr) {
    // threshold only: the caller evaluates the keep predicate itself (cnrma_rma_select_records)
  } else {
    hipLaunchKernelGGL(sample_collect_kernel, dim3(blocks), dim3(256), 0, st, m_dev, m_cap, scores, ws, n_keep);
    hipLaunchKernelGGL(topk_sort_kernel, dim3(1), dim3(1024), 0, st, ws, n_keep, out_idx);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// mask[0..m_cap): exactly min(M, n_keep) ones among the first M = min(m_dev[0], m_cap) rows (uniformly random subset),
// zeros behind them
extern "C" int cnrma_sample_mask(const int32_t* m_dev, int64_t m_cap, int n_keep, uint32_t seed, const uint32_t* seed_dev,
                                 uint8_t* mask, void* workspace, void* stream) {
  if (m_dev == nullptr) return CNRMA_EINVAL;
  return run_select(m_dev, nullptr, m_cap, n_keep, seed, seed_dev, mask, workspace, as_stream(stream));
}

// mask[0..n): ones at the min(n, k) rows with the LARGEST scores (ties by smaller index) -- the keep-set of
// torch.topk(scores, k) without the sort; n = n_dev[0] read on the device
extern "C" int cnrma_topk_mask_f32(const float* scores, const int32_t* n_dev, int64_t n_cap, int k, uint8_t* mask,
                                   void* workspace, void* stream) {
  if (scores == nullptr || n_dev == nullptr) return CNRMA_EINVAL;
  return run_select(n_dev, scores, n_cap, k, 0u, nullptr, mask, workspace, as_stream(stream));
}

// The random subset of cnrma_sample_mask applied to the march's per-ray sample records: records[j] = {ray, step, weight bits,
// 0} of the j-th kept row in row order, n_sel[0] = their number (<= n_keep) -- what cnrma_sample_mask + cnrma_mask_to_index
// + the record scatter of cnrma_rma_neus_emit_rows_f32 produce, without the m_cap-sized mask and index arrays.
// ray_counts [R], ray_offsets [R + 1]: scratch; scan_ws: cnrma_scan_workspace_bytes(R); sample_ws: cnrma_sample_workspace_bytes().
extern "C" int cnrma_rma_select_records(const int32_t* row_offset, int64_t R, const void* kept, int cap, const int32_t* m_dev,
                                        int64_t m_cap, int n_keep, uint32_t seed, const uint32_t* seed_dev, void* sample_ws,
                                        int32_t* ray_counts, int32_t* ray_offsets, void* scan_ws, int64_t rec_cap,
                                        void* records, int32_t* n_sel, void* stream) {
  if (row_offset == nullptr || kept == nullptr || m_dev == nullptr || R <= 0 || cap <= 0 || records == nullptr ||
      ray_counts == nullptr || ray_offsets == nullptr || n_sel == nullptr)
    return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  int rc = run_select(m_dev, nullptr, m_cap, n_keep, seed, seed_dev, nullptr, sample_ws, st, nullptr);
  if (rc != 0) return rc;
  const SampleWs* ws = reinterpret_cast<const SampleWs*>(sample_ws);
  const unsigned blocks = (unsigned)ceil_div(R, 256);
  hipLaunchKernelGGL(select_count_segments_kernel, dim3(blocks), dim3(256), 0, st, R, row_offset, m_dev, m_cap, seed, seed_dev,
                     ws, n_keep, ray_counts);
  rc = run_scan<int32_t, 0>(ray_counts, ray_offsets, n_sel, R, scan_ws, st);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(select_scatter_records_kernel, dim3(blocks), dim3(256), 0, st, R, row_offset, ray_offsets,
                     reinterpret_cast<const int2*>(kept), cap, m_dev, m_cap, seed, seed_dev, ws, n_keep, rec_cap,
                     reinterpret_cast<int4*>(records));
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// out_idx[0..k): the rows of the k largest scores in descending score order (ties by smaller row) -- torch.topk(scores,
// k)[1] -- for k <= 1024; with fewer than k live rows the live rows come first and the remaining slots hold row 0
extern "C" int cnrma_topk_indices_f32(const float* scores, const int32_t* n_dev, int64_t n_cap, int k, int64_t* out_idx,
                                      void* workspace, void* stream) {
  if (scores == nullptr || n_dev == nullptr || out_idx == nullptr || k > 1024) return CNRMA_EINVAL;
  return run_select(n_dev, scores, n_cap, k, 0u, nullptr, nullptr, workspace, as_stream(stream), out_idx);
}

extern "C" size_t cnrma_scan_workspace_bytes(int64_t n) {
  int64_t n_tiles = ceil_div(n > 0 ? n : 1, SCAN_TILE);
  size_t scan = (size_t)(n_tiles + 2) * sizeof(int32_t);
  size_t sum = 1024 * sizeof(double);
  return (scan > sum ? scan : sum) + 64;
}

extern "C" int cnrma_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* workspace, void* stream) {
  return run_scan<int32_t, 0>(in, out, nullptr, n, workspace, as_stream(stream));
}

extern "C" int cnrma_mask_to_index(const uint8_t* mask, int32_t* sel_index, int32_t* n_sel, int64_t n,
                                   void* workspace, void* stream) {
  return run_scan<uint8_t, 1>(mask, sel_index, n_sel, n, workspace, as_stream(stream));
}

extern "C" int cnrma_sum_f64(const double* in, double* out, int64_t n, void* workspace, void* stream) {
  if (n < 0) return CNRMA_EINVAL;
  int nblk = (int)(n / 2048 + 1);
  if (nblk > 1024) nblk = 1024;
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sum_f64_partial, dim3(nblk), dim3(256), 0, as_stream(stream), in, part, n);
  hipLaunchKernelGGL(sum_f64_final, dim3(1), dim3(256), 0, as_stream(stream), part, nblk, out);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_nchw_to_nhwc_f32(const float* feat_nchw, float* feat_nhwc, int V, int C, int H, int W,
                                      void* stream) {
  if (V <= 0 || C <= 0 || H <= 0 || W <= 0) return CNRMA_EINVAL;
  int64_t HW = (int64_t)H * W;
  if (HW % 4 == 0 && C % 4 == 0 && ((((uintptr_t)feat_nchw) | ((uintptr_t)feat_nhwc)) & 15) == 0) {
    dim3 grid4((unsigned)ceil_div(HW, 64), (unsigned)ceil_div(C, 64), (unsigned)V);
    hipLaunchKernelGGL(nchw_to_nhwc_v4_kernel, grid4, dim3(256), 0, as_stream(stream), feat_nchw, feat_nhwc, C, HW);
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((unsigned)ceil_div(HW, 64), (unsigned)ceil_div(C, 32), (unsigned)V);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, as_stream(stream), feat_nchw, feat_nhwc, C, HW);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
