// Device-side utilities of the hot path: exclusive scan, reductions, mask -> index, NCHW -> NHWC.
// All stream-ordered, no host synchronisation.
#include "common.h"

extern "C" int cnrma_abi_version(void) { return CNRMA_ABI_VERSION; }

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

// phase B: one block scans the tile sums in place (exclusive), writes the grand total to tile_sum[n_tiles]
__global__ __launch_bounds__(1024) void scan_tile_offsets(int32_t* __restrict__ tile_sum, int64_t n_tiles) {
  __shared__ int smem[1024 / 64 + 1];
  int carry = 0;
  for (int64_t base = 0; base < n_tiles; base += 1024) {
    int64_t i = base + threadIdx.x;
    int v = i < n_tiles ? tile_sum[i] : 0;
    int total;
    int ex = block_excl_scan<1024>(v, smem, &total);
    if (i < n_tiles) tile_sum[i] = ex + carry;
    carry += total;
  }
  if (threadIdx.x == 0) tile_sum[n_tiles] = carry;
}

// phase C: rescan each tile with its offset.  MODE 0: out[i] = exclusive sum (and out[n] = total);
// MODE 1 (mask -> index): out[i] = in[i] ? rank : -1, n_sel[0] = total.
template <typename T, int MODE>
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
  int total;
  int run = block_excl_scan<SCAN_BLOCK>(s, smem, &total) + tile_off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = (MODE == 0) ? run : (v[j] ? run : -1);
    run += v[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (MODE == 0) out[n] = tile_off[n_tiles];
    if (total_out) total_out[0] = tile_off[n_tiles];
  }
}

template <typename T, int MODE>
int run_scan(const T* in, int32_t* out, int32_t* total_out, int64_t n, void* workspace, hipStream_t st) {
  if (n < 0) return CNRMA_EINVAL;
  int64_t n_tiles = ceil_div(n > 0 ? n : 1, SCAN_TILE);
  int32_t* tile = reinterpret_cast<int32_t*>(workspace);
  hipLaunchKernelGGL((scan_tile_sums<T>), dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, st, in, tile, n);
  hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(1024), 0, st, tile, n_tiles);
  hipLaunchKernelGGL((scan_apply<T, MODE>), dim3((unsigned)n_tiles), dim3(SCAN_BLOCK), 0, st, in, tile, out, total_out,
                     n, n_tiles);
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

}  // namespace

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
  dim3 grid((unsigned)ceil_div(HW, 64), (unsigned)ceil_div(C, 32), (unsigned)V);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, as_stream(stream), feat_nchw, feat_nhwc, C, HW);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
