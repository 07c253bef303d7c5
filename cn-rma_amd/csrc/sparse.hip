// Sparse-voxel engine: voxelisation, coordinate maps, kernel maps, fused sparse convolution on fp32 MFMA,
// generative transposed convolution, pooling, instance norm, union-add, interpolation, pruning.
// Replaces the MinkowskiEngine v0.5.4 surface used by the reference.
//
// Data model: coordinates int32 [N][4] = (batch, x, y, z); features fp32 [N][C] row-major (one or more full
// 128-B lines per row for C >= 32); a coordinate map is an open-addressing hash table (uint64 keys, int32
// row values, power-of-two capacity >= 2N) that lives next to the tensor.  Convolutions are OUTPUT-STATIONARY:
// a neighbour table nbr[No][K] (input row or -1) is built once per (coordinate set, kernel) pair and the
// convolution is a gather-GEMM with no atomics -- deterministic, and the BatchNorm / bias / residual /
// activation epilogue is fused into the store.
#include <cstdio>
#include <type_traits>
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

// ================================================================================================================
// unique (first occurrence wins, output in first-occurrence order) -- shared by voxelise and strided coords
// ================================================================================================================
struct UniqueWs {
  int32_t* slot;    // [n] hash slot of row i
  int32_t* idx;     // [n] output row of i or -1
  uint8_t* flag;    // [n] 1 when row i is the representative of its voxel
  void* scan;       // scan workspace
  // Morton ordering (voxelise): radix-sort double buffers + hipcub temporary storage
  uint64_t* key_a; uint64_t* key_b;
  int32_t* val_a; int32_t* val_b;
  void* sort_tmp; size_t sort_tmp_bytes;
  int32_t* range_err;   // set when a quantised coordinate does not fit the 16-bit fields of coord_key()
};

size_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                                    (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, 0, 64, nullptr);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)n * 16 + (1u << 20);   // no device (CPU-only container): estimate
  (void)hipGetLastError();
  return (bytes + 255) / 256 * 256;
}

__host__ UniqueWs carve_unique_ws(void* workspace, int64_t n) {
  UniqueWs w;
  char* p = reinterpret_cast<char*>(workspace);
  int64_t n4 = (n + 3) / 4 * 4;
  w.key_a = reinterpret_cast<uint64_t*>(p); p += n4 * 8;
  w.key_b = reinterpret_cast<uint64_t*>(p); p += n4 * 8;
  w.val_a = reinterpret_cast<int32_t*>(p); p += n4 * 4;
  w.val_b = reinterpret_cast<int32_t*>(p); p += n4 * 4;
  w.slot = reinterpret_cast<int32_t*>(p); p += n4 * 4;
  w.idx = reinterpret_cast<int32_t*>(p); p += n4 * 4;
  w.flag = reinterpret_cast<uint8_t*>(p); p += n4;
  p = reinterpret_cast<char*>(((uintptr_t)p + 255) & ~(uintptr_t)255);
  w.sort_tmp_bytes = sort_temp_bytes(n);
  w.sort_tmp = p; p += w.sort_tmp_bytes;
  w.range_err = reinterpret_cast<int32_t*>(p); p += 256;
  w.scan = p;
  return w;
}

// 48-bit Morton code of the biased coordinates (bias 32768 is a multiple of every tensor stride, so the children of
// one strided parent are contiguous in this order and every coarser level inherits the ordering), batch on top
__device__ __forceinline__ uint64_t spread3(uint32_t v) {
  uint64_t x = v & 0xFFFFu;
  x = (x | (x << 32)) & 0x00FF00000000FFFFull;   // not needed for 16 bits but keeps the classic ladder readable
  x = (x | (x << 16)) & 0x00FF0000FF0000FFull;
  x = (x | (x << 8)) & 0xF00F00F00F00F00Full;
  x = (x | (x << 4)) & 0x30C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x9249249249249249ull;
  return x;
}
__device__ __forceinline__ uint64_t morton_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint16_t)b << 48) | (spread3((uint32_t)(x + 32768)) << 2) | (spread3((uint32_t)(y + 32768)) << 1) |
         spread3((uint32_t)(z + 32768));
}

// MODE 0: float coords / voxel_size -> floor (voxelise); MODE 1: int coords -> floor(p / s) * s (stride)
template <int MODE>
__device__ __forceinline__ void quantise(const void* src, int64_t i, float vs, int new_stride, int batch_id, int* b,
                                         int* x, int* y, int* z) {
  if (MODE == 0) {
    const float* c = reinterpret_cast<const float*>(src) + i * 3;
    *b = batch_id;
    *x = (int)floorf(c[0] / vs);   // ME batch_sparse_collate: floor then int32 (true fp32 division, :329)
    *y = (int)floorf(c[1] / vs);
    *z = (int)floorf(c[2] / vs);
  } else {
    const int32_t* c = reinterpret_cast<const int32_t*>(src) + i * 4;
    auto fl = [new_stride](int p) {
      int q = p / new_stride;
      if ((p % new_stride != 0) && (p < 0)) --q;  // floor toward -inf
      return q * new_stride;
    };
    *b = c[0]; *x = fl(c[1]); *y = fl(c[2]); *z = fl(c[3]);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void uniq_insert_kernel(const void* __restrict__ src, int64_t n_cap,
                                                          const int32_t* __restrict__ n_dev, float vs, int new_stride,
                                                          int batch_id, uint64_t* __restrict__ keys,
                                                          int32_t* __restrict__ vals, int64_t cap,
                                                          int32_t* __restrict__ slot_out, int32_t* __restrict__ range_err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= live_rows(n_cap, n_dev)) return;
  int b, x, y, z;
  quantise<MODE>(src, i, vs, new_stride, batch_id, &b, &x, &y, &z);
  if (MODE == 0) {
    // coord_key() packs batch / x / y / z into 16 bits each: a coordinate beyond +-32767 voxels (327 m at 1 cm) or a NaN
    // would alias another voxel silently -- reported through n_out = -1 instead
    const float* c = reinterpret_cast<const float*>(src) + i * 3;
    const float lim = 32767.0f * vs;
    const bool ok = fabsf(c[0]) < lim && fabsf(c[1]) < lim && fabsf(c[2]) < lim && b >= 0 && b < 65536;   // false for NaN
    if (!ok) { *range_err = 1; slot_out[i] = -1; return; }
  }
  int64_t s = hash_insert(keys, cap, coord_key(b, x, y, z));
  slot_out[i] = (int32_t)s;
  if (s >= 0) atomicMin(&vals[s], (int32_t)i);
}

__global__ __launch_bounds__(256) void uniq_flag_kernel(int64_t n_cap, const int32_t* __restrict__ n_dev,
                                                        const int32_t* __restrict__ vals,
                                                        const int32_t* __restrict__ slot, uint8_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  uint8_t f = 0;
  if (i < live_rows(n_cap, n_dev)) {
    int32_t s = slot[i];
    f = (s >= 0 && vals[s] == (int32_t)i) ? 1 : 0;
  }
  flag[i] = f;
}

template <int MODE>
__global__ __launch_bounds__(256) void uniq_write_kernel(const void* __restrict__ src, int64_t n_cap,
                                                         const int32_t* __restrict__ n_dev, float vs, int new_stride,
                                                         int batch_id, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ slot, int32_t* __restrict__ vals,
                                                         int32_t* __restrict__ out_coords,
                                                         int32_t* __restrict__ out_src, int64_t out_cap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= live_rows(n_cap, n_dev)) return;
  const int32_t j = idx[i];
  if (j < 0) return;
  if (j >= out_cap) { vals[slot[i]] = -1; return; }   // beyond the planned capacity: dropped, and unmapped (see n_out)
  int b, x, y, z;
  quantise<MODE>(src, i, vs, new_stride, batch_id, &b, &x, &y, &z);
  reinterpret_cast<int4*>(out_coords)[j] = make_int4(b, x, y, z);
  if (out_src) out_src[j] = (int32_t)i;
  vals[slot[i]] = j;  // the table now maps voxel key -> output row
}

// sort key of row i: Morton code for representatives, all-ones (sorted last) for duplicates / dead rows
template <int MODE>
__global__ __launch_bounds__(256) void uniq_sortkey_kernel(const void* __restrict__ src, int64_t n_cap,
                                                           const int32_t* __restrict__ n_dev, float vs, int new_stride,
                                                           int batch_id, const uint8_t* __restrict__ flag,
                                                           uint64_t* __restrict__ key, int32_t* __restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  uint64_t k = ~0ull;
  if (i < live_rows(n_cap, n_dev) && flag[i]) {
    int b, x, y, z;
    quantise<MODE>(src, i, vs, new_stride, batch_id, &b, &x, &y, &z);
    k = morton_key(b, x, y, z);
  }
  key[i] = k;
  val[i] = (int32_t)i;
}

template <int MODE>
__global__ __launch_bounds__(256) void uniq_write_sorted_kernel(const void* __restrict__ src, int64_t n_cap, float vs,
                                                                int new_stride, int batch_id,
                                                                const int32_t* __restrict__ order,
                                                                const int32_t* __restrict__ n_out,
                                                                const int32_t* __restrict__ slot,
                                                                int32_t* __restrict__ vals,
                                                                int32_t* __restrict__ out_coords,
                                                                int32_t* __restrict__ out_src, int64_t out_cap) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= live_rows(n_cap, n_out)) return;
  const int64_t i = order[j];
  if (j >= out_cap) { vals[slot[i]] = -1; return; }
  int b, x, y, z;
  quantise<MODE>(src, i, vs, new_stride, batch_id, &b, &x, &y, &z);
  reinterpret_cast<int4*>(out_coords)[j] = make_int4(b, x, y, z);
  if (out_src) out_src[j] = (int32_t)i;
  vals[slot[i]] = (int32_t)j;
}

// out[j][:] = in[src[j]][:]   (lanes across channels: coalesced row copies)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ src,
                                                          int64_t n_cap, const int32_t* __restrict__ n_dev, int C,
                                                          float* __restrict__ out) {
  const int64_t n = live_rows(n_cap, n_dev);
  if ((C & 3) == 0) {
    const int c4 = C >> 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * c4; t += (int64_t)gridDim.x * blockDim.x) {
      int64_t j = t / c4;
      int c = (int)(t - j * c4);
      reinterpret_cast<float4*>(out)[j * c4 + c] = reinterpret_cast<const float4*>(in)[(int64_t)src[j] * c4 + c];
    }
  } else {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
      int64_t j = t / C;
      int c = (int)(t - j * C);
      out[j * C + c] = in[(int64_t)src[j] * C + c];
    }
  }
}

__global__ void uniq_range_report_kernel(const int32_t* __restrict__ range_err, int32_t* __restrict__ n_out) {
  if (*range_err) n_out[0] = -1;
}

template <int MODE>
int run_unique(const void* src, int64_t n_cap, const int32_t* n_dev, float vs, int new_stride, int batch_id,
               uint64_t* keys, int32_t* vals, int64_t cap, int32_t* out_coords, int32_t* out_src, int32_t* n_out,
               void* workspace, hipStream_t st, bool morton = false, int64_t out_cap = 0) {
  if (out_cap <= 0 || out_cap > n_cap) out_cap = n_cap;
  if (n_cap <= 0 || cap < 2 || (cap & (cap - 1)) != 0 || cap >= ((int64_t)1 << 31)) return CNRMA_EINVAL;
  UniqueWs w = carve_unique_ws(workspace, n_cap);
  hipError_t e = cnrma_fill_bytes3(keys, 0xFF, (size_t)cap * sizeof(uint64_t), vals, 0x7F, (size_t)cap * sizeof(int32_t),
                                   w.range_err, 0, 4, st);                      // one launch instead of three
  if (e != hipSuccess) return -(int)e;
  const unsigned nb = (unsigned)ceil_div(n_cap, 256);
  hipLaunchKernelGGL((uniq_insert_kernel<MODE>), dim3(nb), dim3(256), 0, st, src, n_cap, n_dev, vs, new_stride,
                     batch_id, keys, vals, cap, w.slot, w.range_err);
  hipLaunchKernelGGL(uniq_flag_kernel, dim3(nb), dim3(256), 0, st, n_cap, n_dev, vals, w.slot, w.flag);
  int rc = cnrma_mask_to_index(w.flag, w.idx, n_out, n_cap, w.scan, st);
  if (rc) return rc;
  if (MODE == 0) hipLaunchKernelGGL(uniq_range_report_kernel, dim3(1), dim3(1), 0, st, w.range_err, n_out);
  if (!morton) {
    hipLaunchKernelGGL((uniq_write_kernel<MODE>), dim3(nb), dim3(256), 0, st, src, n_cap, n_dev, vs, new_stride,
                       batch_id, w.idx, w.slot, vals, out_coords, out_src, out_cap);
  } else {
    // spatial (Morton) row order: a tile of consecutive rows is a compact block of voxels, so the gathers of the
    // convolutions hit L2 and whole kernel offsets can be skipped per tile on thin surfaces
    hipLaunchKernelGGL((uniq_sortkey_kernel<MODE>), dim3(nb), dim3(256), 0, st, src, n_cap, n_dev, vs, new_stride,
                       batch_id, w.flag, w.key_a, w.val_a);
    size_t tmp = w.sort_tmp_bytes;
    // batch id sits in bits 48..63 and dead rows carry the all-ones key: for batch 0 bits [0, 49) order everything
    const int end_bit = batch_id == 0 ? 49 : 64;
    hipError_t e2 = hipcub::DeviceRadixSort::SortPairs(w.sort_tmp, tmp, w.key_a, w.key_b, w.val_a, w.val_b, (int)n_cap,
                                                       0, end_bit, st);
    if (e2 != hipSuccess) return -(int)e2;
    hipLaunchKernelGGL((uniq_write_sorted_kernel<MODE>), dim3(nb), dim3(256), 0, st, src, n_cap, vs, new_stride,
                       batch_id, w.val_b, n_out, w.slot, vals, out_coords, out_src, out_cap);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================================
// coordinate map / kernel map
// ================================================================================================================
__global__ __launch_bounds__(256) void build_map_kernel(const int32_t* __restrict__ coords, int64_t n_cap,
                                                        const int32_t* __restrict__ n_dev, uint64_t* __restrict__ keys,
                                                        int32_t* __restrict__ vals, int64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= live_rows(n_cap, n_dev)) return;
  int4 c = reinterpret_cast<const int4*>(coords)[i];
  int64_t s = hash_insert(keys, cap, coord_key(c.x, c.y, c.z, c.w));
  if (s >= 0) vals[s] = (int32_t)i;
}

__global__ __launch_bounds__(256) void kernel_map_kernel(const int32_t* __restrict__ out_coords, int64_t no_cap,
                                                         const int32_t* __restrict__ no_dev,
                                                         const uint64_t* __restrict__ keys,
                                                         const int32_t* __restrict__ vals, int64_t cap,
                                                         const int32_t* __restrict__ offsets, int K,
                                                         int32_t* __restrict__ nbr) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = live_rows(no_cap, no_dev);
  if (t >= n * K) return;
  const int64_t o = t / K;
  const int k = (int)(t - o * K);
  int4 c = reinterpret_cast<const int4*>(out_coords)[o];
  int64_t s = hash_find(keys, cap, coord_key(c.x, c.y + offsets[k * 3], c.z + offsets[k * 3 + 1], c.w + offsets[k * 3 + 2]));
  nbr[t] = s >= 0 ? vals[s] : -1;
}

// stride-1, odd kernel: the neighbour relation is symmetric (nbr[o][k] = i  <=>  nbr[i][K-1-k] = o), so only the
// first K/2 offsets are looked up and both entries are written; the centre is the row itself.  nbr must be
// pre-filled with -1.
__global__ __launch_bounds__(256) void kernel_map_symmetric_kernel(const int32_t* __restrict__ coords, int64_t n_cap,
                                                                   const int32_t* __restrict__ n_dev,
                                                                   const uint64_t* __restrict__ keys,
                                                                   const int32_t* __restrict__ vals, int64_t cap,
                                                                   const int32_t* __restrict__ offsets, int K,
                                                                   int32_t* __restrict__ nbr) {
  const int half = K / 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = live_rows(n_cap, n_dev);
  if (t >= n * (half + 1)) return;
  const int64_t o = t / (half + 1);
  const int k = (int)(t - o * (half + 1));
  if (k == half) { nbr[o * K + half] = (int32_t)o; return; }
  int4 c = reinterpret_cast<const int4*>(coords)[o];
  int64_t s = hash_find(keys, cap, coord_key(c.x, c.y + offsets[k * 3], c.z + offsets[k * 3 + 1], c.w + offsets[k * 3 + 2]));
  if (s >= 0) {
    const int32_t i = vals[s];
    if (i >= 0) {                      // < 0: a row dropped by its producer (over the planned capacity)
      nbr[o * K + k] = i;
      nbr[(int64_t)i * K + (K - 1 - k)] = (int32_t)o;
    }
  }
}

// strided maps (stride-2 conv k3 / k1, pooling k2), driven from the INPUT side: an input at p can only feed the
// outputs o = p - off*s that lie on the coarse lattice -- 1, 2, 4 or 8 candidates for k3 (3.4 on average instead
// of 27 probes per output), exactly one for k2 (its parent) and for k1 (itself, if on the lattice).
// Lookups go to the OUTPUT coordinate map; nbr must be pre-filled with -1.
__global__ __launch_bounds__(256) void kernel_map_strided_kernel(const int32_t* __restrict__ in_coords, int64_t n_cap,
                                                                 const int32_t* __restrict__ n_dev, int s, int ksize,
                                                                 const uint64_t* __restrict__ out_keys,
                                                                 const int32_t* __restrict__ out_vals, int64_t cap,
                                                                 int32_t* __restrict__ nbr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= live_rows(n_cap, n_dev)) return;
  const int4 c = reinterpret_cast<const int4*>(in_coords)[i];
  const int s2 = 2 * s;
  auto fl = [s2](int p) { int q = p / s2; if ((p % s2 != 0) && (p < 0)) --q; return q * s2; };
  const int px = fl(c.y), py = fl(c.z), pz = fl(c.w);              // parent site
  const int ox = (c.y - px) / s, oy = (c.z - py) / s, oz = (c.w - pz) / s;   // 0 or 1: position inside the parent cell
  if (ksize == 2) {
    const int64_t slot = hash_find(out_keys, cap, coord_key(c.x, px, py, pz));
    if (slot >= 0 && out_vals[slot] >= 0) nbr[(int64_t)out_vals[slot] * 8 + (ox + 2 * oy + 4 * oz)] = (int32_t)i;
    return;
  }
  if (ksize == 1) {
    if (ox | oy | oz) return;
    const int64_t slot = hash_find(out_keys, cap, coord_key(c.x, px, py, pz));
    if (slot >= 0 && out_vals[slot] >= 0) nbr[out_vals[slot]] = (int32_t)i;
    return;
  }
  // ksize == 3: per axis the offset (in units of s) from the output to this input is 0 when the input sits on the
  // lattice, else +1 (output = parent) or -1 (output = parent + 2s)
  const int nx = ox ? 2 : 1, ny = oy ? 2 : 1, nz = oz ? 2 : 1;
  for (int a = 0; a < nx; ++a)
    for (int b = 0; b < ny; ++b)
      for (int d = 0; d < nz; ++d) {
        const int offx = ox ? (a == 0 ? 1 : -1) : 0, offy = oy ? (b == 0 ? 1 : -1) : 0, offz = oz ? (d == 0 ? 1 : -1) : 0;
        const int qx = c.y - offx * s, qy = c.z - offy * s, qz = c.w - offz * s;
        const int64_t slot = hash_find(out_keys, cap, coord_key(c.x, qx, qy, qz));
        if (slot >= 0 && out_vals[slot] >= 0)
          nbr[(int64_t)out_vals[slot] * 27 + ((offx + 1) + 3 * (offy + 1) + 9 * (offz + 1))] = (int32_t)i;
      }
}

// ================================================================================================================
// fused sparse convolution: output-stationary gather-GEMM on v_mfma_f32_32x32x2_f32
//   block = 256 threads = 4 waves arranged WAVES_M x WAVES_N; every wave owns TM x TN accumulator tiles of 32x32
//   (register blocking: one LDS read feeds TN resp. TM MFMAs); block tile BM x BN = (32*TM*WAVES_M) x (32*TN*WAVES_N).
//   A "stage" = (kernel offset k, 32-channel slice of Cin): gather A[BM][32] through the neighbour table, stage
//   W[k][32][BN], then 16*TM*TN MFMAs per wave.  Stages are software-pipelined: the global loads of stage s+1 are in
//   flight (in registers) while the MFMAs of stage s run.  Kernel offsets at which no row of the tile has a neighbour
//   are skipped (frequent with Morton-ordered rows on thin surfaces).  Small layers are split over kernel offsets
//   (grid.z) into fp32 partial slabs that a second kernel reduces in a fixed order -- deterministic, no atomics.
// ================================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;
constexpr int64_t CONV_PF2_ROWS = 0;      // output capacities below this run the two-stage prefetch variant (0: off until measured)

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return v > 0.0f ? v : 0.0f;
  if (act == 2) return v > 0.0f ? v : expm1f(v);
  return v;
}

// Generative transpose: child z (offset bits x fastest) of parent i lands in row 8 * i + morton_child(z).  Parent-major rows
// in Morton child order (x the most significant bit, as in the voxeliser's row order) keep every 64-row tile of the children
// a compact 4x4x4-voxel block whenever the parents are compact: what the gather-once convolution needs.
__device__ __forceinline__ int morton_child(int z) { return ((z & 1) << 2) | (z & 2) | ((z >> 2) & 1); }

struct ConvArgs {
  const float* in; int Cin;
  const int32_t* nbr; int K;
  const float* weight; int Cout;
  const float* scale; const float* shift; const float* residual; int act;
  float* out; int64_t no_cap; const int32_t* no_dev;
  int slices;        // > 1: generative transpose, slice z uses W[z] and writes row slices * i + morton_child(z)
  int splits;        // > 1: split over kernel offsets, raw partial sums go to slab[z]
  int k_per_split;
  float* slab;       // [splits][no_cap][Cout]
  // pre-split bf16 companions of the features: [rows + 1][C/8][3 planes][8] (hi/mid/lo of 8 channels = 48 B), the
  // extra last row is all zeros and stands in for missing neighbours (no select in the gather)
  const uint16_t* in_split; int64_t in_zero_row;
  uint16_t* out_split; int64_t out_zero_row;
  // f16x3: upper bounds of |features| / |weights| (device scalars) that fix the power-of-two operand scales, and the
  // running maximum of |outputs| for the consumers of this layer (atomicMax on the bit pattern; pre-zeroed by the host)
  const float* in_amax; const float* w_amax; float* out_amax;
  // pair-list mode (cnrma_sparse_conv_pairs_f16x3): the rows are (output, input) pairs grouped by kernel offset in runs
  // padded to 128 rows; tile_tap[row / 128] = the offset whose weights the run uses, w_taps = offsets in the image
  const int32_t* tile_tap; int w_taps;
  int xcd_tiles;     // > 0: number of row tiles; block b works on tile (b % 8) * ceil(tiles / 8) + b / 8 -- workgroups are dealt round-
                     // robin over the 8 XCDs, so each XCD (private L2) then owns one contiguous eighth of the rows
  int ablate;        // diagnostic kernels only (ABL = true; cnrma_debug_conv_tuning): bit 0 no MFMAs, 1 no A loads, 2 no B loads,
                     // 3 no LDS stores, 4 no barriers -- after the first stage; results are then meaningless, only the time counts
};

// exact 3-way split by truncation: h = top 8 significant bits of a, m = next 8, l = last 8 (a == h + m + l)
__device__ __forceinline__ void split3_trunc(float a, uint16_t& h, uint16_t& m, uint16_t& l) {
  const uint32_t uh = __float_as_uint(a) & 0xFFFF0000u;
  const float r1 = a - __uint_as_float(uh);
  const uint32_t um = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(um);
  h = (uint16_t)(uh >> 16); m = (uint16_t)(um >> 16); l = (uint16_t)(__float_as_uint(r2) >> 16);
}

// one output element into the split companion (column `col` of row `row`)
__device__ __forceinline__ void store_split(uint16_t* sp, int64_t row, int C, int col, float v) {
  uint16_t h, m, l;
  split3_trunc(v, h, m, l);
  uint16_t* q = sp + (row * (C >> 3) + (col >> 3)) * 24 + (col & 7);
  q[0] = h; q[8] = m; q[16] = l;
}

// FAST: Cin % 32 == 0 and Cout % 4 == 0 and Cout >= 4 -- every staging load is an unconditional 16-byte load with
// a clamped address and a select afterwards (no per-element branches: hipcc would wait vmcnt(0) inside each one and
// serialise the whole prefetch).
template <int WAVES_M, int WAVES_N, int TM, int TN, bool FAST, bool HAS_RES>
__global__ __launch_bounds__(256) void sparse_conv_mfma_kernel(ConvArgs p) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  constexpr int LDA = BM + 1;   // k-major A tile, odd stride: conflict-light scattered 4-byte writes
  constexpr int LDB = BN + 4;   // 16-B aligned rows for ds_write_b128
  constexpr int A_ITERS = BM * (BK / 4) / 256;
  constexpr int B_ITERS = BK * (BN / 4) / 256;
  static_assert(WAVES_M * WAVES_N == 4 && A_ITERS >= 1 && B_ITERS >= 1, "tile shape");
  __shared__ float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  __shared__ unsigned mask_s;

  const int64_t n_live = live_rows(p.no_cap, p.no_dev);
  const int64_t tile0 = (int64_t)blockIdx.x * BM;
  if (tile0 >= n_live) return;
  const int cout0 = blockIdx.y * BN;
  const int zs = blockIdx.z;
  const int Cin = p.Cin, Cout = p.Cout, K = p.K;
  // pair-list mode (cnrma_sparse_conv_pairs_f32): a K = 1 convolution over (output, input) pairs regrouped by kernel offset in runs
  // padded to 128 rows; the run's offset picks the weight slice
  const float* Wz = p.weight + (p.slices > 1 ? (int64_t)zs * K * Cin * Cout : 0) +
                    (p.tile_tap ? (int64_t)p.tile_tap[tile0 >> 7] * Cin * Cout : 0);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid / WAVES_N, wc = wid % WAVES_N;

  // ---- which kernel offsets have at least one neighbour in this tile (restricted to this block's k-range)
  int k_lo = 0, k_hi = K;
  if (p.splits > 1) { k_lo = zs * p.k_per_split; k_hi = min(K, k_lo + p.k_per_split); }
  unsigned mask = 0;
  const int rows_here = (int)min((int64_t)BM, n_live - tile0);
  if (p.nbr == nullptr) {
    mask = 1u;                                   // identity map (K == 1)
  } else {
    if (tid == 0) mask_s = 0;
    __syncthreads();
    unsigned local = 0;
    const int32_t* nb = p.nbr + tile0 * K;
    for (int i = tid; i < rows_here * K; i += 256) {
      const int k = i % K;
      if (k >= k_lo && k < k_hi && nb[i] >= 0) local |= 1u << k;
    }
    if (local) atomicOr(&mask_s, local);
    __syncthreads();
    mask = mask_s;
  }
  // neighbour rows of this thread's A_ITERS staging rows: for the current offset (src_cur) and, prefetched one
  // offset ahead, for the next active one (src_nxt) -- the gathers never wait on a dependent index load
  int32_t src_cur[A_ITERS], src_nxt[A_ITERS];
  auto load_src = [&](int k, int32_t* dst) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int row = (tid + i * 256) >> 3;
      int32_t v = -1;
      if (row < rows_here && k >= 0) v = p.nbr ? p.nbr[(tile0 + row) * K + k] : (int32_t)(tile0 + row);
      dst[i] = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  float4 ra[A_ITERS], rb[B_ITERS];
  unsigned a_ok = 0, b_ok = 0;   // FAST path: validity bits of the prefetched registers; the zero-select happens in
                                 // store_stage so that nothing consumes a loaded value before the MFMAs have run
  auto load_stage = [&](int k, int cin0, const int32_t* srcs) {
    a_ok = 0; b_ok = 0;
    const float* Wk = Wz + (int64_t)k * Cin * Cout;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, kc = idx & 7;
      const int32_t src = srcs[i];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int cin = cin0 + kc * 4;
      if constexpr (FAST) {
        v = *reinterpret_cast<const float4*>(p.in + (int64_t)(src < 0 ? 0 : src) * Cin + cin);
        a_ok |= (src >= 0 ? 1u : 0u) << i;
      } else if (src >= 0) {
        const float* q = p.in + (int64_t)src * Cin + cin;
        if (cin + 3 < Cin && (Cin & 3) == 0) {
          v = *reinterpret_cast<const float4*>(q);
        } else {
          if (cin < Cin) v.x = q[0];
          if (cin + 1 < Cin) v.y = q[1];
          if (cin + 2 < Cin) v.z = q[2];
          if (cin + 3 < Cin) v.w = q[3];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int idx = tid + i * 256;
      const int r = idx / (BN / 4), c4 = idx % (BN / 4);
      const int cin = cin0 + r, col = cout0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (FAST) {
        const bool ok = col < Cout;               // Cout % 4 == 0: a 4-column group is all in or all out
        v = *reinterpret_cast<const float4*>(Wk + (int64_t)cin * Cout + (ok ? col : 0));
        b_ok |= (ok ? 1u : 0u) << i;
      } else if (cin < Cin) {
        const float* q = Wk + (int64_t)cin * Cout + col;
        if (col + 3 < Cout && (Cout & 3) == 0) {
          v = *reinterpret_cast<const float4*>(q);
        } else {
          if (col < Cout) v.x = q[0];
          if (col + 1 < Cout) v.y = q[1];
          if (col + 2 < Cout) v.z = q[2];
          if (col + 3 < Cout) v.w = q[3];
        }
      }
      rb[i] = v;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, kc = idx & 7;
      float4 v = ra[i];
      if constexpr (FAST) {
        const bool ok = (a_ok >> i) & 1u;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      }
      As[(kc * 4 + 0) * LDA + row] = v.x;
      As[(kc * 4 + 1) * LDA + row] = v.y;
      As[(kc * 4 + 2) * LDA + row] = v.z;
      As[(kc * 4 + 3) * LDA + row] = v.w;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int idx = tid + i * 256;
      const int r = idx / (BN / 4), c4 = idx % (BN / 4);
      float4 v = rb[i];
      if constexpr (FAST) {
        const bool ok = (b_ok >> i) & 1u;
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      }
      *reinterpret_cast<float4*>(&Bs[r * LDB + c4 * 4]) = v;
    }
  };

  // ---- pipelined stage loop over (active k, cin slice)
  auto next_active = [&](int k) { const unsigned rest = mask & ~((2u << k) - 1u); return rest ? __ffs(rest) - 1 : -1; };
  int k = mask ? __ffs(mask) - 1 : -1;
  int cin0 = 0;
  load_src(k, src_cur);
  load_src(k >= 0 ? next_active(k) : -1, src_nxt);
  if (k >= 0) load_stage(k, 0, src_cur);
  while (k >= 0) {
    store_stage();
    __syncthreads();
    // next stage
    int nk = k, ncin = cin0 + BK;
    if (ncin >= Cin) {
      ncin = 0;
      nk = next_active(k);
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) src_cur[i] = src_nxt[i];
      if (nk >= 0) load_src(next_active(nk), src_nxt);      // index prefetch one offset ahead
    }
    if (nk >= 0) load_stage(nk, ncin, src_cur);   // global loads in flight during the MFMAs below
    const float* a_p = As + (lane >> 5) * LDA + wr * (32 * TM) + (lane & 31);
    const float* b_p = Bs + (lane >> 5) * LDB + wc * (32 * TN) + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) av[a] = a_p[kk * LDA + a * 32];
#pragma unroll
      for (int b = 0; b < TN; ++b) bv[b] = b_p[kk * LDB + b * 32];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    k = nk;
    cin0 = ncin;
  }

  // ---- epilogue.  D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool partial = p.splits > 1;
  float* dst = partial ? p.slab + (int64_t)zs * p.no_cap * Cout : p.out;
  const int child = p.slices > 1 ? morton_child(zs) : 0, row_step = p.slices > 1 ? p.slices : 1;   // see slice_out_row()
  const bool use_scale = !partial && p.scale != nullptr, use_shift = !partial && p.shift != nullptr;
  const int act = partial ? 0 : p.act;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = cout0 + wc * (32 * TN) + b * 32 + (lane & 31);
    const bool col_ok = col < Cout;
    const int colc = col_ok ? col : 0;
    const float sc = use_scale ? p.scale[colc] : 1.0f;
    const float sh = use_shift ? p.shift[colc] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int64_t row0 = tile0 + wr * (32 * TM) + a * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {                                 // 4 rows at a time: short live ranges
        float res[4];
        if constexpr (HAS_RES) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t row = row0 + q + 8 * rg;
            const int64_t rc = row < n_live ? row : n_live - 1;        // clamped, branch-free
            res[q] = p.residual[(rc * row_step + child) * Cout + colc];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t row = row0 + q + 8 * rg;
          float v = acc[a][b][rg * 4 + q];
          v = v * sc;
          v = v + sh;
          if constexpr (HAS_RES) v = v + res[q];
          v = apply_act(v, act);
          if (col_ok && row < n_live) dst[(row * row_step + child) * Cout + col] = v;
        }
      }
    }
  }
}

// ================================================================================================================
// fp32-grade convolution on the bf16 matrix cores ("bf16x6"): every fp32 operand is split exactly into three bf16
// pieces a = h + m + l (|a - h - m - l| <= 2^-27 |a|), and the product is rebuilt from the six partial products whose
// magnitude is >= 2^-16 |ab| (hh, hm, mh, hl, lh, mm; the dropped ml/lm/ll terms are <= 2^-23 |ab|), accumulated in
// fp32.  Six v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each):
// 2.67x fewer matrix-pipe cycles at the accuracy of an fp32 fma chain.  Weights are pre-split / pre-transposed once
// (cnrma_sparse_conv_prepare_weights); features are split while they are staged into LDS.
// LDS images: [plane][row][k] bf16, see lds_slot() below.
// ================================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// LDS images: [plane][row][32 k] 16-bit, unpadded 64-byte rows; the four 16-byte slots of a row are XORed with bits
// 2..3 of the row.  Fragment reads (ds_read_b128: 16-lane groups of rows {0-3,12-15,20-27} / {4-11,16-19,28-31}, same
// k-slot) then fall on 16 distinct 16-byte bank positions, and the staging stores (ds_write_b64 / _b128: consecutive
// lanes fill whole rows) stay a permutation of contiguous 128 bytes -- both conflict-free.  (The former padded 80-byte
// rows were conflict-free for the reads only: SQ_LDS_BANK_CONFLICT was 33 % of SQ_LDS_IDX_ACTIVE, profiles/.)
constexpr int LDK = 32;
__device__ __forceinline__ int lds_slot(int row, int slot) { return row * LDK + ((slot ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ void split3(const float4& v, uint2& h, uint2& m, uint2& l) {
  uint16_t hh[4], mm[4], ll[4];
  split3_trunc(v.x, hh[0], mm[0], ll[0]);
  split3_trunc(v.y, hh[1], mm[1], ll[1]);
  split3_trunc(v.z, hh[2], mm[2], ll[2]);
  split3_trunc(v.w, hh[3], mm[3], ll[3]);
  h = make_uint2((uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16));
  m = make_uint2((uint32_t)mm[0] | ((uint32_t)mm[1] << 16), (uint32_t)mm[2] | ((uint32_t)mm[3] << 16));
  l = make_uint2((uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16));
}

// ---- f16x3: a * 2^s = h + m with h, m fp16 (round to nearest): |a 2^s - h - m| <= 2^-22 |a 2^s|; the products hh, hm, mh
// are exact in fp32, the dropped mm term is <= 2^-22 |ab|.  The power-of-two scale (from an upper bound of the
// tensor's magnitude) keeps the largest operand at 2^13..2^14: no overflow, and every element down to 2^-17 of the
// largest keeps its 22 bits before m runs into fp16 subnormals.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float f16_scale_for(float amax) {
  if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);                                  // amax < 2^e
  int sh = 14 - e;
  sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  return ldexpf(1.0f, sh);
}

__device__ __forceinline__ void split2_f16(float x, uint16_t& h, uint16_t& m) {
  const _Float16 hh = (_Float16)x;
  const float r = x - (float)hh;                            // exact
  const _Float16 mm = (_Float16)r;
  h = __builtin_bit_cast(uint16_t, hh);
  m = __builtin_bit_cast(uint16_t, mm);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// (v * scale) -> fp16 pairs h, m in packed arithmetic: v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_add_f32 (3 VALU per value)
__device__ __forceinline__ void split2(const float4& v, float scale, uint2& h, uint2& m) {
  f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  a *= scale; b *= scale;
  const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
  const f32x2_t ra = a - __builtin_convertvector(ha, f32x2_t), rb = b - __builtin_convertvector(hb, f32x2_t);   // exact
  const f16x2_t ma = __builtin_convertvector(ra, f16x2_t), mb = __builtin_convertvector(rb, f16x2_t);
  h = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  m = make_uint2(__builtin_bit_cast(uint32_t, ma), __builtin_bit_cast(uint32_t, mb));
}

// A tensor's magnitude bound lives in AMAX_SLOTS words, one per 64-byte line: a block publishes ONE candidate (wave
// shuffle + LDS), into the slot its block index hashes to, and only when it beats the value currently visible (the
// maximum is monotone, so a stale read only costs a redundant atomic).  Thousands of same-address atomics per launch
// serialise at the memory side (measured: +1.1 ms per scene with one atomic per wave on a single word).
constexpr int AMAX_SLOTS = 64, AMAX_STRIDE = 16;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ float read_amax(const float* slots) {            // every lane returns the bound
  return wave_max(slots[(threadIdx.x & 63) * AMAX_STRIDE]);
}

// v >= 0 (uint order == float order); blockDim.x == 256, every thread of the block calls it; sh4: 4 floats of LDS
__device__ __forceinline__ void block_amax_publish(float* slots, float v, float* sh4) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
    const unsigned slot = (blockIdx.x + 7u * blockIdx.y + 13u * blockIdx.z) & (AMAX_SLOTS - 1);
    unsigned* d = reinterpret_cast<unsigned*>(slots + slot * AMAX_STRIDE);
    const unsigned bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, bits);
  }
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ in, int64_t n_cap,
                                                     const int32_t* __restrict__ n_dev, int C, float* __restrict__ out) {
  const int64_t total = live_rows(n_cap, n_dev) * C;
  float mx = 0.0f;
  for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; t < total; t += (int64_t)gridDim.x * blockDim.x * 4) {
    if (t + 3 < total && (C & 3) == 0) {
      const float4 v = *reinterpret_cast<const float4*>(in + t);
      mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
    } else {
      for (int j = 0; j < 4 && t + j < total; ++j) mx = fmaxf(mx, fabsf(in[t + j]));
    }
  }
  __shared__ float sh4[4];
  block_amax_publish(out, mx, sh4);
}

// feats[j][:] = feat[ray_j][:] * (w_j / w_div) for the point records {ray, step, weight bits, 0} (the feature half of
// neus_emit_rows_kernel of rma.hip, same arithmetic), in whatever order the records are given -- the static trace hands them
// over in VOXEL order (the voxeliser carried the 16-byte records through its representative selection and sort), so the
// sparse tensor's feature rows are written once, in place, with their magnitude bound: no [M, C] intermediate, no row gather,
// no absmax pass.  LPR lanes per row.
template <int LPR>
__global__ __launch_bounds__(256) void emit_features_kernel(const float* feat, const float* const* __restrict__ feat_ref, int C,
                                                            const int4* __restrict__ rec, int64_t n_cap,
                                                            const int32_t* __restrict__ n_dev, const float* __restrict__ w_div,
                                                            float* __restrict__ out, int out_stride, float* __restrict__ out_amax) {
  if (feat_ref != nullptr) feat = *feat_ref;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j = t / LPR;
  const int sub = (int)(t % LPR);
  float mx = 0.0f;
  if (j < live_rows(n_cap, n_dev)) {
    const int4 rc = rec[j];
    const float w = __int_as_float(rc.z);
    const float* f = feat + (int64_t)rc.x * C;
    float* q = out + j * out_stride;
    const bool scaled = w_div != nullptr;
    const float scale = scaled ? w / w_div[0] : 1.0f;
    const bool vec = ((C | out_stride) & 3) == 0 && ((((uintptr_t)out) | ((uintptr_t)feat)) & 15) == 0;
    if (vec) {
      for (int c = 4 * sub; c < C; c += 4 * LPR) {
        float4 x = *reinterpret_cast<const float4*>(f + c);
        if (scaled) { x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale; }
        *reinterpret_cast<float4*>(q + c) = x;
        mx = fmaxf(fmaxf(mx, fabsf(x.x)), fmaxf(fmaxf(fabsf(x.y), fabsf(x.z)), fabsf(x.w)));
      }
    } else {
      for (int c = sub; c < C; c += LPR) {
        const float x = scaled ? f[c] * scale : f[c];
        q[c] = x;
        mx = fmaxf(mx, fabsf(x));
      }
    }
  }
  if (out_amax != nullptr) {
    __shared__ float sh4[4];
    block_amax_publish(out_amax, mx, sh4);
  }
}

// prepared weight images pad Cout to a multiple of 128 with zero rows (the largest column tile): B-tile loads need no
// bounds check and no select
__host__ __device__ inline int conv_cout_padded(int Cout) { return (Cout + 127) & ~127; }

// element t of a prepared weight image -> (offset k, output column co, input channel cin).  Image order per plane:
// [K][Cin / 32][Cout_p][32] -- the 32-channel slice of ALL columns of one offset is contiguous (Cin % 32 == 0: the only
// case the MFMA kernels take; other channel counts keep the plain [K][Cout_p][Cin] order and are never read)
__device__ __forceinline__ void weight_image_coords(int64_t t, int Cin, int Cp, int* k, int* co, int* cin) {
  if ((Cin & (BK - 1)) == 0) {
    const int c32 = (int)(t & (BK - 1));
    const int64_t q = t / BK;
    *co = (int)(q % Cp);
    const int64_t q2 = q / Cp;
    const int ns = Cin / BK;
    *cin = (int)(q2 % ns) * BK + c32;
    *k = (int)(q2 / ns);
  } else {
    *cin = (int)(t % Cin);
    const int64_t q = t / Cin;
    *co = (int)(q % Cp);
    *k = (int)(q / Cp);
  }
}

// W fp32 [K][Cin][Cout] -> Wt fp16 [2 planes][K][Cin/32][Cout_p][32] scaled by f16_scale_for(*amax), + trailer float = *amax
__global__ __launch_bounds__(256) void prep_weights_f16_kernel(const float* __restrict__ w, uint16_t* __restrict__ wt, int K,
                                                               int Cin, int Cout, const float* __restrict__ amax) {
  const int Cp = conv_cout_padded(Cout);
  const int64_t total = (int64_t)K * Cin * Cp;
  const float am = read_amax(amax);
  const float sc = f16_scale_for(am);
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(wt + 2 * total) = am;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int cin, co, k;
    weight_image_coords(t, Cin, Cp, &k, &co, &cin);
    uint16_t hh, mm;
    split2_f16(co < Cout ? w[((int64_t)k * Cin + cin) * Cout + co] * sc : 0.0f, hh, mm);
    wt[t] = hh;
    wt[total + t] = mm;
  }
}

// fp32 [N][C] -> split companion [N+1][C/8][3][8] bf16 (+ the zero row at index n_cap); one lane per 8 channels
__global__ __launch_bounds__(256) void split_features_kernel(const float* __restrict__ in, int64_t n_cap,
                                                             const int32_t* __restrict__ n_dev, int C,
                                                             uint16_t* __restrict__ out) {
  const int64_t n = live_rows(n_cap, n_dev);
  const int G = C >> 3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < G) {                                           // zero row
    uint4* z = reinterpret_cast<uint4*>(out + (n_cap * G + t) * 24);
    z[0] = make_uint4(0, 0, 0, 0); z[1] = make_uint4(0, 0, 0, 0); z[2] = make_uint4(0, 0, 0, 0);
  }
  if (t >= n * G) return;
  const float4 a = *reinterpret_cast<const float4*>(in + t * 8);
  const float4 b = *reinterpret_cast<const float4*>(in + t * 8 + 4);
  uint2 h0, m0, l0, h1, m1, l1;
  split3(a, h0, m0, l0);
  split3(b, h1, m1, l1);
  uint4* q = reinterpret_cast<uint4*>(out + t * 24);
  q[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
  q[1] = make_uint4(m0.x, m0.y, m1.x, m1.y);
  q[2] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// W fp32 [K][Cin][Cout] -> Wt bf16 [3 planes][K][Cin/32][Cout_p][32]
__global__ __launch_bounds__(256) void prep_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int K,
                                                           int Cin, int Cout) {
  const int Cp = conv_cout_padded(Cout);
  const int64_t total = (int64_t)K * Cin * Cp;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int cin, co, k;
    weight_image_coords(t, Cin, Cp, &k, &co, &cin);
    const float a = co < Cout ? w[((int64_t)k * Cin + cin) * Cout + co] : 0.0f;
    uint16_t hh, mm, ll;
    split3_trunc(a, hh, mm, ll);
    uint16_t* o = reinterpret_cast<uint16_t*>(wt);
    o[t] = hh;
    o[total + t] = mm;
    o[2 * total + t] = ll;
  }
}

// W fp32 [K][Cin][Cout] -> Wt bf16 [K][Cin/32][Cout_p][32], round to nearest (MODE 2)
__global__ __launch_bounds__(256) void prep_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int K,
                                                                int Cin, int Cout) {
  const int Cp = conv_cout_padded(Cout);
  const int64_t total = (int64_t)K * Cin * Cp;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int cin, co, k;
    weight_image_coords(t, Cin, Cp, &k, &co, &cin);
    wt[t] = (__bf16)(co < Cout ? w[((int64_t)k * Cin + cin) * Cout + co] : 0.0f);
  }
}

// the image of the TRANSPOSED weights for dgrad, straight from W: Wd[k] = W[flip ? K - 1 - k : k]^T ([Cout] -> [Cin]), bf16
// [K][Cout/32][Cin_p][32] -- flip: a symmetric (same coordinates, odd kernel) map transposes by mirroring its offsets
__global__ __launch_bounds__(256) void prep_weights_bf16_t_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int K,
                                                                  int Cin, int Cout, int flip) {
  const int Cp = conv_cout_padded(Cin);
  const int64_t total = (int64_t)K * Cout * Cp;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int cin, co, k;
    weight_image_coords(t, Cout, Cp, &k, &co, &cin);              // co: a channel of grad_in (W's Cin), cin: of grad_out
    const int ks = flip ? K - 1 - k : k;
    wt[t] = (__bf16)(co < Cin ? w[((int64_t)ks * Cin + co) * Cout + cin] : 0.0f);
  }
}

// prefetch registers of the A operand + their staging code, one specialisation per input format (keeps the unused
// format's registers out of the kernel; plain members instead of lambda-captured arrays so they stay in VGPRs)
template <bool IN_SPLIT, int N, int NT = 256> struct AStageRegs;     // NT = threads that share the staging work

template <int N, int NT> struct AStageRegs<false, N, NT> {        // fp32 features: (row, 4 channels) per task, split at the LDS store
  float4 r[N];
  unsigned ok;
  __device__ __forceinline__ void load(const float* __restrict__ in, const uint16_t*, int64_t, int tid,
                                       const int32_t* srcs, int cin0, int, int Cin) {
    ok = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int kc = (tid + i * NT) & 7;
      const int32_t src = srcs[i];
      r[i] = *reinterpret_cast<const float4*>(in + (int64_t)(src < 0 ? 0 : src) * Cin + cin0 + kc * 4);
      ok |= (src >= 0 ? 1u : 0u) << i;
    }
  }
  template <int MODE>
  __device__ __forceinline__ void store(int tid, __bf16* a0, __bf16* a1, __bf16* a2, float a_scale) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = tid + i * NT;
      const int row = idx >> 3, kc = idx & 7;
      float4 v = r[i];
      const bool k = (ok >> i) & 1u;
      if constexpr (MODE == 0) {
        v.x = k ? v.x : 0.f; v.y = k ? v.y : 0.f; v.z = k ? v.z : 0.f; v.w = k ? v.w : 0.f;
        uint2 h, m, l;
        split3(v, h, m, l);
        const int o = lds_slot(row, kc >> 1) + (kc & 1) * 4;
        *reinterpret_cast<uint2*>(a0 + o) = h;
        *reinterpret_cast<uint2*>(a1 + o) = m;
        *reinterpret_cast<uint2*>(a2 + o) = l;
      } else if constexpr (MODE == 1) {
        // a missing neighbour was loaded from row 0 (clamped address): its scale is 0 instead of a select per value
        uint2 h, m;
        split2(v, k ? a_scale : 0.0f, h, m);
        const int o = lds_slot(row, kc >> 1) + (kc & 1) * 4;
        *reinterpret_cast<uint2*>(a0 + o) = h;
        *reinterpret_cast<uint2*>(a1 + o) = m;
      } else {
        // plain bf16 (autocast training): one round-to-nearest piece per value
        const float z = k ? 1.0f : 0.0f;
        const bf16x4_t q = {(__bf16)(v.x * z), (__bf16)(v.y * z), (__bf16)(v.z * z), (__bf16)(v.w * z)};
        const int o = lds_slot(row, kc >> 1) + (kc & 1) * 4;
        *reinterpret_cast<bf16x4_t*>(a0 + o) = q;
      }
    }
  }
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // first-class vector: stays in VGPRs across the loop
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int N, int NT> struct AStageRegs<true, N, NT> {         // pre-split companion: (row, 8 channels) per task, 3 x 16 B
  u32x4_t h[N], m[N], l[N];
  __device__ __forceinline__ void load(const float*, const uint16_t* __restrict__ in_split, int64_t zero_row, int tid,
                                       const int32_t* srcs, int cin0, int G8, int) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int g = (tid + i * NT) & 3;
      const int32_t src = srcs[i];
      const int64_t srow = src < 0 ? zero_row : (int64_t)src;             // missing neighbour -> the all-zero row
      const u32x4_t* q = reinterpret_cast<const u32x4_t*>(in_split + (srow * G8 + (cin0 >> 3) + g) * 24);
      h[i] = q[0]; m[i] = q[1]; l[i] = q[2];
    }
  }
  template <int MODE>
  __device__ __forceinline__ void store(int tid, __bf16* a0, __bf16* a1, __bf16* a2, float) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = tid + i * NT;
      const int row = idx >> 2, g = idx & 3;
      *reinterpret_cast<u32x4_t*>(a0 + lds_slot(row, g)) = h[i];
      *reinterpret_cast<u32x4_t*>(a1 + lds_slot(row, g)) = m[i];
      *reinterpret_cast<u32x4_t*>(a2 + lds_slot(row, g)) = l[i];
    }
  }
};

// MODE 0 = bf16x6 (3 planes, 6 products), MODE 1 = f16x3 (2 planes, 3 products, power-of-two operand scales),
// MODE 2 = bf16 (1 plane, 1 product: the autocast training precision -- bf16 operands, fp32 accumulation)
// PF = global-load stages in flight beyond the one being multiplied: 1 = the loads of stage s + 1 fly during the MFMAs of
// stage s (every variant); 2 = two register sets, stages s + 1 and s + 2 in flight (short layers: a block of a < 16 k-row
// layer walks a chain of 16-48 stages with at most 3 blocks per CU -- each stage then costs one whole L2 round trip,
// not its 384 MFMA cycles; the second set costs 24-32 registers, irrelevant at that occupancy)
template <int WAVES_M, int WAVES_N, int TM, int TN, bool HAS_RES, bool IN_SPLIT, int MODE, int PF = 1, bool ABL = false>
__global__ __launch_bounds__(256, 2) void sparse_conv_bf16x6_kernel(ConvArgs p, const __bf16* __restrict__ wt) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  constexpr int NP = MODE == 0 ? 3 : (MODE == 1 ? 2 : 1);
  static_assert(MODE == 0 || !IN_SPLIT, "companions exist for bf16x6 only");
  // staging tasks per stage: fp32 input = (row, 4 channels) -> 8 per row; pre-split input = (row, 8 channels) -> 4 per row
  constexpr int ROW_SHIFT = IN_SPLIT ? 2 : 3;
  constexpr int A_ITERS = (BM << ROW_SHIFT) / 256;
  constexpr int B_CHUNKS = NP * BN * (BK / 8);          // 16-byte (8 x 16-bit) chunks over the planes
  constexpr int B_ITERS = (B_CHUNKS + 255) / 256;
  static_assert(WAVES_M * WAVES_N == 4 && A_ITERS >= 1 && B_ITERS >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) __bf16 As[NP][BM * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[NP][BN * LDK];
  __shared__ unsigned mask_s;

  const int64_t n_live = live_rows(p.no_cap, p.no_dev);
  int64_t tile_id = blockIdx.x;
  if (p.xcd_tiles > 0) {
    const int per = (p.xcd_tiles + 7) >> 3;
    tile_id = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile_id >= p.xcd_tiles) return;
  }
  const int64_t tile0 = tile_id * BM;
  if (tile0 >= n_live) return;
  const int cout0 = blockIdx.y * BN;
  const int zs = blockIdx.z;
  const int Cin = p.Cin, Cout = p.Cout, K = p.K;
  const int Cout_p = conv_cout_padded(Cout);     // prepared images pad Cout with zero rows: no bounds check on B
  const int64_t plane_elems = (int64_t)(p.slices > 1 ? p.slices : 1) * (p.tile_tap ? p.w_taps : K) * Cin * Cout_p;
  const __bf16* Wz = wt + (p.slices > 1 ? (int64_t)zs * K * Cin * Cout_p : 0) +
                     (p.tile_tap ? (int64_t)p.tile_tap[tile0 >> 7] * Cin * Cout_p : 0);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid / WAVES_N, wc = wid % WAVES_N;
  float a_scale = 1.0f, out_scale = 1.0f;
  if constexpr (MODE == 1) {
    a_scale = f16_scale_for(read_amax(p.in_amax));
    out_scale = 1.0f / (a_scale * f16_scale_for(*p.w_amax));          // powers of two: exact
  }

  int k_lo = 0, k_hi = K;
  if (p.splits > 1) { k_lo = zs * p.k_per_split; k_hi = min(K, k_lo + p.k_per_split); }
  unsigned mask = 0;
  const int rows_here = (int)min((int64_t)BM, n_live - tile0);
  if (p.nbr == nullptr) {
    mask = 1u;                                   // identity map (K == 1)
  } else {
    if (tid == 0) mask_s = 0;
    __syncthreads();
    unsigned local = 0;
    const int32_t* nb = p.nbr + tile0 * K;
    const int kr = k_hi - k_lo;                  // only this block's offsets are looked at (a split block owns 2-3 of 27)
    for (int i = tid; i < rows_here * kr; i += 256) {
      const int row = i / kr, k = k_lo + (i - row * kr);
      if (nb[row * K + k] >= 0) local |= 1u << k;
    }
    if (local) atomicOr(&mask_s, local);
    __syncthreads();
    mask = mask_s;
  }
  // neighbour rows of this thread's A_ITERS staging rows: for the current offset (src_cur) and, prefetched one
  // offset ahead, for the next active one (src_nxt) -- the gathers never wait on a dependent index load
  int32_t src_cur[A_ITERS], src_nxt[A_ITERS];
  auto load_src = [&](int k, int32_t* dst) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int row = (tid + i * 256) >> ROW_SHIFT;
      int32_t v = -1;
      if (row < rows_here && k >= 0) v = p.nbr ? p.nbr[(tile0 + row) * K + k] : (int32_t)(tile0 + row);
      dst[i] = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  typedef AStageRegs<IN_SPLIT, A_ITERS> ARegs;
  ARegs areg;
  u32x4_t rb[B_ITERS];                            // first-class vectors (HIP's uint4 struct arrays end up in scratch)
  const int G8 = Cin >> 3;
  const float* in_f32 = p.in;
  const uint16_t* in_sp = p.in_split;
  const int64_t in_zero = p.in_zero_row;
  auto load_stage = [&](ARegs& ar, u32x4_t* br, int k, int cin0, const int32_t* srcs, int skip = 0) {
    if (!ABL || !(skip & 2)) ar.load(in_f32, in_sp, in_zero, tid, srcs, cin0, G8, Cin);
    if (ABL && (skip & 4)) return;
    // image order [offset][32-channel slice][output column][32]: the B tile of a stage is ONE contiguous run of whole
    // 128-byte lines (with [offset][column][Cin] every stage fetched the 64-byte half of a line and left the other half
    // to the next stage: the B operand moved twice through the L2 -> L1 path, which is what bounds this kernel)
    const __bf16* Wk = Wz + ((int64_t)k * (Cin / BK) + cin0 / BK) * Cout_p * BK;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      int idx = tid + i * 256;                         // over [plane][row][chunk]; the lane part is loop invariant
      idx = idx < B_CHUNKS ? idx : 0;
      const int chunk = idx & 3, row = (idx >> 2) % BN, pl = idx / (4 * BN);
      br[i] = *reinterpret_cast<const u32x4_t*>(Wk + pl * plane_elems + (int64_t)(cout0 + row) * BK + chunk * 8);
    }
  };
  auto store_stage = [&](const ARegs& ar, const u32x4_t* br) {
    ar.template store<MODE>(tid, &As[0][0], &As[1][0], &As[NP - 1][0], a_scale);
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int idx = tid + i * 256;
      if (idx < B_CHUNKS) {
        const int chunk = idx & 3, row = (idx >> 2) % BN, pl = idx / (4 * BN);
        *reinterpret_cast<u32x4_t*>(&Bs[pl][lds_slot(row, chunk)]) = br[i];
      }
    }
  };
  auto mfma_stage = [&]() {
    const int a_row = wr * (32 * TM) + (lane & 31), b_row = wc * (32 * TN) + (lane & 31), fhalf = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      if constexpr (MODE == 0) {
        bf16x8_t af[TM][3], bf[TN][3];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) af[a][pl] = *reinterpret_cast<const bf16x8_t*>(&As[pl][lds_slot(a_row + a * 32, (ks >> 3) + fhalf)]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bf[b][pl] = *reinterpret_cast<const bf16x8_t*>(&Bs[pl][lds_slot(b_row + b * 32, (ks >> 3) + fhalf)]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            f32x16 c = acc[a][b];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], c, 0, 0, 0);   // l*h
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], c, 0, 0, 0);   // h*l
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], c, 0, 0, 0);   // m*m
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], c, 0, 0, 0);   // m*h
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], c, 0, 0, 0);   // h*m
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], c, 0, 0, 0);   // h*h
            acc[a][b] = c;
          }
      } else if constexpr (MODE == 2) {
        bf16x8_t af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const bf16x8_t*>(&As[0][lds_slot(a_row + a * 32, (ks >> 3) + fhalf)]);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const bf16x8_t*>(&Bs[0][lds_slot(b_row + b * 32, (ks >> 3) + fhalf)]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
      } else {
        f16x8_t af[TM][2], bf[TN][2];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) af[a][pl] = *reinterpret_cast<const f16x8_t*>(&As[pl][lds_slot(a_row + a * 32, (ks >> 3) + fhalf)]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) bf[b][pl] = *reinterpret_cast<const f16x8_t*>(&Bs[pl][lds_slot(b_row + b * 32, (ks >> 3) + fhalf)]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            f32x16 c = acc[a][b];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][1], bf[b][0], c, 0, 0, 0);    // m*h
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][0], bf[b][1], c, 0, 0, 0);    // h*m
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a][0], bf[b][0], c, 0, 0, 0);    // h*h
            acc[a][b] = c;
          }
      }
    }
  };

  auto next_active = [&](int k) { const unsigned rest = mask & ~((2u << k) - 1u); return rest ? __ffs(rest) - 1 : -1; };
  int k = mask ? __ffs(mask) - 1 : -1;
  int cin0 = 0;
  load_src(k, src_cur);
  load_src(k >= 0 ? next_active(k) : -1, src_nxt);
  if constexpr (ABL) {
    // diagnostic build of the PF = 1 loop: stage components switched off by p.ablate after the first stage
    const int ab = p.ablate;
    if (k >= 0) load_stage(areg, rb, k, 0, src_cur);
    bool first = true;
    while (k >= 0) {
      if (first || !(ab & 8)) store_stage(areg, rb);
      if (first || !(ab & 16)) __syncthreads();
      int nk = k, ncin = cin0 + BK;
      if (ncin >= Cin) {
        ncin = 0;
        nk = next_active(k);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) src_cur[i] = src_nxt[i];
        if (nk >= 0) load_src(next_active(nk), src_nxt);
      }
      if (nk >= 0) load_stage(areg, rb, nk, ncin, src_cur, ab);
      if (first || !(ab & 1)) mfma_stage();
      if (first || !(ab & 16)) __syncthreads();
      k = nk;
      cin0 = ncin;
      first = false;
    }
  } else if constexpr (PF == 1) {
    if (k >= 0) load_stage(areg, rb, k, 0, src_cur);
    while (k >= 0) {
      store_stage(areg, rb);
      __syncthreads();
      int nk = k, ncin = cin0 + BK;
      if (ncin >= Cin) {
        ncin = 0;
        nk = next_active(k);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) src_cur[i] = src_nxt[i];
        if (nk >= 0) load_src(next_active(nk), src_nxt);
      }
      if (nk >= 0) load_stage(areg, rb, nk, ncin, src_cur);
      mfma_stage();
      __syncthreads();
      k = nk;
      cin0 = ncin;
    }
  } else {
    // two register sets: while stage s is multiplied, the loads of s + 1 AND s + 2 are in flight.  (k, cin0) is the LOAD
    // frontier here; the stages are multiplied in the order they were loaded.  Needs Cin >= 2 BK: the neighbour indices of
    // an offset are fetched one offset ahead of the frontier, i.e. at least two stages before their gathers are issued.
    ARegs areg2;
    u32x4_t rb2[B_ITERS];
    int todo = __popc(mask) * (Cin / BK);            // stages not yet multiplied
    auto advance = [&]() {
      cin0 += BK;
      if (cin0 >= Cin) {
        cin0 = 0;
        k = next_active(k);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) src_cur[i] = src_nxt[i];
        if (k >= 0) load_src(next_active(k), src_nxt);
      }
    };
    if (k >= 0) { load_stage(areg, rb, k, cin0, src_cur); advance(); }
    if (k >= 0) { load_stage(areg2, rb2, k, cin0, src_cur); advance(); }
    while (todo > 0) {
      store_stage(areg, rb);
      __syncthreads();
      if (k >= 0) { load_stage(areg, rb, k, cin0, src_cur); advance(); }
      mfma_stage();
      __syncthreads();
      if (--todo == 0) break;
      store_stage(areg2, rb2);
      __syncthreads();
      if (k >= 0) { load_stage(areg2, rb2, k, cin0, src_cur); advance(); }
      mfma_stage();
      __syncthreads();
      --todo;
    }
  }

  if (p.out_split && p.splits <= 1 && blockIdx.x == 0 && blockIdx.y == 0 && zs == 0) {
    for (int i = tid; i < (Cout >> 3) * 24; i += 256) p.out_split[p.out_zero_row * (Cout >> 3) * 24 + i] = 0;
  }
  const bool partial = p.splits > 1;
  float* dst = partial ? p.slab + (int64_t)zs * p.no_cap * Cout : p.out;
  const int child = p.slices > 1 ? morton_child(zs) : 0, row_step = p.slices > 1 ? p.slices : 1;   // see slice_out_row()
  const bool use_scale = !partial && p.scale != nullptr, use_shift = !partial && p.shift != nullptr;
  const int act = partial ? 0 : p.act;
  float mx = 0.0f;                                                     // largest |output| of this lane (f16x3 consumers)
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = cout0 + wc * (32 * TN) + b * 32 + (lane & 31);
    const bool col_ok = col < Cout;
    const int colc = col_ok ? col : 0;
    const float sc = use_scale ? p.scale[colc] : 1.0f;
    const float sh = use_shift ? p.shift[colc] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int64_t row0 = tile0 + wr * (32 * TM) + a * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {                                 // 4 rows at a time: short live ranges
        float res[4];
        if constexpr (HAS_RES) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t row = row0 + q + 8 * rg;
            const int64_t rc = row < n_live ? row : n_live - 1;        // clamped, branch-free
            res[q] = p.residual[(rc * row_step + child) * Cout + colc];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t row = row0 + q + 8 * rg;
          float v = acc[a][b][rg * 4 + q];
          if constexpr (MODE == 1) v = v * out_scale;                  // undo the operand scales (a power of two)
          v = v * sc;
          v = v + sh;
          if constexpr (HAS_RES) v = v + res[q];
          v = apply_act(v, act);
          if (col_ok && row < n_live) {
            dst[(row * row_step + child) * Cout + col] = v;
            if (!partial && p.out_split) store_split(p.out_split, row * row_step + child, Cout, col, v);
            mx = fmaxf(mx, fabsf(v));
          }
        }
      }
    }
  }
  if (!partial && p.out_amax != nullptr) {
    __syncthreads();                                                   // the A image is no longer read: reuse 16 bytes
    block_amax_publish(p.out_amax, mx, reinterpret_cast<float*>(&As[0][0]));
  }
}

// reduce the split-K slabs in a fixed order and apply the fused epilogue
__global__ __launch_bounds__(256) void conv_reduce_kernel(ConvArgs p) {
  const int64_t n_live = live_rows(p.no_cap, p.no_dev);
  if (p.out_split && blockIdx.x == 0)
    for (int i = threadIdx.x; i < (p.Cout >> 3) * 24; i += 256) p.out_split[p.out_zero_row * (p.Cout >> 3) * 24 + i] = 0;
  const int64_t total = n_live * p.Cout;
  const int64_t slab_stride = p.no_cap * p.Cout;
  float mx = 0.0f;
  for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; t < total; t += (int64_t)gridDim.x * blockDim.x * 4) {
    if ((p.Cout & 3) == 0) {
      // every load of this element group is issued before the first is used, none sits behind a branch (a load behind a
      // branch or in a loop of unknown length gets its own s_waitcnt vmcnt(0): the slabs were read one round trip at a time
      // and the epilogue operands one scalar at a time -- ~19 dependent round trips, the 9 us this launch took whatever
      // its size); the slabs are still added in slab order
      const int col = (int)(t % p.Cout);
      const float* dummy = p.slab + t;                                        // a valid 16-byte aligned address
      const float4 sc4 = *reinterpret_cast<const float4*>(p.scale ? p.scale + col : dummy);
      const float4 sh4 = *reinterpret_cast<const float4*>(p.shift ? p.shift + col : dummy);
      const float4 rs4 = *reinterpret_cast<const float4*>(p.residual ? p.residual + t : dummy);
      float4 s = *reinterpret_cast<const float4*>(p.slab + t);
      int z = 1;
      for (; z + 3 < p.splits; z += 4) {
        const float* b = p.slab + (int64_t)z * slab_stride + t;
        const float4 q0 = *reinterpret_cast<const float4*>(b), q1 = *reinterpret_cast<const float4*>(b + slab_stride),
                     q2 = *reinterpret_cast<const float4*>(b + 2 * slab_stride), q3 = *reinterpret_cast<const float4*>(b + 3 * slab_stride);
        s.x += q0.x; s.y += q0.y; s.z += q0.z; s.w += q0.w;
        s.x += q1.x; s.y += q1.y; s.z += q1.z; s.w += q1.w;
        s.x += q2.x; s.y += q2.y; s.z += q2.z; s.w += q2.w;
        s.x += q3.x; s.y += q3.y; s.z += q3.z; s.w += q3.w;
      }
      {                                                                      // the last <= 3 slabs: clamped (re-read), added under a uniform test
        const int z1 = min(z + 1, p.splits - 1), z2 = min(z + 2, p.splits - 1);
        const float4 q0 = *reinterpret_cast<const float4*>(p.slab + (int64_t)min(z, p.splits - 1) * slab_stride + t),
                     q1 = *reinterpret_cast<const float4*>(p.slab + (int64_t)z1 * slab_stride + t),
                     q2 = *reinterpret_cast<const float4*>(p.slab + (int64_t)z2 * slab_stride + t);
        if (z < p.splits) { s.x += q0.x; s.y += q0.y; s.z += q0.z; s.w += q0.w; }
        if (z + 1 < p.splits) { s.x += q1.x; s.y += q1.y; s.z += q1.z; s.w += q1.w; }
        if (z + 2 < p.splits) { s.x += q2.x; s.y += q2.y; s.z += q2.z; s.w += q2.w; }
      }
      float v[4] = {s.x, s.y, s.z, s.w};
      const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, rsv[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = v[j];
        if (p.scale) x = x * scv[j];
        if (p.shift) x = x + shv[j];
        if (p.residual) x = x + rsv[j];
        v[j] = apply_act(x, p.act);
        mx = fmaxf(mx, fabsf(v[j]));
        if (p.out_split) store_split(p.out_split, t / p.Cout, p.Cout, col + j, v[j]);
      }
      *reinterpret_cast<float4*>(p.out + t) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int j = 0; j < 4 && t + j < total; ++j) {
        float x = 0.0f;
        for (int z = 0; z < p.splits; ++z) x += p.slab[(int64_t)z * slab_stride + t + j];
        const int col = (int)((t + j) % p.Cout);
        if (p.scale) x = x * p.scale[col];
        if (p.shift) x = x + p.shift[col];
        if (p.residual) x = x + p.residual[t + j];
        x = apply_act(x, p.act);
        mx = fmaxf(mx, fabsf(x));
        p.out[t + j] = x;
      }
    }
  }
  if (p.out_amax != nullptr) {
    __shared__ float sh4[4];
    block_amax_publish(p.out_amax, mx, sh4);
  }
}

// how many kernel-offset groups a layer is split into: enough blocks to fill 256 CUs a few times over
int choose_splits(int64_t rows, int Cout, int K, int bm, int bn, size_t ws_bytes) {
  if (K <= 1) return 1;
  const int64_t tiles = ceil_div(rows, bm) * ceil_div(Cout, bn);
  if (tiles >= 384) return 1;
  int s = (int)ceil_div(768, tiles);
  if (s > K) s = K;
  const size_t per = (size_t)rows * Cout * sizeof(float);
  if (per == 0) return 1;
  const int64_t fit = (int64_t)(ws_bytes / per);
  if (s > fit) s = (int)fit;
  return s < 2 ? 1 : s;
}

#ifdef CNRMA_EXPERIMENTS
#include "sparse_exp_ws.inc"
#endif

constexpr int CONV_XCD_ORDER = 0;      // product default of the XCD-aware tile order of the stage kernel -- until measured
#ifdef CNRMA_EXPERIMENTS
constexpr int CONV_WS_SLOTS = 0;       // product default of the warp-specialised kernel's ring (0: stage kernel) -- until measured
#endif

// Debug / A-B switches of the convolution launcher (cnrma_debug_conv_tuning: scripts/conv_sweep.py and the tests that force a
// variant).  Product code never changes them; -1 = the launcher's own choice.
struct ConvTune { int shape = -1; int splits = -1; int pf = -1; int ablate = 0; int ws = -1; int xcd = -1; int go = -1; int nb = -1; };   // ws: LDS ring slots of the warp-specialised kernel (0 = stage kernel)
#ifdef CNRMA_EXPERIMENTS
static ConvTune g_conv_tune;                 // libcnrma_hip_exp.so only: written by cnrma_debug_conv_tuning
#define CNRMA_CONV_TUNE g_conv_tune
#else
// the product library has no tuning state at all: every launcher decision is a pure function of its arguments (no
// global mutable state), the experimental kernels and cnrma_debug_conv_tuning are not compiled in
static constexpr ConvTune k_conv_tune{};
#define CNRMA_CONV_TUNE k_conv_tune
#endif

enum ConvShape { T128x128, T128x64, T64x64, T128x32, T64x128, T256x128, T256x64, N_CONV_SHAPES };   // the last two: warp-specialised kernel only
struct ConvPlan { int shape, bm, bn, splits, k_per_split, pf; };

// tile shape, split count over the kernel offsets and prefetch depth of one launch: a pure function of the layer's
// sizes (the CAPACITY of the output, not its live row count), so a captured launch sequence replays the same kernels
ConvPlan plan_conv(int64_t no_cap, int Cin, int Cout, int K, int mode, bool six, int slices, bool has_ws, size_t ws_bytes) {
  static const int bms[] = {128, 128, 64, 128, 64, 256, 256}, bns[] = {128, 64, 64, 32, 128, 128, 64};
  // tile choice, measured per layer class and precision on MI355X at the ScanNet shape: f16x3 tiles
  // need fewer registers and less LDS (4-7 blocks per CU), which moves the optimum to 64-row tiles almost everywhere
  int sh;
  if (Cout <= 32) sh = T128x32;
  else if (six && mode >= 1) {
    if (Cout >= 128) sh = no_cap >= 16384 && no_cap < 40000 ? T128x128 : (no_cap < 1000 && Cout < 256 ? T64x64 : T64x128);
    else sh = no_cap >= 200000 && Cin > 32 ? T128x64 : T64x64;
  }
  else if (six && no_cap < 1000 && Cout >= 256) sh = T64x128;
  else if (no_cap < 4000) sh = T64x64;
  else if (no_cap < 16384) sh = six ? T128x64 : T64x64;
  else if (Cout >= 128) sh = T128x128;
  else if (six && Cin <= 32) sh = T64x64;
  else sh = T128x64;
  const ConvTune t = CNRMA_CONV_TUNE;
  if (t.shape >= 0 && t.shape < N_CONV_SHAPES && (six || t.shape < T64x128) && (t.shape < T256x128 || (six && mode == 1 && t.ws >= 2)))
    sh = t.shape;
  ConvPlan pl{sh, bms[sh], bns[sh], 1, K, 1};
  if (slices == 1 && has_ws) {
    pl.splits = choose_splits(no_cap, Cout, K, pl.bm, pl.bn, ws_bytes);
    if (t.splits > 0) {
      pl.splits = t.splits > K ? K : t.splits;
      const size_t per = (size_t)no_cap * Cout * sizeof(float);
      if (per > 0 && (size_t)pl.splits * per > ws_bytes) pl.splits = (int)(ws_bytes / per);
    }
    pl.k_per_split = (int)ceil_div(K, pl.splits > 0 ? pl.splits : 1);
    pl.splits = (int)ceil_div(K, pl.k_per_split);
    if (pl.splits < 2) { pl.splits = 1; pl.k_per_split = K; }
  }
  // two stages of loads in flight where a block is a chain of latency-bound stages: short f16x3 layers (Cin >= 64: the
  // neighbour indices must be two stages ahead of their gathers)
  const bool pf2_ok = six && mode == 1 && Cin >= 2 * BK && sh != T128x32 && sh != T128x128 && sh < T256x128;    // 128x128 would spill
  if (pf2_ok && K > 1) pl.pf = no_cap < CONV_PF2_ROWS ? 2 : 1;
  if (t.pf > 0 && pf2_ok) pl.pf = t.pf >= 2 ? 2 : 1;
  return pl;
}

int launch_conv(const float* in, int Cin, const int32_t* nbr, int K, const float* weight, int Cout, const float* scale,
                const float* shift, const float* residual, int act, float* out, int64_t no_cap, const int32_t* no_dev,
                int slices, void* workspace, size_t ws_bytes, hipStream_t st, const void* weight_split = nullptr,
                const void* in_split = nullptr, int64_t in_zero_row = 0, void* out_split = nullptr,
                int64_t out_zero_row = 0, int mode = 0, const float* in_amax = nullptr, float* out_amax = nullptr,
                const int32_t* tile_tap = nullptr, int w_taps = 0) {
  if (Cin <= 0 || Cout <= 0 || K <= 0 || K > 27 || no_cap <= 0) return CNRMA_EINVAL;
  if (out_split != nullptr && (Cout % 8 != 0 || weight_split == nullptr)) return CNRMA_EINVAL;
  if (in_split != nullptr && (Cin % 32 != 0 || weight_split == nullptr)) return CNRMA_EINVAL;
  ConvArgs p{in, Cin, nbr, K, weight, Cout, scale, shift, residual, act, out, no_cap, no_dev, slices, 1, K,
             reinterpret_cast<float*>(workspace), reinterpret_cast<const uint16_t*>(in_split), in_zero_row,
             reinterpret_cast<uint16_t*>(out_split), out_zero_row, in_amax, nullptr, out_amax, tile_tap, w_taps};
  if (mode == 2 && (weight_split == nullptr || in_split != nullptr || out_split != nullptr)) return CNRMA_EINVAL;
  if (mode == 1) {
    if (weight_split == nullptr || in_amax == nullptr || in_split != nullptr || out_split != nullptr) return CNRMA_EINVAL;
    p.w_amax = reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(weight_split) +
                                              2 * (int64_t)(slices > 1 ? slices : 1) * (tile_tap ? w_taps : K) * Cin *
                                                  conv_cout_padded(Cout));
  }
  const bool six = weight_split != nullptr && Cin % 32 == 0;
  const ConvPlan pl = plan_conv(no_cap, Cin, Cout, K, mode, six, slices, workspace != nullptr, ws_bytes);
  const int shape = pl.shape, bm = pl.bm, bn = pl.bn;
  p.ablate = CNRMA_CONV_TUNE.ablate;
  p.splits = pl.splits;
  p.k_per_split = pl.k_per_split;
  dim3 grid((unsigned)ceil_div(no_cap, bm), (unsigned)ceil_div(Cout, bn), (unsigned)(slices > 1 ? slices : p.splits));
  // XCD-aware tile order (stage kernel on prepared weights; the pair-list runs keep their tile_tap order)
  const int xcd = CNRMA_CONV_TUNE.xcd >= 0 ? CNRMA_CONV_TUNE.xcd : CONV_XCD_ORDER;
  if (xcd && weight_split != nullptr && Cin % 32 == 0 && tile_tap == nullptr && grid.x >= 64) {
    p.xcd_tiles = (int)grid.x;
    grid.x = (grid.x + 7u) / 8u * 8u;
  }
  const bool fast = (Cin % 32 == 0) && (Cout % 4 == 0);
  const bool has_res = residual != nullptr && p.splits == 1;   // split layers add the residual in the reduce kernel
  if (weight_split != nullptr && Cin % 32 == 0) {
    const __bf16* wt = reinterpret_cast<const __bf16*>(weight_split);
#ifdef CNRMA_EXPERIMENTS            // the diagnostic (ablation) instantiation of the stage kernel
#define CNRMA_CONV6_ABL(WM, WN, TM_, TN_)                                                                          \
    else if (mode == 1 && !has_res && CNRMA_CONV_TUNE.ablate != 0)                                                 \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, false, false, 1, 1, true>), grid, dim3(256), 0, st, p, wt);
#else
#define CNRMA_CONV6_ABL(WM, WN, TM_, TN_)
#endif
#define CNRMA_CONV6_LAUNCH(WM, WN, TM_, TN_)                                                                       \
  do {                                                                                                             \
    if (mode == 2 && has_res)                                                                                      \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, true, false, 2>), grid, dim3(256), 0, st, p, wt);  \
    else if (mode == 2)                                                                                            \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, false, false, 2>), grid, dim3(256), 0, st, p, wt); \
    CNRMA_CONV6_ABL(WM, WN, TM_, TN_)                                                                              \
    else if (mode == 1 && has_res && pl.pf == 2)                                                                   \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, true, false, 1, (TM_ * TN_ <= 2 ? 2 : 1)>), grid, dim3(256), 0, st, p, wt);  \
    else if (mode == 1 && pl.pf == 2)                                                                              \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, false, false, 1, (TM_ * TN_ <= 2 ? 2 : 1)>), grid, dim3(256), 0, st, p, wt); \
    else if (mode == 1 && has_res)                                                                                      \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, true, false, 1>), grid, dim3(256), 0, st, p, wt);  \
    else if (mode == 1)                                                                                            \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, false, false, 1>), grid, dim3(256), 0, st, p, wt); \
    else if (has_res && in_split)                                                                                  \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, true, true, 0>), grid, dim3(256), 0, st, p, wt);   \
    else if (has_res)                                                                                              \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, true, false, 0>), grid, dim3(256), 0, st, p, wt);  \
    else if (in_split)                                                                                             \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, false, true, 0>), grid, dim3(256), 0, st, p, wt);  \
    else                                                                                                           \
      hipLaunchKernelGGL((sparse_conv_bf16x6_kernel<WM, WN, TM_, TN_, false, false, 0>), grid, dim3(256), 0, st, p, wt); \
  } while (0)
#ifdef CNRMA_EXPERIMENTS
    const int ws_slots = mode == 1 && in_split == nullptr && shape != T128x32 ? (CNRMA_CONV_TUNE.ws >= 0 ? CNRMA_CONV_TUNE.ws : CONV_WS_SLOTS) : 0;
    if (ws_slots >= 2 && CNRMA_CONV_TUNE.ablate == 0) {
      // warp-specialised kernel: 8 waves, dynamic LDS = ring + counters (above 64 KB the limit is raised per kernel)
      const int rc = launch_conv_ws(shape, ws_slots, has_res, grid, p, wt, st);
      if (rc != 0) return rc;
    } else
#endif
    switch (shape) {
      case T128x128: CNRMA_CONV6_LAUNCH(2, 2, 2, 2); break;
      case T128x64: CNRMA_CONV6_LAUNCH(4, 1, 1, 2); break;
      case T64x64: CNRMA_CONV6_LAUNCH(2, 2, 1, 1); break;
      case T128x32: CNRMA_CONV6_LAUNCH(4, 1, 1, 1); break;
      case T64x128: CNRMA_CONV6_LAUNCH(2, 2, 1, 2); break;
      default: return CNRMA_EINVAL;
    }
#undef CNRMA_CONV6_LAUNCH
#undef CNRMA_CONV6_ABL
    if (p.splits > 1) {
      int64_t blocks = ceil_div(no_cap * Cout / 4 + 1, 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    }
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
#define CNRMA_CONV_LAUNCH(WM, WN, TM_, TN_)                                                                        \
  do {                                                                                                             \
    if (fast && has_res)                                                                                           \
      hipLaunchKernelGGL((sparse_conv_mfma_kernel<WM, WN, TM_, TN_, true, true>), grid, dim3(256), 0, st, p);      \
    else if (fast)                                                                                                 \
      hipLaunchKernelGGL((sparse_conv_mfma_kernel<WM, WN, TM_, TN_, true, false>), grid, dim3(256), 0, st, p);     \
    else if (has_res)                                                                                              \
      hipLaunchKernelGGL((sparse_conv_mfma_kernel<WM, WN, TM_, TN_, false, true>), grid, dim3(256), 0, st, p);     \
    else                                                                                                           \
      hipLaunchKernelGGL((sparse_conv_mfma_kernel<WM, WN, TM_, TN_, false, false>), grid, dim3(256), 0, st, p);    \
  } while (0)
  switch (shape) {
    case T128x128: CNRMA_CONV_LAUNCH(2, 2, 2, 2); break;
    case T128x64: CNRMA_CONV_LAUNCH(4, 1, 1, 2); break;
    case T64x64: CNRMA_CONV_LAUNCH(2, 2, 1, 1); break;
    case T128x32: CNRMA_CONV_LAUNCH(4, 1, 1, 1); break;
    default: return CNRMA_EINVAL;
  }
#undef CNRMA_CONV_LAUNCH
  if (p.splits > 1) {
    int64_t blocks = ceil_div(no_cap * Cout / 4 + 1, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================================
// "Gather-once" convolution (round 4).  What the per-(offset, 32-channel slice) stage structure above costs was measured
// with the diagnostic kernels (cnrma_debug_conv_tuning ablation masks, scripts/conv_sweep.py): on the 277 k-row 64 -> 64
// layer the MFMAs + fragment reads take 100 us of 356, the loads 110, the LDS stores 30-60, and the bare loop skeleton
// (two barriers, index bookkeeping per 384 MFMA cycles) 92 -- added up, not overlapped.  The remedy is not a faster
// stage but far fewer of them:
//   * the rows a 64-row output tile gathers over its 27 offsets overlap heavily when the rows are a compact block of
//     voxels (Morton order): ~200-270 DISTINCT input rows instead of 64 x ~20.  tile_union_kernel lists them once per
//     (coordinate set, kernel) -- cached with the neighbour table -- in groups of offsets whose union fits the LDS image
//     (one group for compact tiles; tiles without locality degrade gracefully to several groups);
//   * per 32-channel slice the union rows are fetched ONCE, split into the two fp16 planes and stored in LDS; then the
//     offsets of the group run back to back WITHOUT barriers, LDS stores or index loads: the A fragment of output row r at
//     offset k is an indexed LDS read (row lidx[r][k] of the image), the B fragment comes straight from memory in MFMA
//     operand order (weight image [K][Cin/32][Cout_p/32][plane][k-step][lane][8]: one coalesced 1-KB load per wave and
//     fragment, prefetched one offset ahead) and never touches LDS.
// Barriers per tile: 2 per (slice, group) instead of 2 per (offset, slice); A traffic from L2 / Infinity Cache: one row
// per union entry instead of one per (row, offset).  Sums run over (slice, group, offset) instead of (offset, slice): the
// result differs from the stage kernel's in fp32 rounding order only.
// ================================================================================================================
constexpr int GO_BM = 64;            // output rows per tile
constexpr int GO_UMAX = 280;         // union rows of one group held in LDS (+ 1 zero row): 2 planes x 64 B x 281 = 36 KB; with the
                                     // local indices 39.4 KB per block: FOUR blocks per CU (320 rows: 43.5 KB, three)
constexpr int GO_HASH = 2048;        // per-wave hash slots of the builder (>= 27 x 64 entries: a try never fills the set)
constexpr int GO_HDR = 84;           // ints per tile: [0] groups, then per group {offset mask, first entry, entries}
constexpr int GO_ROWS = 27 * GO_BM;  // worst case entries per tile (every (row, offset) distinct)
// two offsets of weights are in flight per wave (registers b0 / b1 of the kernel: 123-128 registers, four waves per SIMD; with four
// in flight 158-166 and three -- measured S 247 -> 256-258, NS 58.4 -> 59.0 scenes/s for the smaller footprint)

// one wave per tile: groups of offsets + sorted union lists + local indices.
//   * the tile's 64 x 27 slice of the neighbour table is contiguous: 27 independent coalesced loads, then registers;
//   * a group = a range of offsets whose distinct rows fit the LDS image.  A range is tried as a whole: every lane inserts its
//     entries of the range into a per-wave LDS hash set, the set is compacted with ballots, and if it holds <= GO_UMAX rows
//     the group is closed; otherwise the range is cut into ceil(1.15 * rows / GO_UMAX) parts that are tried in order
//     (compact tiles close [0, 27) at the first attempt; a single offset always fits: <= 64 rows);
//   * insertion without atomics and without a per-offset dependency chain: the set belongs to ONE wave, whose LDS
//     instructions execute in order, so a round is "read the slot of every pending entry; write where it was empty (the
//     last writer wins); read back": an entry is placed when the read-back shows its row, else it moves to the next slot.
//     Entries with the same row share home slot and probe path and move in lockstep, so a row never lands twice;
//   * closing a group: ranks by counting (each lane ranks its <= 5 entries against the list read 4 at a time: ascending
//     row order -- neighbours of consecutive Morton rows are then mostly consecutive image rows and the indexed fragment
//     reads of the convolution stay nearly conflict-free); the rank replaces the key in the set, and the local indices are
//     read from the slots remembered at insertion.
constexpr int GO_HASH_BITS = 11;
static_assert(GO_HASH == 1 << GO_HASH_BITS, "hash size");
__global__ __launch_bounds__(256) void tile_union_kernel(const int32_t* __restrict__ nbr, int64_t no_cap,
                                                         const int32_t* __restrict__ no_dev, int K,
                                                         int32_t* __restrict__ hdr, int32_t* __restrict__ rows,
                                                         uint16_t* __restrict__ lidx, int ablate) {
  __shared__ int32_t hkey[4][GO_HASH + 1];                       // + 1: the slot the branch-free "no write" goes to
  __shared__ __attribute__((aligned(16))) int32_t nloc[4][GO_BM * 27];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t n_live = live_rows(no_cap, no_dev);
  const int64_t tile0 = tile * GO_BM;
  if (tile0 >= n_live) return;
  const int rows_here = (int)min((int64_t)GO_BM, n_live - tile0);
  int32_t* hk = hkey[wave];
  int32_t* nl = nloc[wave];
  uint16_t* stage = reinterpret_cast<uint16_t*>(nl);
  int32_t* th = hdr + tile * GO_HDR;
  int32_t* tr = rows + tile * GO_ROWS;
  // the tile's table slice [rows_here][27] is flat in memory: 27 coalesced loads, then through LDS so that lane r holds
  // row r's 27 entries (stride 27 ints: conflict-free).  The buffer becomes the staging area of the local indices.
  int32_t v[27];
  {
    const int32_t* nb = nbr + tile0 * 27;
    const int total = rows_here * 27;
    int32_t f[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) f[q] = q * 64 + lane < total ? nb[q * 64 + lane] : -1;
#pragma unroll
    for (int q = 0; q < 27; ++q) nl[q * 64 + lane] = f[q];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 27; ++q) v[q] = nl[lane * 27 + q];
    __builtin_amdgcn_wave_barrier();
  }
  int li[27];                                                          // local indices of the lane's row (GO_UMAX = none)
#pragma unroll
  for (int q = 0; q < 27; ++q) li[q] = GO_UMAX;
  unsigned valid_k = 0;
#pragma unroll
  for (int q = 0; q < 27; ++q)
    if (__ballot(v[q] >= 0) != 0ull) valid_k |= 1u << q;

  int n_groups = 0, u_begin = 0;
  // the ranges still to do are consecutive: `cuts` has bit i set where a range ends after offset i
  unsigned cuts = 1u << 26;
  int lo = 0;
  while (lo < 27) {
    const int hi = __builtin_ctz(cuts >> lo) + lo + 1;
    const unsigned rmask = ((hi >= 32 ? 0u : (1u << hi)) - 1u) & ~((1u << lo) - 1u) & valid_k;
    if (rmask == 0u) { lo = hi; continue; }
    for (int i = lane; i < GO_HASH; i += 64) hk[i] = -1;
    __builtin_amdgcn_wave_barrier();
    int sl[27];
    unsigned pend = 0;
#pragma unroll
    for (int q = 0; q < 27; ++q) {
      sl[q] = (int)((unsigned)v[q] * 2654435761u >> (32 - GO_HASH_BITS));
      pend |= (((rmask >> q) & 1u) & (v[q] >= 0 ? 1u : 0u)) << q;
    }
    if (ablate & 32) pend = 0;                                         // diagnostics (cnrma_debug_conv_tuning): phases off
    const int max_rounds = hi - lo > GO_UMAX / GO_BM ? 8 : GO_HASH;    // <= GO_UMAX / 64 offsets always fit and always complete
    int rounds = 0;
    // branch-free rounds (the per-entry branches of a masked formulation cost more instructions than the work itself)
    while (__ballot(pend != 0u) != 0ull && rounds < max_rounds) {
      int32_t cur[27];
#pragma unroll
      for (int q = 0; q < 27; ++q) cur[q] = hk[sl[q]];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 27; ++q) hk[(((pend >> q) & 1u) != 0u && cur[q] == -1) ? sl[q] : GO_HASH] = v[q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 27; ++q) cur[q] = hk[sl[q]];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 27; ++q) {
        const unsigned pq = (pend >> q) & 1u, ok = cur[q] == v[q] ? 1u : 0u;
        pend &= ~((pq & ok) << q);
        sl[q] = (pq & (ok ^ 1u)) ? ((sl[q] + 1) & (GO_HASH - 1)) : sl[q];
      }
      ++rounds;
    }
    const bool crowded = __ballot(pend != 0u) != 0ull;                 // long probe chains: far more than GO_UMAX rows
    __builtin_amdgcn_wave_barrier();
    // image row of a distinct input row = its order of first appearance over (offset, lane): the entry with the smallest
    // q * 64 + lane among the holders of a slot owns it (an LDS minimum), owners are numbered with ballots
    const unsigned act = crowded || (ablate & 64) ? 0u : rmask;
    unsigned own = 0;
    int cnt = 0;
    int mypos[27];
    {
#pragma unroll
      for (int q = 0; q < 27; ++q) {
        const bool mine = ((act >> q) & 1u) != 0u && v[q] >= 0;
        hk[mine ? sl[q] : GO_HASH] = 0x7fffffff;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 27; ++q) {
        const bool mine = ((act >> q) & 1u) != 0u && v[q] >= 0;
        atomicMin(&hk[mine ? sl[q] : GO_HASH], q * 64 + lane);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 27; ++q) {
        const bool mine = ((act >> q) & 1u) != 0u && v[q] >= 0;
        const bool owner = mine && hk[sl[q]] == q * 64 + lane;
        const unsigned long long bal = __ballot(owner);
        mypos[q] = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        own |= (owner ? 1u : 0u) << q;
        cnt += __popcll(bal);
      }
    }
    if (crowded || cnt > GO_UMAX) {
      // cut the range; its first part runs next.  Up to 2 x GO_UMAX rows: two halves, each tried once more; beyond (a tile
      // without locality): pieces of GO_UMAX / 64 offsets, which fit whatever their rows are -- a tile costs <= 9 attempts
      const int len = hi - lo;
      constexpr int SURE = GO_UMAX / GO_BM;
      if (!crowded && cnt <= 2 * GO_UMAX - GO_UMAX / 4 && len > 2 * SURE) {
        cuts |= 1u << (lo + len / 2 - 1);
      } else {
        for (int c = lo + SURE; c < hi; c += SURE) cuts |= 1u << (c - 1);
      }
      continue;
    }
    const int un = cnt;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 27; ++q) {
      const bool owner = ((own >> q) & 1u) != 0u;
      hk[owner ? sl[q] : GO_HASH] = mypos[q];                          // the set now maps slot -> image row
      if (owner) tr[u_begin + mypos[q]] = v[q];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 27; ++q) {
      const int r = hk[sl[q]];
      li[q] = (((rmask >> q) & 1u) != 0u && v[q] >= 0) ? r : li[q];
    }
    if (lane == 0) { th[1 + 3 * n_groups] = (int)rmask; th[2 + 3 * n_groups] = u_begin; th[3 + 3 * n_groups] = un; }
    u_begin += un;
    ++n_groups;
    lo = hi;
    __builtin_amdgcn_wave_barrier();
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 27; ++q) stage[lane * 27 + q] = (uint16_t)li[q];
  __builtin_amdgcn_wave_barrier();
  const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
  uint32_t* t32 = reinterpret_cast<uint32_t*>(lidx + tile * (GO_BM * 27));      // tile * 3456 bytes: 4-byte aligned
  for (int i = lane; i < GO_BM * 27 / 2; i += 64) t32[i] = s32[i];
  if (lane == 0) th[0] = n_groups;
}

struct GoArgs {
  const int32_t* hdr; const int32_t* rows; const uint16_t* lidx;
  int slices_per_split;      // 32-channel slices per blockIdx.z (splits > 1: partial slabs, reduced by conv_reduce_kernel ...
  unsigned* counters;        // ... or, when given, by the last block of each tile: one zeroed word per (row tile, column tile)
};

// W fp32 [K][Cin][Cout] -> fp16 fragment-order image [2 planes interleaved below][...]: element order
//   [k][slice][column tile of 32][plane][k-step (2)][lane (64)][8]   with lane = 32 * (kk / 8 % 2) + column % 32,
//   kk = channel inside the slice = 16 * k-step + 8 * (lane / 32) + j  -- exactly the B operand registers of
//   v_mfma_f32_32x32x16_f16, so that a wave fetches one fragment with one contiguous 1-KB load.  Trailer: max|W|.
__global__ __launch_bounds__(256) void prep_weights_f16_frag_kernel(const float* __restrict__ w, uint16_t* __restrict__ wt, int K,
                                                                    int Cin, int Cout, const float* __restrict__ amax) {
  const int Cp = conv_cout_padded(Cout);
  const int64_t total = (int64_t)K * Cin * Cp;              // elements per plane
  const float am = read_amax(amax);
  const float sc = f16_scale_for(am);
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(wt + 2 * total) = am;
  const int ns = Cin / BK, nt = Cp / 32;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    // t enumerates [k][slice][tile][ks][lane][j] (one plane); the two planes of a (k, slice, tile) are adjacent
    const int j = (int)(t & 7), lane = (int)((t >> 3) & 63), ks = (int)((t >> 9) & 1);
    int64_t q = t >> 10;
    const int tile = (int)(q % nt); q /= nt;
    const int slice = (int)(q % ns);
    const int k = (int)(q / ns);
    const int cin = slice * BK + ks * 16 + (lane >> 5) * 8 + j, co = tile * 32 + (lane & 31);
    uint16_t hh, mm;
    split2_f16(co < Cout ? w[((int64_t)k * Cin + cin) * Cout + co] * sc : 0.0f, hh, mm);
    const int64_t base = (((int64_t)k * ns + slice) * nt + tile) * 2048 + ks * 512 + lane * 8 + j;    // 2 planes x 1024 per tile
    wt[base] = hh;
    wt[base + 1024] = mm;
  }
}

#ifdef CNRMA_EXPERIMENTS
#include "sparse_exp_go1.inc"
#endif

// ================================================================================================================
// Gather-once convolution, second form (round 5).  The first form's blocks were timed phase by phase with s_memtime stamps
// (scripts/go_stamps.py, profiles/r05_go_stamps_*.log; 277 k rows, 64 -> 64, 77.8 k cycles per block at four blocks per CU):
// header / local indices / first barrier 15 %, the two union gathers 11 % each, the two offset phases 24 + 20 %, the barrier
// between them 7 %, merge + epilogue 12 % -- the matrix pipe sees a block for 44 % of its life (SQ_VALU_MFMA_BUSY 0.52 with four
// blocks per CU), the rest is chains of dependent loads.  Two tiles per block (half the weight bytes per MFMA, two blocks per
// CU) and four weight offsets in flight were built and measured 10-40 % / 0-5 % SLOWER (profiles/r05_go_forms_S.log): the
// weight stream is not the bound, the chains are.  This form keeps the tile, the image and the summation order (bit-identical
// results: the test) and shortens the chains:
//   * everything a block needs first is requested before anything is waited for: the tile header as ONE 16-byte scalar load,
//     the local indices, the first 128 row numbers of group 0 (speculatively: group 0 starts at entry 0 of the tile's list),
//     the magnitude bounds and the epilogue's scale / shift;
//   * the row numbers of group 0 stay in LDS for the later channel slices: their gathers are one round trip, not two;
//   * the offset loop is scalar: the wave index is made uniform (readfirstlane), so the offset masks, the pop of the next offset
//     and the weight addresses live in SGPRs (hipcc had kept them in VGPRs -- a divergent loop with 64-bit vector multiplies per
//     step); weight fragments are loaded with an SGPR base + lane offset;
//   * the local indices of the next offset are read one step ahead of its MFMAs;
//   * the grid is one-dimensional and decoded so that the blocks sharing operands run on ONE XCD (workgroups are dealt round-
//     robin over the 8 XCDs, each with a private 4-MB L2): layers split over many (column tile, channel slice) groups -- the
//     small levels, whose weight tensors are 7-28 MB -- keep every group on one XCD, so a group's weights are pulled from the
//     fabric once instead of once per XCD (541 rows x 512 -> 512: 56.7 -> 38 us); layers with few groups give each XCD one
//     contiguous eighth of the row tiles (neighbouring tiles share most of their union rows).
// ================================================================================================================
struct Go2Map { int tiles, ncol, ng, mode, per; };   // mode 0: block L = tile L / ng, group L % ng; 1: groups -> XCDs; 2: tiles -> XCDs
// diagnostic build only (STAMP instantiation, conv_tuning(ablate=64)): phase boundaries of every block as s_memtime stamps of its first wave, 16 words
// per block in the buffer handed over as tile_counters (cdna_hip_programming.md "In-kernel stamps"); read the shares, not the time
__device__ __forceinline__ void go_stamp(unsigned long long* dbg, int slot) {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0 && slot < 16) dbg[(size_t)blockIdx.x * 16 + slot] = t;
}

template <int N>
__device__ __forceinline__ void go_waitn(u32x4_t (&b)[2][2]) {              // everything but the N newest loads has landed
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]) : "n"(N));
}
__device__ __forceinline__ void go_drain(u32x4_t (&b)[2][2]) {              // every load has landed (names what it releases)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
}
// one offset's four fragment loads (planes x k-steps, 1024 bytes apart) from an SGPR base + lane offset, as ONE statement.  The
// s_nop covers the "VALU writes an SGPR, vector memory reads it" hazard (5 wait states): hipcc's hazard recogniser does not look
// inside inline asm, and when it forms the base with v_readfirstlane right in front of the statement -- seen in the diagnostic
// instantiation -- the load went out with a stale half of the address (memory fault).  Early-clobber outputs: a landed fragment
// must not overwrite the lane offset the later loads of the statement still read.
__device__ __forceinline__ void go_load_frag(u32x4_t (&b)[2][2], const uint16_t* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\t"
               "global_load_dwordx4 %0, %4, %5 offset:0\n\t"
               "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
               "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
               "global_load_dwordx4 %3, %4, %5 offset:3072"
               : "=&v"(b[0][0]), "=&v"(b[0][1]), "=&v"(b[1][0]), "=&v"(b[1][1]) : "v"(voff), "s"(sbase));
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// the single-plane (bf16) variant: two loads per fragment (k-steps), the second plane's registers never named
template <int N>
__device__ __forceinline__ void go_waitn1(u32x4_t (&b)[2][2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0][0]), "+v"(b[0][1]) : "n"(N));
}
__device__ __forceinline__ void go_drain1(u32x4_t (&b)[2][2]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]));
}
__device__ __forceinline__ void go_load_frag1(u32x4_t (&b)[2][2], const uint16_t* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\t"
               "global_load_dwordx4 %0, %2, %3 offset:0\n\t"
               "global_load_dwordx4 %1, %2, %3 offset:1024"
               : "=&v"(b[0][0]), "=&v"(b[0][1]) : "v"(voff), "s"(sbase));
}

constexpr int GO2_US = (GO_UMAX + 2) * LDK;              // 16-bit elements of one plane of the image
constexpr int GO2_RS = 288;                             // row numbers parked per tile (>= GO_UMAX; two loads per thread)
constexpr int GO2_LDS = 2 * GO2_US * 2 + GO_BM * 27 * 2 + GO2_RS * 4;      // bytes: two planes, local indices, row numbers

// BF1: ONE bf16 plane per operand (round to nearest, no scaling), one MFMA per product on v_mfma_f32_32x32x16_bf16 -- the
// arithmetic of cnrma_sparse_conv_bf16 (autocast training) on the gather-once structure; weight image
// [k][slice][column tile][k-step][lane][8] (cnrma_sparse_conv_prepare_weights_bf16_frag)
// APF (round 6): the A fragments of a whole OFFSET are read one offset ahead of their MFMAs -- every fragment register is refilled
// for the next offset right behind its last use, the local indices run two offsets ahead -- so that no MFMA group waits for an
// LDS round trip (the form above exposes four per offset: 2 k-steps x 2 row tiles; on a grid that leaves <= 3 blocks per CU that
// latency is not covered by other waves: the 27-offset chain of a block is what the short levels cost).  ~16 more registers: three
// blocks per CU, which those grids do not reach anyway.  Same products in the same order per accumulator: bit-identical results.
template <int WAVES_N, int KS, bool HAS_RES, int NB, bool STAMP = false, bool BF1 = false, bool APF = false>
__global__ __launch_bounds__(256, APF ? 3 : ((NB == 2 && !STAMP) ? 4 : 2)) void sparse_conv_go2_kernel(ConvArgs p, GoArgs g, const uint16_t* __restrict__ wfrag,
                                                                             Go2Map mp) {
  static_assert(WAVES_N * KS == 4 && (KS == 1 || KS == 2), "four waves: column tiles x offset halves");
  static_assert(!APF || (NB == 2 && !BF1 && !STAMP), "fragment look-ahead: the f16x3 product form only");
  constexpr int TM = 2, BN = 32 * WAVES_N;
  extern __shared__ __attribute__((aligned(16))) unsigned char go2_smem[];
  __bf16* const Us = reinterpret_cast<__bf16*>(go2_smem);                                // [2 planes][(GO_UMAX + 2) * LDK]
  uint16_t* const Ls = reinterpret_cast<uint16_t*>(go2_smem + 2 * GO2_US * 2);            // [GO_BM * 27]
  int32_t* const Rs = reinterpret_cast<int32_t*>(go2_smem + 2 * GO2_US * 2 + GO_BM * 27 * 2);   // [GO_UMAX] row numbers of group 0
  // STAMP: the diagnostic instantiation (conv_tuning(ablate=64)): the product code + s_memtime stamps at its phase boundaries
  unsigned long long* const dbg = STAMP ? reinterpret_cast<unsigned long long*>(g.counters) : nullptr;
  int n_stamp = 0;
  if (STAMP && dbg) go_stamp(dbg, n_stamp++);                  // 0: block start
  if (STAMP && dbg && threadIdx.x == 0)                        // 14: where the block ran -- XCC_ID (hwreg 20) | HW_ID (hwreg 4) << 32
    dbg[(size_t)blockIdx.x * 16 + 14] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) |
                                        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
  const int L = blockIdx.x;
  int tile_i, grp;
  if (mp.mode == 1) { grp = (L & 7) + 8 * ((L >> 3) / mp.tiles); tile_i = (L >> 3) % mp.tiles; }
  else if (mp.mode == 2) { tile_i = (L & 7) * mp.per + (L >> 3) / mp.ng; grp = (L >> 3) % mp.ng; if ((L >> 3) / mp.ng >= mp.per) return; }
  else { tile_i = L / mp.ng; grp = L % mp.ng; }
  if (tile_i >= mp.tiles || grp >= mp.ng) return;
  const int64_t n_live = live_rows(p.no_cap, p.no_dev);
  const int64_t tile = tile_i, tile0 = tile * GO_BM;
  if (tile0 >= n_live) return;
  const int cout0 = (grp % mp.ncol) * BN, zs = grp / mp.ncol;
  const int Cin = p.Cin, Cout = p.Cout;
  const int Cout_p = conv_cout_padded(Cout), nt = Cout_p / 32, ns = Cin / BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);                  // uniform: everything derived from it stays scalar
  const int kg = wid / WAVES_N, wc = wid % WAVES_N;
  const int32_t* th = g.hdr + tile * GO_HDR;
  const int32_t* tr = g.rows + tile * GO_ROWS;
  // ---- requests first, waits later: header (one scalar 16-byte load: groups, then group 0's {mask, first entry, entries}),
  // local indices, the row numbers of the first gather batch (group 0 starts at entry 0; entries behind the union are read
  // but never used as addresses), bounds, epilogue operands
  const int4 h0 = *reinterpret_cast<const int4*>(th);
  uint4 lv = make_uint4(0x00010000u, 0x00030002u, 0x00050004u, 0x00070006u);
  if (tid < GO_BM * 27 / 8) lv = reinterpret_cast<const uint4*>(g.lidx + tile * (GO_BM * 27))[tid];
  const int32_t rn0 = tr[tid], rn1 = tr[256 + (tid & 31)];     // all row numbers a union of group 0 can have (GO2_RS >= GO_UMAX)
  const int col = cout0 + wc * 32 + (lane & 31);
  const bool col_ok = col < Cout;
  const int colc = col_ok ? col : 0;
  const bool use_scale = p.scale != nullptr && p.splits == 1, use_shift = p.shift != nullptr && p.splits == 1;
  const float sc = use_scale ? p.scale[colc] : 1.0f;
  const float sh = use_shift ? p.shift[colc] : 0.0f;
  const float a_scale = BF1 ? 1.0f : f16_scale_for(read_amax(p.in_amax));
  const float out_scale = BF1 ? 1.0f : 1.0f / (a_scale * f16_scale_for(*p.w_amax));
  if (tid < GO_BM * 27 / 8) reinterpret_cast<uint4*>(Ls)[tid] = lv;
  Rs[tid] = rn0;
  if (tid < GO2_RS - 256) Rs[256 + tid] = rn1;
  if (tid < 16) {                                            // the zero row of both planes (64 bytes each)
    reinterpret_cast<uint32_t*>(Us + GO_UMAX * LDK)[tid] = 0u;
    reinterpret_cast<uint32_t*>(Us + GO2_US + GO_UMAX * LDK)[tid] = 0u;
  }
  const int n_groups = __builtin_amdgcn_readfirstlane(h0.x);
  int s_lo = 0, s_hi = ns;
  if (p.splits > 1) { s_lo = zs * g.slices_per_split; s_hi = min(ns, s_lo + g.slices_per_split); }

  f32x16 acc[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;

  const int fhalf = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;
  const uint16_t* ls0 = Ls + (lane & 31) * 27;               // this lane's rows of the local indices: row, row + 32
  auto load_b = [&](u32x4_t (&bf)[2][2], int k, int slice) {
    const uint64_t ba = reinterpret_cast<uint64_t>(wfrag + (((int64_t)k * ns + slice) * nt + (cout0 >> 5) + wc) * (BF1 ? 1024 : 2048));
    const uint16_t* base = reinterpret_cast<const uint16_t*>(                          // uniform by construction: say so
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ba >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ba));
    if constexpr (BF1) go_load_frag1(bf, base, lane16);        // [k-step]
    else go_load_frag(bf, base, lane16);                       // [plane][k-step]: 1024 bytes apart
  };
  auto load_li = [&](int (&li)[TM], int k) {
#pragma unroll
    for (int a = 0; a < TM; ++a) li[a] = ls0[a * (32 * 27) + k];
  };
  // one offset: A fragments one ahead of their MFMAs (two registers in rotation), as in the first form
  auto mfma_k = [&](const u32x4_t (&bf)[2][2], const int (&li)[TM]) {
    if constexpr (BF1) {                                     // one plane, one MFMA per (k-step, row tile); fragments one ahead
      auto rdb = [&](int a, int ks) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(Us + lds_slot(li[a], ks * 2 + fhalf));
      };
      bf16x8_t curb = rdb(0, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, bf[0][ks]);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          bf16x8_t nxtb = curb;
          if (!(ks == 1 && a == TM - 1)) nxtb = a + 1 < TM ? rdb(a + 1, ks) : rdb(0, ks + 1);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(curb, bb, acc[a], 0, 0, 0);
          curb = nxtb;
        }
      }
      return;
    }
    auto rd = [&](int a, int pl, int ks) -> f16x8_t {
      return *reinterpret_cast<const f16x8_t*>(Us + pl * GO2_US + lds_slot(li[a], ks * 2 + fhalf));
    };
    f16x8_t cur = rd(0, 1, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f16x8_t bh = __builtin_bit_cast(f16x8_t, bf[0][ks]), bm = __builtin_bit_cast(f16x8_t, bf[1][ks]);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        f16x8_t nxt = rd(a, 0, ks);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, bh, acc[a], 0, 0, 0);       // m*h
        cur = nxt;
        const bool last = ks == 1 && a == TM - 1;
        if (!last) nxt = a + 1 < TM ? rd(a + 1, 1, ks) : rd(0, 1, ks + 1);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, bm, acc[a], 0, 0, 0);       // h*m
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur, bh, acc[a], 0, 0, 0);       // h*h
        cur = nxt;
      }
    }
  };

  for (int slice = s_lo; slice < s_hi; ++slice) {
    const int cin0 = slice * BK;
    for (int grpi = 0; grpi < n_groups; ++grpi) {
      // (readfirstlane: the header is the same for every lane, but a load hipcc cannot prove unclobbered is a vector load)
      const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane(grpi == 0 ? h0.y : th[1 + 3 * grpi]);
      const int ub = __builtin_amdgcn_readfirstlane(grpi == 0 ? h0.z : th[2 + 3 * grpi]);
      const int un = __builtin_amdgcn_readfirstlane(grpi == 0 ? h0.w : th[3 + 3 * grpi]);
      unsigned mymask = mask;
      if constexpr (KS == 2) {                               // every second offset of the group
        mymask = 0;
        unsigned m = mask;
        int r = 0;
        while (m) {
          const unsigned low = m & (0u - m);
          if ((r & 1) == kg) mymask |= low;
          m ^= low;
          ++r;
        }
      }
      u32x4_t bf[NB][2][2];
      int kk[NB];
      unsigned rest = mymask;
      const int n_off = __popc(mymask);
      int k_last = 0;
      auto pop = [&]() { if (rest) { k_last = __ffs(rest) - 1; rest &= rest - 1u; } return k_last; };
#pragma unroll
      for (int j = 0; j < NB; ++j) { kk[j] = pop(); load_b(bf[j], kk[j], slice); }
      __syncthreads();                                       // the previous stage's fragment reads are done
      if (STAMP && dbg) go_stamp(dbg, n_stamp++);              // 1 + 3 i: metadata / previous phase done, weights requested
      // ---- the union rows of the group, once: 8 lanes per row (4 channels each), 4 rows per thread in flight.  Row numbers:
      // group 0 in its first slice from memory (the first batch was requested at the top of the kernel) and kept in LDS for
      // the later slices; other groups (tiles without locality) from memory every time
      // Row numbers: group 0 from LDS (all of them were requested at the top of the kernel: every gather batch of every slice
      // is ONE round trip), other groups (tiles without locality) from memory
      const int tasks = un * 8;
      const int32_t* rows_g = tr + ub;
      for (int t0 = 0; t0 < tasks; t0 += 256 * 4) {
        float4 v[4];
        int32_t src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tk = t0 + i * 256 + tid;
          const int r = tk < tasks ? (tk >> 3) : 0;
          src[i] = grpi == 0 ? Rs[r] : rows_g[r];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(p.in + (int64_t)src[i] * Cin + cin0 + (tid & 7) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // every loaded value is consumed on every path (tasks behind the union go to the dump row), see the first form
          const int tk = t0 + i * 256 + tid;
          const int u = tk < tasks ? (tk >> 3) : GO_UMAX + 1, kc = tid & 7;
          __bf16* d = Us + lds_slot(u, kc >> 1) + (kc & 1) * 4;
          if constexpr (BF1) {
            *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16x2(v[i].x, v[i].y), pack_bf16x2(v[i].z, v[i].w));
          } else {
            uint2 h, m;
            split2(v[i], a_scale, h, m);
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + GO2_US) = m;
          }
        }
      }
      __syncthreads();
      if (STAMP && dbg) go_stamp(dbg, n_stamp++);              // 2 + 3 i: union image staged
      // ---- the group's offsets back to back: no barrier, no LDS store, no index load from memory; the local indices of an
      // offset are read while the offset before it runs
      if constexpr (APF) {
        // fragments one offset ahead.  fr[a][plane][k-step] holds the CURRENT offset's operands; each is reloaded for the next
        // offset (rows li_n) right behind its last MFMA.  Local indices: two offsets ahead (an offset's row numbers must be in
        // registers when the offset before it starts).
        f16x8_t fr[TM][2][2];
        auto rdf = [&](const int (&li)[TM], int a, int pl, int ks) -> f16x8_t {
          return *reinterpret_cast<const f16x8_t*>(Us + pl * GO2_US + lds_slot(li[a], ks * 2 + fhalf));
        };
        auto mfma_ahead = [&](auto sync_tag, const u32x4_t (&bfr)[2][2], const int (&li_next)[TM], bool more) {
          constexpr int SY = decltype(sync_tag)::value;      // one scheduling pipeline per call site
          static_assert(TM == 2, "two row tiles per wave");
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const f16x8_t bh = __builtin_bit_cast(f16x8_t, bfr[0][ks]), bm = __builtin_bit_cast(f16x8_t, bfr[1][ks]);
            // the two accumulators alternate (a dependent MFMA issues every 64 cycles, two chains fill the pipe); per accumulator the
            // order stays m*h, h*m, h*h as in the form above
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][1][ks], bh, acc[0], 0, 0, 0);
            if (more) fr[0][1][ks] = rdf(li_next, 0, 1, ks);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][1][ks], bh, acc[1], 0, 0, 0);
            if (more) fr[1][1][ks] = rdf(li_next, 1, 1, ks);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][0][ks], bm, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][0][ks], bm, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][0][ks], bh, acc[0], 0, 0, 0);
            if (more) fr[0][0][ks] = rdf(li_next, 0, 0, ks);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][0][ks], bh, acc[1], 0, 0, 0);
            if (more) fr[1][0][ks] = rdf(li_next, 1, 0, ks);
          }
          // pin that issue order (mh0 R mh1 R hm0 hm1 hh0 R hh1 R per k-step): every fragment is re-read right behind its last use --
          // left alone the scheduler parks all eight reads behind the 8th-11th MFMA, and the next offset's first MFMA waits for them
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SY); __builtin_amdgcn_sched_group_barrier(0x100, 1, SY);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SY); __builtin_amdgcn_sched_group_barrier(0x100, 1, SY);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, SY); __builtin_amdgcn_sched_group_barrier(0x100, 1, SY);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, SY); __builtin_amdgcn_sched_group_barrier(0x100, 1, SY);
          }
        };
        int li_n[TM];
        {
          int li_c[TM];
          load_li(li_c, kk[0]);
          load_li(li_n, kk[1 % NB]);                           // offset 1 (one offset only: the same again, never used)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < TM; ++a) { fr[a][1][ks] = rdf(li_c, a, 1, ks); fr[a][0][ks] = rdf(li_c, a, 0, ks); }
        }
        int i = 0;
        auto step = [&](auto jt) {                             // one steady-state offset; jt: which of the two weight sets
          constexpr int j = decltype(jt)::value;
          int li[TM];
#pragma unroll
          for (int a = 0; a < TM; ++a) li[a] = li_n[a];          // rows of the NEXT offset
          const int k2 = pop();                                // the offset after it (behind the end: the last one again)
          load_li(li_n, k2);
          go_waitn<(NB - 1) * 4>(bf[j]);
          mfma_ahead(jt, bf[j], li, true);
          kk[j] = k2;
          load_b(bf[j], k2, slice);
        };
        for (; i + NB < n_off; i += NB) {
          step(std::integral_constant<int, 0>{});
          step(std::integral_constant<int, 1>{});
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) go_drain(bf[j]);
        {
          int li[TM];
#pragma unroll
          for (int a = 0; a < TM; ++a) li[a] = li_n[a];
          if (i < n_off) mfma_ahead(std::integral_constant<int, 2>{}, bf[0], li, i + 1 < n_off);
          if (i + 1 < n_off) mfma_ahead(std::integral_constant<int, 3>{}, bf[1], li, false);
        }
      } else {
        int lin[TM];
        load_li(lin, kk[0]);
        int i = 0;
        for (; i + NB < n_off; i += NB) {                        // steady state: NB offsets per turn, their successors prefetched
  #pragma unroll
          for (int j = 0; j < NB; ++j) {
            int li[TM];
  #pragma unroll
            for (int a = 0; a < TM; ++a) li[a] = lin[a];
            load_li(lin, kk[(j + 1) % NB]);                      // j + 1 < NB: this turn's next offset; else the next turn's first
            if constexpr (BF1) go_waitn1<(NB - 1) * 2>(bf[j]);
            else go_waitn<(NB - 1) * 4>(bf[j]);                  // bf[j] has landed; the NB - 1 sets behind it may still fly
            mfma_k(bf[j], li);
            kk[j] = pop();                                       // behind the last offset: the last one again (never used)
            load_b(bf[j], kk[j], slice);
          }
        }
        // the last <= NB offsets: nothing more to prefetch, and NO load may be left in flight (the asm loads are invisible to the
        // compiler, which is free to reuse their destination registers from here on)
  #pragma unroll
        for (int j = 0; j < NB; ++j) {
          if constexpr (BF1) go_drain1(bf[j]);
          else go_drain(bf[j]);
        }
  #pragma unroll
        for (int j = 0; j < NB; ++j) {
          int li[TM];
  #pragma unroll
          for (int a = 0; a < TM; ++a) li[a] = lin[a];
          if (j + 1 < NB) load_li(lin, kk[j + 1]);
          if (i + j < n_off) mfma_k(bf[j], li);
        }
      }
      if (STAMP && dbg) go_stamp(dbg, n_stamp++);              // 3 + 3 i: this wave's offsets issued
    }
  }

  if constexpr (KS == 2) {
    // the two offset halves meet: each wave hands the row tile it does not finish to its partner through LDS (the image is
    // dead) and finishes the other -- wave (kg, wc) writes rows 32 * kg .. + 31 of columns 32 * wc .. + 31
    __syncthreads();
    float* X = reinterpret_cast<float*>(go2_smem);           // 4 x 4 KB
    float* mine = X + (wc * 2 + kg) * 1024;
    const float* theirs = X + (wc * 2 + (kg ^ 1)) * 1024;
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = acc[1][i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = acc[0][i];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] += theirs[i * 64 + lane];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[1][i] += theirs[i * 64 + lane];
    }
  }

  float mx = 0.0f;
  if (p.splits > 1) {                                        // partial tile into this split's slab; conv_reduce_kernel follows
    float* slab = p.slab + (int64_t)zs * p.no_cap * Cout;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (KS == 2 && a != kg) continue;
      const int64_t row0 = tile0 + a * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = row0 + (i & 3) + 8 * (i >> 2);
        if (col_ok && row < n_live) slab[row * Cout + col] = acc[a][i] * out_scale;
      }
    }
    if (STAMP && dbg) go_stamp(dbg, 15);
    return;
  }
  const int act = p.act;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    if (KS == 2 && a != kg) continue;
    const int64_t row0 = tile0 + a * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float res[4];
      if (HAS_RES) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t row = row0 + q + 8 * rg;
          const int64_t rc = row < n_live ? row : n_live - 1;
          res[q] = p.residual[rc * Cout + colc];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t row = row0 + q + 8 * rg;
        float v = acc[a][rg * 4 + q];
        v = v * out_scale;
        v = v * sc;
        v = v + sh;
        if (HAS_RES) v = v + res[q];
        v = apply_act(v, act);
        if (col_ok && row < n_live) {
          p.out[row * Cout + col] = v;
          mx = fmaxf(mx, fabsf(v));
        }
      }
    }
  }
  if (STAMP && dbg) go_stamp(dbg, 15);                         // 15: epilogue stores issued
  if (p.out_amax != nullptr) {
    __syncthreads();
    block_amax_publish(p.out_amax, mx, reinterpret_cast<float*>(go2_smem));
  }
}

// ================================================================================================================
// Gather-once convolution in exact fp32 (round 5): the structure of sparse_conv_go2_kernel on v_mfma_f32_32x32x2_f32, for
// CONV_PRECISION = "f32" -- the reference's own arithmetic (fcaf3d_backbone.py:26-31 runs ME's fp32 GEMMs), whose 3x3x3 stride-1
// layers had stayed on the round-1 stage kernel (a barrier pair per (offset, 32-channel slice), 0.36 of the fp32 MFMA peak).
//   * image: the tile's union rows as raw fp32, 128 bytes per row (32 channels), eight 16-byte chunks XORed with bits 1..3
//     of the row: a lane reads its operands as ds_read_b128 (4 floats = 4 MFMAs' worth), 16 rows of a lane group on 16 distinct
//     bank positions; no split, no scale;
//   * one MFMA multiplies 32 rows x 2 channels; lanes 0-31 hold one channel of the pair, lanes 32-63 the other.  A lane's four
//     floats are channels 8 i + 4 h + e (i = read 0..3, h = lane / 32, e = 0..3): MFMA (i, e) pairs channel 8 i + e with channel
//     8 i + 4 + e.  The weight image is in the matching fragment order [k][slice][column tile][i][lane][e] -- one 16-byte load
//     per lane and read, straight from memory into the B operand registers, prefetched two offsets ahead like the f16x3 form;
//   * sums run over (slice, group, offset, i, e): an fp32 fma chain per output in that fixed order -- the stage kernel's order is
//     (offset, slice, channel pair), so the two differ in rounding order only (test: 2e-6 vs the fp64 oracle, like the stage kernel).
// ================================================================================================================
constexpr int GOF_ROW = 32;                                 // floats per image row
constexpr int GOF_IMG = (GO_UMAX + 2) * GOF_ROW;            // floats: union rows + zero row + dump row
constexpr int GOF_LDS = GOF_IMG * 4 + GO_BM * 27 * 2 + GO_UMAX * 4;
static_assert(GOF_LDS <= 48 * 1024 && 4 * GOF_LDS <= 160 * 1024, "four blocks per CU");
__device__ __forceinline__ int gof_slot(int row, int chunk) { return row * GOF_ROW + ((chunk ^ ((row >> 1) & 7)) << 2); }

// W fp32 [K][Cin][Cout] -> fragment-order image, element order [k][slice][column tile of 32][i (4)][lane (64)][e (4)]:
// channel slice * 32 + 8 i + 4 (lane / 32) + e, column tile * 32 + lane % 32; columns padded to Cout_p with zeros
__global__ __launch_bounds__(256) void prep_weights_f32_frag_kernel(const float* __restrict__ w, float* __restrict__ wt, int K,
                                                                    int Cin, int Cout) {
  const int Cp = conv_cout_padded(Cout);
  const int64_t total = (int64_t)K * Cin * Cp;
  const int ns = Cin / BK, nt = Cp / 32;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(t & 3), lane = (int)((t >> 2) & 63), i = (int)((t >> 8) & 3);
    int64_t q = t >> 10;
    const int tile = (int)(q % nt); q /= nt;
    const int slice = (int)(q % ns);
    const int k = (int)(q / ns);
    const int cin = slice * BK + 8 * i + 4 * (lane >> 5) + e, co = tile * 32 + (lane & 31);
    wt[t] = co < Cout ? w[((int64_t)k * Cin + cin) * Cout + co] : 0.0f;
  }
}


template <int WAVES_N, int KS, bool HAS_RES>
__global__ __launch_bounds__(256, 4) void sparse_conv_gof_kernel(ConvArgs p, GoArgs g, const float* __restrict__ wfrag, Go2Map mp) {
  static_assert(WAVES_N * KS == 4 && (KS == 1 || KS == 2), "four waves: column tiles x offset halves");
  constexpr int TM = 2, BN = 32 * WAVES_N, NB = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char gof_smem[];
  float* const Us = reinterpret_cast<float*>(gof_smem);                                      // [(GO_UMAX + 2)][32] swizzled
  uint16_t* const Ls = reinterpret_cast<uint16_t*>(gof_smem + GOF_IMG * 4);                   // [GO_BM * 27]
  int32_t* const Rs = reinterpret_cast<int32_t*>(gof_smem + GOF_IMG * 4 + GO_BM * 27 * 2);     // [GO_UMAX] row numbers of group 0
  const int L = blockIdx.x;
  int tile_i, grp;
  if (mp.mode == 1) { grp = (L & 7) + 8 * ((L >> 3) / mp.tiles); tile_i = (L >> 3) % mp.tiles; }
  else if (mp.mode == 2) { tile_i = (L & 7) * mp.per + (L >> 3) / mp.ng; grp = (L >> 3) % mp.ng; if ((L >> 3) / mp.ng >= mp.per) return; }
  else { tile_i = L / mp.ng; grp = L % mp.ng; }
  if (tile_i >= mp.tiles || grp >= mp.ng) return;
  const int64_t n_live = live_rows(p.no_cap, p.no_dev);
  const int64_t tile = tile_i, tile0 = tile * GO_BM;
  if (tile0 >= n_live) return;
  const int cout0 = (grp % mp.ncol) * BN, zs = grp / mp.ncol;
  const int Cin = p.Cin, Cout = p.Cout;
  const int Cout_p = conv_cout_padded(Cout), nt = Cout_p / 32, ns = Cin / BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wid / WAVES_N, wc = wid % WAVES_N;
  const int32_t* th = g.hdr + tile * GO_HDR;
  const int32_t* tr = g.rows + tile * GO_ROWS;
  // requests first, waits later (see sparse_conv_go2_kernel)
  const int4 h0 = *reinterpret_cast<const int4*>(th);
  uint4 lv = make_uint4(0u, 0u, 0u, 0u);
  if (tid < GO_BM * 27 / 8) lv = reinterpret_cast<const uint4*>(g.lidx + tile * (GO_BM * 27))[tid];
  int32_t pre[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pre[i] = tr[(tid >> 3) + 32 * i];
  const int col = cout0 + wc * 32 + (lane & 31);
  const bool col_ok = col < Cout;
  const int colc = col_ok ? col : 0;
  const bool use_scale = p.scale != nullptr && p.splits == 1, use_shift = p.shift != nullptr && p.splits == 1;
  const float sc = use_scale ? p.scale[colc] : 1.0f;
  const float sh = use_shift ? p.shift[colc] : 0.0f;
  if (tid < GO_BM * 27 / 8) reinterpret_cast<uint4*>(Ls)[tid] = lv;
  if (tid < 32) Us[GO_UMAX * GOF_ROW + tid] = 0.0f;          // the zero row
  const int n_groups = __builtin_amdgcn_readfirstlane(h0.x);
  int s_lo = 0, s_hi = ns;
  if (p.splits > 1) { s_lo = zs * g.slices_per_split; s_hi = min(ns, s_lo + g.slices_per_split); }

  f32x16 acc[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;

  const int fhalf = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;
  const uint16_t* ls0 = Ls + (lane & 31) * 27;
  auto load_b = [&](u32x4_t (&bf)[2][2], int k, int slice) {
    const uint64_t ba = reinterpret_cast<uint64_t>(wfrag + (((int64_t)k * ns + slice) * nt + (cout0 >> 5) + wc) * 1024);
    const uint16_t* base = reinterpret_cast<const uint16_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ba >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ba));
    go_load_frag(bf, base, lane16);                            // reads i = 0..3, 1024 bytes apart: bf[i >> 1][i & 1]
  };
  auto load_li = [&](int (&li)[TM], int k) {
#pragma unroll
    for (int a = 0; a < TM; ++a) li[a] = ls0[a * (32 * 27) + k];
  };
  // one offset: 4 reads x 2 row tiles, each read one ahead of its 4 MFMAs (two registers in rotation)
  auto mfma_k = [&](const u32x4_t (&bf)[2][2], const int (&li)[TM]) {
    auto rd = [&](int a, int i) -> f32x4_t {
      return *reinterpret_cast<const f32x4_t*>(Us + gof_slot(li[a], 2 * i + fhalf));
    };
    f32x4_t cur = rd(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_t b4 = __builtin_bit_cast(f32x4_t, bf[i >> 1][i & 1]);
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const bool last = i == 3 && a == TM - 1;
        f32x4_t nxt = cur;
        if (!last) nxt = a + 1 < TM ? rd(a + 1, i) : rd(0, i + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[e], b4[e], acc[a], 0, 0, 0);
        cur = nxt;
      }
    }
  };

  for (int slice = s_lo; slice < s_hi; ++slice) {
    const int cin0 = slice * BK;
    for (int grpi = 0; grpi < n_groups; ++grpi) {
      const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane(grpi == 0 ? h0.y : th[1 + 3 * grpi]);
      const int ub = __builtin_amdgcn_readfirstlane(grpi == 0 ? h0.z : th[2 + 3 * grpi]);
      const int un = __builtin_amdgcn_readfirstlane(grpi == 0 ? h0.w : th[3 + 3 * grpi]);
      unsigned mymask = mask;
      if constexpr (KS == 2) {                               // every second offset of the group
        mymask = 0;
        unsigned m = mask;
        int r = 0;
        while (m) {
          const unsigned low = m & (0u - m);
          if ((r & 1) == kg) mymask |= low;
          m ^= low;
          ++r;
        }
      }
      u32x4_t bf[NB][2][2];
      int kk[NB];
      unsigned rest = mymask;
      const int n_off = __popc(mymask);
      int k_last = 0;
      auto pop = [&]() { if (rest) { k_last = __ffs(rest) - 1; rest &= rest - 1u; } return k_last; };
#pragma unroll
      for (int j = 0; j < NB; ++j) { kk[j] = pop(); load_b(bf[j], kk[j], slice); }
      __syncthreads();                                       // the previous stage's fragment reads are done
      const bool cached = grpi == 0 && slice != s_lo, keep = grpi == 0 && slice == s_lo && s_hi - s_lo > 1;
      const int tasks = un * 8;
      for (int t0 = 0; t0 < tasks; t0 += 256 * 4) {
        float4 v[4];
        int32_t src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tk = t0 + i * 256 + tid;
          const int r = tk < tasks ? (tk >> 3) : 0;
          if (cached) src[i] = Rs[r];
          else if (grpi == 0 && t0 == 0) src[i] = tk < tasks ? pre[i] : tr[0];
          else src[i] = tr[ub + r];
          if (keep && (tk & 7) == 0 && tk < tasks) Rs[r] = src[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tk = t0 + i * 256 + tid;
          v[i] = *reinterpret_cast<const float4*>(p.in + (int64_t)src[i] * Cin + cin0 + (tk & 7) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tk = t0 + i * 256 + tid;
          const int u = tk < tasks ? (tk >> 3) : GO_UMAX + 1, kc = tk & 7;
          *reinterpret_cast<float4*>(Us + gof_slot(u, kc)) = v[i];
        }
      }
      __syncthreads();
      int lin[TM];
      load_li(lin, kk[0]);
      int i = 0;
      for (; i + NB < n_off; i += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          int li[TM];
#pragma unroll
          for (int a = 0; a < TM; ++a) li[a] = lin[a];
          load_li(lin, kk[(j + 1) % NB]);
          go_waitn<(NB - 1) * 4>(bf[j]);
          mfma_k(bf[j], li);
          kk[j] = pop();
          load_b(bf[j], kk[j], slice);
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) go_drain(bf[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int li[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) li[a] = lin[a];
        if (j + 1 < NB) load_li(lin, kk[j + 1]);
        if (i + j < n_off) mfma_k(bf[j], li);
      }
    }
  }

  if constexpr (KS == 2) {
    __syncthreads();
    float* X = reinterpret_cast<float*>(gof_smem);           // 4 x 4 KB
    float* mine = X + (wc * 2 + kg) * 1024;
    const float* theirs = X + (wc * 2 + (kg ^ 1)) * 1024;
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = acc[1][i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = acc[0][i];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] += theirs[i * 64 + lane];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[1][i] += theirs[i * 64 + lane];
    }
  }

  if (p.splits > 1) {                                        // partial tile into this split's slab; conv_reduce_kernel follows
    float* slab = p.slab + (int64_t)zs * p.no_cap * Cout;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (KS == 2 && a != kg) continue;
      const int64_t row0 = tile0 + a * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = row0 + (i & 3) + 8 * (i >> 2);
        if (col_ok && row < n_live) slab[row * Cout + col] = acc[a][i];
      }
    }
    return;
  }
  const int act = p.act;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    if (KS == 2 && a != kg) continue;
    const int64_t row0 = tile0 + a * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float res[4];
      if (HAS_RES) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t row = row0 + q + 8 * rg;
          const int64_t rc = row < n_live ? row : n_live - 1;
          res[q] = p.residual[rc * Cout + colc];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t row = row0 + q + 8 * rg;
        float v = acc[a][rg * 4 + q];
        v = v * sc;
        v = v + sh;
        if (HAS_RES) v = v + res[q];
        v = apply_act(v, act);
        if (col_ok && row < n_live) p.out[row * Cout + col] = v;
      }
    }
  }
}

#ifdef CNRMA_EXPERIMENTS
#include "sparse_exp_go3.inc"
#endif

// children coordinates of the generative transposed conv: child k (x fastest) of parent i is row 8 * i + morton_child(k)
__global__ __launch_bounds__(256) void convtr_coords_kernel(const int32_t* __restrict__ in_coords, int64_t n_cap,
                                                            const int32_t* __restrict__ n_dev, int half,
                                                            int32_t* __restrict__ out_coords) {
  const int64_t n = live_rows(n_cap, n_dev);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  const int m = (int)(t & 7);                  // Morton rank of the child: x is the most significant bit
  const int64_t i = t >> 3;
  int4 c = reinterpret_cast<const int4*>(in_coords)[i];
  c.y += ((m >> 2) & 1) * half;
  c.z += ((m >> 1) & 1) * half;
  c.w += (m & 1) * half;
  reinterpret_cast<int4*>(out_coords)[t] = c;
}

// ================================================================================================================
// pooling / normalisation / elementwise
// ================================================================================================================
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, int C, const int32_t* __restrict__ nbr,
                                                      int K, float* __restrict__ out, int64_t no_cap,
                                                      const int32_t* __restrict__ no_dev) {
  const int64_t n = live_rows(no_cap, no_dev);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((C & 3) == 0) {                                   // 4 channels per lane: C/4 lanes share one row's neighbour list
    const int c4 = C >> 2;
    if (t >= n * c4) return;
    const int64_t o = t / c4;
    const int c = (int)(t - o * c4);
    float4 m = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
    bool any = false;
    for (int k = 0; k < K; ++k) {
      const int32_t s = nbr[o * K + k];
      if (s >= 0) {
        const float4 v = reinterpret_cast<const float4*>(in)[(int64_t)s * c4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        any = true;
      }
    }
    reinterpret_cast<float4*>(out)[t] = any ? m : make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (t >= n * C) return;
  const int64_t o = t / C;
  const int c = (int)(t - o * C);
  float m = -__builtin_inff();
  bool any = false;
  for (int k = 0; k < K; ++k) {
    int32_t s = nbr[o * K + k];
    if (s >= 0) { m = fmaxf(m, in[(int64_t)s * C + c]); any = true; }
  }
  out[t] = any ? m : 0.0f;
}

// per-channel sum / sum of squares in fp64, deterministic two-stage (partials [nblk][2C])
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float* __restrict__ in, int64_t n_cap,
                                                               const int32_t* __restrict__ n_dev,
                                                               const int32_t* __restrict__ row0_dev, int C,
                                                               double* __restrict__ part) {
  const int64_t n = live_rows(n_cap, n_dev);
  if (row0_dev != nullptr) in += (int64_t)__builtin_nontemporal_load(row0_dev) * C;     // a scene's row segment
  // thread -> channel c = tid % C (C <= 256), row group = tid / C
  const int c = threadIdx.x % C;
  const int groups = blockDim.x / C;
  const int gi = threadIdx.x / C;
  double s = 0.0, q = 0.0;
  if (gi < groups) {
    for (int64_t r = (int64_t)blockIdx.x * groups + gi; r < n; r += (int64_t)gridDim.x * groups) {
      double v = (double)in[r * C + c];
      s += v;
      q += v * v;
    }
  }
  __shared__ double sm[2 * 256];
  sm[threadIdx.x] = s;
  sm[256 + threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < C) {
    double ts = 0.0, tq = 0.0;
    for (int g = 0; g < groups; ++g) { ts += sm[g * C + threadIdx.x]; tq += sm[256 + g * C + threadIdx.x]; }
    part[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = ts;
    part[(int64_t)blockIdx.x * 2 * C + C + threadIdx.x] = tq;
  }
}

// one block per channel: 256 lanes add the per-block partials in a fixed order (deterministic)
__global__ __launch_bounds__(256) void colstats_final_kernel(const double* __restrict__ part, int nblk, int C, int64_t n_cap,
                                                             const int32_t* __restrict__ n_dev, double* __restrict__ stats) {
  const int c = blockIdx.x;
  __shared__ double ss[256], qq[256];
  double s = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) { s += part[(int64_t)b * 2 * C + c]; q += part[(int64_t)b * 2 * C + C + c]; }
  ss[threadIdx.x] = s; qq[threadIdx.x] = q;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) { ss[threadIdx.x] += ss[threadIdx.x + d]; qq[threadIdx.x] += qq[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double n = (double)live_rows(n_cap, n_dev);
    const double mean = ss[0] / n;
    double var = qq[0] / n - mean * mean;   // biased variance
    if (var < 0.0) var = 0.0;
    stats[c] = mean;
    stats[C + c] = var;
  }
}

__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ in, int64_t n_cap,
                                                             const int32_t* __restrict__ n_dev,
                                                             const int32_t* __restrict__ row0_dev, int C,
                                                             const double* __restrict__ stats,
                                                             const float* __restrict__ weight,
                                                             const float* __restrict__ bias, float eps, int relu,
                                                             float* __restrict__ out) {
  const int64_t n = live_rows(n_cap, n_dev);
  if (row0_dev != nullptr) {
    const int64_t r0 = (int64_t)__builtin_nontemporal_load(row0_dev) * C;
    in += r0;
    out += r0;
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const float mean = (float)stats[c];
    const float inv = 1.0f / sqrtf((float)stats[C + c] + eps);
    float v = (in[t] - mean) * inv;
    if (weight) v = v * weight[c];
    if (bias) v = v + bias[c];
    out[t] = relu ? fmaxf(v, 0.0f) : v;
  }
}

// InstanceNorm (+ ReLU) applied on the fly to the candidates of a max pooling (the stem: conv - InstanceNorm - ReLU - MaxPool,
// fcaf3d_backbone.py:25-32): the normalised tensor is never written or re-read, the pooled rows come with their magnitude
// bound.  The same operations in the same order as instnorm_apply_kernel followed by maxpool_kernel: bit-identical.
__global__ __launch_bounds__(256) void instnorm_maxpool_kernel(const float* __restrict__ in, int C,
                                                               const double* __restrict__ stats,
                                                               const float* __restrict__ weight,
                                                               const float* __restrict__ bias, float eps, int relu,
                                                               const int32_t* __restrict__ nbr, int K, float* __restrict__ out,
                                                               int64_t no_cap, const int32_t* __restrict__ no_dev,
                                                               float* __restrict__ out_amax) {
  const int64_t n = live_rows(no_cap, no_dev);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;                                  // C % 4 == 0 (checked by the entry point)
  float mx = 0.0f;
  if (t < n * c4) {
    const int64_t o = t / c4;
    const int c = (int)(t - o * c4) * 4;
    float mean[4], inv[4], w[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mean[j] = (float)stats[c + j];
      inv[j] = 1.0f / sqrtf((float)stats[C + c + j] + eps);
      w[j] = weight ? weight[c + j] : 1.0f;
      b[j] = bias ? bias[c + j] : 0.0f;
    }
    float m[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    bool any = false;
    for (int k = 0; k < K; ++k) {
      const int32_t s_ = nbr[o * K + k];
      if (s_ >= 0) {
        const float4 q = *reinterpret_cast<const float4*>(in + (int64_t)s_ * C + c);
        const float x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = (x[j] - mean[j]) * inv[j];
          if (weight) v = v * w[j];
          if (bias) v = v + b[j];
          if (relu) v = fmaxf(v, 0.0f);
          m[j] = fmaxf(m[j], v);
        }
        any = true;
      }
    }
    float4 r = any ? make_float4(m[0], m[1], m[2], m[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(out + o * C + c) = r;
    mx = fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  if (out_amax != nullptr) {
    __shared__ float sh4[4];
    block_amax_publish(out_amax, mx, sh4);
  }
}

// ---- BatchNorm (training) backward over the rows of a [n][C] matrix -----------------------------------------------------
// s1[c] = sum_r dy[r][c], s2[c] = sum_r dy[r][c] * x[r][c] in fp64, deterministic two-stage like colstats; then
// dgamma = (s2 - mean * s1) / sigma, dbeta = s1, dx = gamma / sigma * (dy - s1 / n - xhat * dgamma / n)
__global__ __launch_bounds__(256) void colsum2_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              int64_t n, int C, double* __restrict__ part) {
  const int c = threadIdx.x % C;
  const int groups = blockDim.x / C;
  const int gi = threadIdx.x / C;
  double s = 0.0, q = 0.0;
  if (gi < groups) {
    for (int64_t r = (int64_t)blockIdx.x * groups + gi; r < n; r += (int64_t)gridDim.x * groups) {
      const double g = (double)dy[r * C + c];
      s += g;
      q += g * (double)x[r * C + c];
    }
  }
  __shared__ double sm[2 * 256];
  sm[threadIdx.x] = s;
  sm[256 + threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < C) {
    double ts = 0.0, tq = 0.0;
    for (int g = 0; g < groups; ++g) { ts += sm[g * C + threadIdx.x]; tq += sm[256 + g * C + threadIdx.x]; }
    part[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = ts;
    part[(int64_t)blockIdx.x * 2 * C + C + threadIdx.x] = tq;
  }
}

__global__ __launch_bounds__(256) void bn_backward_final_kernel(const double* __restrict__ part, int nblk, int C, int64_t n,
                                                                const double* __restrict__ stats, float eps,
                                                                float* __restrict__ dweight, float* __restrict__ dbias,
                                                                double* __restrict__ sums) {
  const int c = blockIdx.x;
  __shared__ double ss[256], qq[256];
  double s = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) { s += part[(int64_t)b * 2 * C + c]; q += part[(int64_t)b * 2 * C + C + c]; }
  ss[threadIdx.x] = s; qq[threadIdx.x] = q;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) { ss[threadIdx.x] += ss[threadIdx.x + d]; qq[threadIdx.x] += qq[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = stats[c], inv = 1.0 / sqrt(stats[C + c] + (double)eps);
    const double dg = (qq[0] - mean * ss[0]) * inv;
    if (dweight) dweight[c] = (float)dg;
    if (dbias) dbias[c] = (float)ss[0];
    sums[c] = ss[0] / (double)n;          // mean(dy)
    sums[C + c] = dg / (double)n;         // mean(dy * xhat)
  }
}

__global__ __launch_bounds__(256) void bn_backward_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                int64_t n, int C, const double* __restrict__ stats,
                                                                const double* __restrict__ sums,
                                                                const float* __restrict__ weight, float eps,
                                                                float* __restrict__ dx) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const float mean = (float)stats[c];
    const float inv = 1.0f / sqrtf((float)stats[C + c] + eps);
    const float xhat = (x[t] - mean) * inv;
    const float g = weight ? weight[c] : 1.0f;
    dx[t] = g * inv * (dy[t] - (float)sums[c] - xhat * (float)sums[C + c]);
  }
}

// ---- BatchNorm (training) fused with what follows it in a residual block (round 5): y = [relu]( bn(x) [+ residual] ) ---------------
// forward: column statistics as above (fp64 sums in a fixed order); the final stage also updates the running statistics
// (running_var takes the unbiased estimate, as nn.BatchNorm1d) and the batch counter -- eight tiny torch launches otherwise;
// backward: with a ReLU behind the normalisation the incoming gradient is masked by y > 0 on the fly (y = the saved output),
// the masked gradient is also the residual branch's gradient.
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, int nblk, int C, int64_t n,
                                                             double* __restrict__ stats, float momentum,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             int64_t* __restrict__ batches) {
  const int c = blockIdx.x;
  __shared__ double ss[256], qq[256];
  double s = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) { s += part[(int64_t)b * 2 * C + c]; q += part[(int64_t)b * 2 * C + C + c]; }
  ss[threadIdx.x] = s; qq[threadIdx.x] = q;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) { ss[threadIdx.x] += ss[threadIdx.x + d]; qq[threadIdx.x] += qq[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double nn = (double)n;
    const double mean = ss[0] / nn;
    double var = qq[0] / nn - mean * mean;   // biased variance
    if (var < 0.0) var = 0.0;
    stats[c] = mean;
    stats[C + c] = var;
    if (running_mean != nullptr) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var != nullptr) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(var * (nn / (nn - 1.0)));
    if (batches != nullptr && c == 0) *batches += 1;
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ in, int64_t n, int C, const double* __restrict__ stats,
                                                       const float* __restrict__ weight, const float* __restrict__ bias, float eps,
                                                       const float* __restrict__ residual, int relu, float* __restrict__ out) {
  const int64_t total4 = n * C / 4;                         // C % 4 == 0 (checked by the entry point)
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((t * 4) % C);
    const float4 x = reinterpret_cast<const float4*>(in)[t];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (residual) r = reinterpret_cast<const float4*>(residual)[t];
    const float xv[4] = {x.x, x.y, x.z, x.w}, rv[4] = {r.x, r.y, r.z, r.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mean = (float)stats[c + j];
      const float inv = 1.0f / sqrtf((float)stats[C + c + j] + eps);
      float v = (xv[j] - mean) * inv;
      if (weight) v = v * weight[c + j];
      if (bias) v = v + bias[c + j];
      if (residual) v = v + rv[j];
      o[j] = relu == 1 ? fmaxf(v, 0.0f) : (relu == 2 ? (v > 0.0f ? v : expm1f(v)) : v);      // 2: ELU (alpha 1)
    }
    reinterpret_cast<float4*>(out)[t] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// s1 / s2 of the MASKED gradient (y given: dy counts where y > 0)
__global__ __launch_bounds__(256) void bn_colsum2_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ y, int act, int64_t n, int C,
                                                                 double* __restrict__ part) {
  const int c = threadIdx.x % C;
  const int groups = blockDim.x / C;
  const int gi = threadIdx.x / C;
  double s = 0.0, q = 0.0;
  if (gi < groups) {
    for (int64_t r = (int64_t)blockIdx.x * groups + gi; r < n; r += (int64_t)gridDim.x * groups) {
      float gf = dy[r * C + c];
      if (y != nullptr) {
        const float yv = y[r * C + c];
        if (!(yv > 0.0f)) gf = act == 2 ? gf * (yv + 1.0f) : 0.0f;        // ELU: d/dx = y + 1 below zero; ReLU: 0
      }
      const double g = (double)gf;
      s += g;
      q += g * (double)x[r * C + c];
    }
  }
  __shared__ double sm[2 * 256];
  sm[threadIdx.x] = s;
  sm[256 + threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < C) {
    double ts = 0.0, tq = 0.0;
    for (int g = 0; g < groups; ++g) { ts += sm[g * C + threadIdx.x]; tq += sm[256 + g * C + threadIdx.x]; }
    part[(int64_t)blockIdx.x * 2 * C + threadIdx.x] = ts;
    part[(int64_t)blockIdx.x * 2 * C + C + threadIdx.x] = tq;
  }
}

__global__ __launch_bounds__(256) void bn_backward_apply2_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ y, int act, int64_t n, int C,
                                                                 const double* __restrict__ stats, const double* __restrict__ sums,
                                                                 const float* __restrict__ weight, float eps,
                                                                 float* __restrict__ dx, float* __restrict__ dres) {
  const int64_t total4 = n * C / 4;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((t * 4) % C);
    const float4 g4 = reinterpret_cast<const float4*>(dy)[t], x4 = reinterpret_cast<const float4*>(x)[t];
    float gv[4] = {g4.x, g4.y, g4.z, g4.w};
    const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
    if (y != nullptr) {
      const float4 y4 = reinterpret_cast<const float4*>(y)[t];
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) if (!(yv[j] > 0.0f)) gv[j] = act == 2 ? gv[j] * (yv[j] + 1.0f) : 0.0f;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mean = (float)stats[c + j];
      const float inv = 1.0f / sqrtf((float)stats[C + c + j] + eps);
      const float xhat = (xv[j] - mean) * inv;
      const float g = weight ? weight[c + j] : 1.0f;
      o[j] = g * inv * (gv[j] - (float)sums[c + j] - xhat * (float)sums[C + c + j]);
    }
    reinterpret_cast<float4*>(dx)[t] = make_float4(o[0], o[1], o[2], o[3]);
    if (dres != nullptr) reinterpret_cast<float4*>(dres)[t] = make_float4(gv[0], gv[1], gv[2], gv[3]);
  }
}

__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ in, int64_t n_cap,
                                                     const int32_t* __restrict__ n_dev, int C, float* __restrict__ out) {
  const int64_t n = live_rows(n_cap, n_dev);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m = in[i * C];
  for (int c = 1; c < C; ++c) m = fmaxf(m, in[i * C + c]);
  out[i] = m;
}

// ================================================================================================================
// union-add, interpolation, pruning
// ================================================================================================================
__global__ __launch_bounds__(256) void union_flag_kernel(const int32_t* __restrict__ b_coords, int64_t nb_cap,
                                                         const int32_t* __restrict__ nb_dev,
                                                         const uint64_t* __restrict__ keys,
                                                         const int32_t* __restrict__ vals, int64_t cap,
                                                         int32_t* __restrict__ match, uint8_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb_cap) return;
  uint8_t f = 0;
  if (i < live_rows(nb_cap, nb_dev)) {
    int4 c = reinterpret_cast<const int4*>(b_coords)[i];
    int64_t s = hash_find(keys, cap, coord_key(c.x, c.y, c.z, c.w));
    match[i] = s >= 0 ? vals[s] : -1;
    f = s >= 0 ? 0 : 1;
  }
  flag[i] = f;
}

__global__ __launch_bounds__(256) void union_copy_a_kernel(const int32_t* __restrict__ a_coords,
                                                           const float* __restrict__ a_feats, int64_t na_cap,
                                                           const int32_t* __restrict__ na_dev, int C,
                                                           int32_t* __restrict__ out_coords,
                                                           float* __restrict__ out_feats) {
  const int64_t n = live_rows(na_cap, na_dev);
  if ((C & 3) == 0) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * (C >> 2); t += (int64_t)gridDim.x * blockDim.x)
      reinterpret_cast<float4*>(out_feats)[t] = reinterpret_cast<const float4*>(a_feats)[t];
  } else {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x)
      out_feats[t] = a_feats[t];
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<int4*>(out_coords)[t] = reinterpret_cast<const int4*>(a_coords)[t];
}

__global__ __launch_bounds__(256) void union_merge_b_kernel(const int32_t* __restrict__ b_coords,
                                                            const float* __restrict__ b_feats, int64_t nb_cap,
                                                            const int32_t* __restrict__ nb_dev, int C,
                                                            const int32_t* __restrict__ match,
                                                            const int32_t* __restrict__ idx, int64_t na_cap,
                                                            const int32_t* __restrict__ na_dev,
                                                            uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                            int64_t cap, int32_t* __restrict__ out_coords,
                                                            float* __restrict__ out_feats,
                                                            const int32_t* __restrict__ n_new, int32_t* __restrict__ n_out,
                                                            int64_t out_cap) {
  const int64_t nb = live_rows(nb_cap, nb_dev);
  const int64_t na = live_rows(na_cap, na_dev);
  if (blockIdx.x == 0 && threadIdx.x == 0) n_out[0] = (int32_t)(na + n_new[0]);
  if ((C & 3) == 0) {                                         // 4 channels per thread
    const int C4 = C >> 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nb * C4; t += (int64_t)gridDim.x * blockDim.x) {
      const int64_t i = t / C4;
      const int c = (int)(t - i * C4) * 4;
      const int32_t m = match[i];
      const float4 v = *reinterpret_cast<const float4*>(b_feats + i * C + c);
      if (m >= 0) {
        float4* d = reinterpret_cast<float4*>(out_feats + (int64_t)m * C + c);     // unique coords in B: one writer per element
        float4 o = *d;
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *d = o;
      } else {
        const int64_t row = na + idx[i];
        if (row >= out_cap) continue;                           // over the planned capacity: dropped (n_out tells)
        *reinterpret_cast<float4*>(out_feats + row * C + c) = v;
        if (c == 0) {
          int4 cc = reinterpret_cast<const int4*>(b_coords)[i];
          reinterpret_cast<int4*>(out_coords)[row] = cc;
          int64_t s = hash_insert(keys, cap, coord_key(cc.x, cc.y, cc.z, cc.w));
          if (s >= 0) vals[s] = (int32_t)row;
        }
      }
    }
    return;
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nb * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    const int32_t m = match[i];
    if (m >= 0) {
      out_feats[(int64_t)m * C + c] += b_feats[t];          // unique coords in B: one writer per element
    } else {
      const int64_t row = na + idx[i];
      if (row >= out_cap) continue;                           // over the planned capacity: dropped (n_out tells)
      out_feats[row * C + c] = b_feats[t];
      if (c == 0) {
        int4 cc = reinterpret_cast<const int4*>(b_coords)[i];
        reinterpret_cast<int4*>(out_coords)[row] = cc;
        int64_t s = hash_insert(keys, cap, coord_key(cc.x, cc.y, cc.z, cc.w));
        if (s >= 0) vals[s] = (int32_t)row;
      }
    }
  }
}

__global__ __launch_bounds__(256) void interp_kernel(const int32_t* __restrict__ q_coords, int64_t n_cap,
                                                     const int32_t* __restrict__ n_dev, const float* __restrict__ score,
                                                     const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                     int64_t cap, int s, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= live_rows(n_cap, n_dev)) return;
  int4 q = reinterpret_cast<const int4*>(q_coords)[i];
  auto fl = [s](int p) { int r = p / s; if ((p % s != 0) && (p < 0)) --r; return r * s; };
  const int bx = fl(q.y), by = fl(q.z), bz = fl(q.w);
  const float fs = (float)s;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int cx = bx + (k & 1) * s, cy = by + ((k >> 1) & 1) * s, cz = bz + ((k >> 2) & 1) * s;
    const float w = (1.0f - fabsf((float)(q.y - cx)) / fs) * (1.0f - fabsf((float)(q.z - cy)) / fs) *
                    (1.0f - fabsf((float)(q.w - cz)) / fs);
    if (w == 0.0f) continue;
    int64_t slot = hash_find(keys, cap, coord_key(q.x, cx, cy, cz));
    if (slot >= 0 && vals[slot] >= 0) acc += w * score[vals[slot]];
  }
  out[i] = acc;
}

__global__ __launch_bounds__(256) void prune_kernel(const int32_t* __restrict__ in_coords,
                                                    const float* __restrict__ in_feats, int64_t n_cap,
                                                    const int32_t* __restrict__ n_dev, int C,
                                                    const int32_t* __restrict__ sel, int32_t* __restrict__ out_coords,
                                                    float* __restrict__ out_feats) {
  const int64_t n = live_rows(n_cap, n_dev);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    const int32_t j = sel[i];
    if (j < 0) continue;
    out_feats[(int64_t)j * C + c] = in_feats[t];
    if (c == 0) reinterpret_cast<int4*>(out_coords)[j] = reinterpret_cast<const int4*>(in_coords)[i];
  }
}

inline unsigned grid_for(int64_t work, int block = 256, int64_t max_blocks = 65536) {
  int64_t b = ceil_div(work > 0 ? work : 1, block);
  return (unsigned)(b < max_blocks ? b : max_blocks);
}

}  // namespace

// ================================================================================================================
// C-ABI
// ================================================================================================================
// ================================================================================================================
// Backward of the sparse convolution (training).
//   dgrad: grad_in[i] = sum_k grad_out[nbrT[i][k]] @ W[k]^T -- the forward kernel on the TRANSPOSED neighbour table
//          (nbrT[i][k] = the output row o with nbr[o][k] == i; for a fixed offset the map o -> i is injective);
//   wgrad: gradW[k] = sum_o in[nbr[o][k]]^T (x) grad_out[o] -- a [Cin x rows] x [rows x Cout] product per offset on the
//          fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact products, fp32 accumulation), rows split into chunks whose
//          partial sums go to slabs that the caller adds up (deterministic, no atomics).
// ================================================================================================================
__global__ __launch_bounds__(256) void kernel_map_transpose_kernel(const int32_t* __restrict__ nbr, int64_t no_cap,
                                                                   const int32_t* __restrict__ no_dev, int K,
                                                                   int64_t n_in, int32_t* __restrict__ nbr_t) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= live_rows(no_cap, no_dev) * K) return;
  const int32_t i = nbr[t];
  if (i >= 0 && i < n_in) nbr_t[(int64_t)i * K + (t % K)] = (int32_t)(t / K);
}

// one block of 8 waves per (row chunk, offset, 64 x 64 tile of gradW[k]): the chunk's 32-row steps go round-robin to the waves,
// whose eight accumulator tiles are added through LDS in a fixed tree ((0+4)+(2+6))+((1+5)+(3+7)) -- deterministic, and one
// slab per BLOCK (the one-wave-per-block kernel of rounds 2-4 wrote one per wave: 125 MB of slabs per layer at any size,
// zeroed, written and read back = 19 GB of the step's traffic).
// The 1-D grid is decoded XCD-first (block b runs on XCD b % 8): all K offsets and all tiles of a chunk land on ONE XCD back
// to back, so the chunk's grad_out rows and the input rows around it come over the fabric once and are served from that
// XCD's L2 to the other K * tiles - 1 blocks (chunk-major order spread them over 8 L2s and over time: ~1 GB of fabric reads
// for a 73 k-row 64 -> 64 layer).
// Lane (m, kh): channels ci0 + 2m, ci0 + 2m + 1 of the tile's rows and co0 + 2m, co0 + 2m + 1 of its columns, one 8-byte
// load each per row (V2: both channel counts even) -- the MFMA does not care which channel sits in which matrix row as long
// as the store agrees: acc[x][y] holds (ci0 + 2 row + x, co0 + 2 col + y).
// BF16: operands rounded to bf16, fp32 accumulation on v_mfma_f32_32x32x16_bf16 (lane (m, kh) holds rows 8 kh .. 8 kh + 7 of
// a 16-row half step); else exact products on v_mfma_f32_32x32x2_f32 (lane (m, kh) holds row 2 s + kh of pair s).
constexpr int WG_WAVES = 8;
struct WgradMap { int chunks, per, tiles, tiles_co, by_chunk; };
template <bool BF16, bool V2>
__global__ __launch_bounds__(64 * WG_WAVES, 2) void conv_wgrad_block_kernel(
    const float* __restrict__ in, int Cin, const int32_t* __restrict__ nbr, int K, const float* __restrict__ gout, int Cout,
    int64_t no_cap, const int32_t* __restrict__ no_dev, int rows_per_chunk, float* __restrict__ slab, WgradMap map) {
  __shared__ float red[4 * 4096];
  const unsigned bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3;
  int chunk, k, tile;
  if (map.by_chunk) {                                                // many chunks: a chunk's K * tiles blocks on one XCD
    chunk = (int)(j / (unsigned)map.per) * 8 + (int)xcd;
    const int rem = (int)(j % (unsigned)map.per);
    k = rem / map.tiles; tile = rem % map.tiles;
  } else {                                                           // few chunks: a (chunk, offset)'s tiles on one XCD
    const int unit = (int)(j / (unsigned)map.tiles) * 8 + (int)xcd;
    chunk = unit / K; k = unit % K; tile = (int)(j % (unsigned)map.tiles);
  }
  if (chunk >= map.chunks) return;                                   // block-uniform, before any barrier
  const int ci0 = (tile / map.tiles_co) * 64, co0 = (tile % map.tiles_co) * 64;
  const int64_t n_live = live_rows(no_cap, no_dev);
  const int lane = threadIdx.x & 63, m = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  const int64_t r1 = min(n_live, r0 + rows_per_chunk);
  // out-of-range channels / missing rows load from a clamped (valid) address and are zeroed AFTER all loads of the step
  // are in flight: a select right behind each load makes the compiler wait for every load in turn
  const bool ci_ok0 = ci0 + 2 * m < Cin, ci_ok1 = ci0 + 2 * m + 1 < Cin, co_ok0 = co0 + 2 * m < Cout, co_ok1 = co0 + 2 * m + 1 < Cout;
  const int ca0 = ci_ok0 ? ci0 + 2 * m : 0, ca1 = ci_ok1 ? ci0 + 2 * m + 1 : 0;
  const int cb0 = co_ok0 ? co0 + 2 * m : 0, cb1 = co_ok1 ? co0 + 2 * m + 1 : 0;
  auto src_of = [&](int64_t o0) -> int32_t {              // lane l < 32: input row of output row o0 + l at this offset
    int32_t v = -1;
    if (lane < 32 && o0 + lane < r1) v = nbr ? nbr[(o0 + lane) * K + k] : (int32_t)(o0 + lane);
    return v;
  };
  int64_t o0 = r0 + 32 * wave;
  int32_t src_l = o0 < r1 ? src_of(o0) : -1;
  for (; o0 < r1; o0 += 32 * WG_WAVES) {
    float a0[16], a1[16], b0[16], b1[16];
    unsigned ok_a = 0, ok_b = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = BF16 ? 16 * (q >> 3) + 8 * kh + (q & 7) : 2 * q + kh;
      const int32_t src = __shfl(src_l, row, 64);
      const bool live = o0 + row < r1;
      const float* ap = in + (int64_t)(src < 0 ? 0 : src) * Cin;
      const float* gp = gout + (live ? o0 + row : r0) * Cout;
      if (V2) {
        const float2 ta = *reinterpret_cast<const float2*>(ap + ca0), tb = *reinterpret_cast<const float2*>(gp + cb0);
        a0[q] = ta.x; a1[q] = ta.y; b0[q] = tb.x; b1[q] = tb.y;
      } else {
        a0[q] = ap[ca0]; a1[q] = ap[ca1];
        b0[q] = gp[cb0]; b1[q] = gp[cb1];
      }
      ok_a |= (src >= 0 ? 1u : 0u) << q;
      ok_b |= (live ? 1u : 0u) << q;
    }
    const int64_t on = o0 + 32 * WG_WAVES;                // the next step's row numbers ride behind this step's operands
    src_l = on < r1 ? src_of(on) : -1;
    if (BF16) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8_t fa0, fa1, fb0, fb1;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int q = 8 * t + jj;
          const bool oa = (ok_a >> q) & 1u, ob = (ok_b >> q) & 1u;
          fa0[jj] = (__bf16)(oa && ci_ok0 ? a0[q] : 0.0f); fa1[jj] = (__bf16)(oa && ci_ok1 ? a1[q] : 0.0f);
          fb0[jj] = (__bf16)(ob && co_ok0 ? b0[q] : 0.0f); fb1[jj] = (__bf16)(ob && co_ok1 ? b1[q] : 0.0f);
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const bool oa = (ok_a >> q) & 1u, ob = (ok_b >> q) & 1u;
        const float va0 = oa && ci_ok0 ? a0[q] : 0.0f, va1 = oa && ci_ok1 ? a1[q] : 0.0f;
        const float vb0 = ob && co_ok0 ? b0[q] : 0.0f, vb1 = ob && co_ok1 ? b1[q] : 0.0f;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, vb0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, vb1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, vb0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, vb1, acc[1][1], 0, 0, 0);
      }
    }
  }
  // the eight waves' tiles: a fixed tree through LDS (element e of lane l at [e][l]: conflict-free)
  auto put = [&](int slot) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[slot * 4096 + ((x * 2 + y) * 16 + i) * 64 + lane] = acc[x][y][i];
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[x][y][i] += red[slot * 4096 + ((x * 2 + y) * 16 + i) * 64 + lane];
  };
  if (wave >= 4) put(wave - 4);
  __syncthreads();
  if (wave < 4) add(wave);
  __syncthreads();
  if (wave == 2 || wave == 3) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave != 0) return;
  add(0);
  float* dst = slab + ((int64_t)chunk * K + k) * Cin * Cout;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = ci0 + 2 * (8 * (i >> 2) + 4 * kh + (i & 3)) + x;
      const int co = co0 + 2 * m;
      if (ci >= Cin) continue;
      if (V2) {
        if (co < Cout) *reinterpret_cast<float2*>(dst + (int64_t)ci * Cout + co) = make_float2(acc[x][0][i], acc[x][1][i]);
      } else {
        if (co < Cout) dst[(int64_t)ci * Cout + co] = acc[x][0][i];
        if (co + 1 < Cout) dst[(int64_t)ci * Cout + co + 1] = acc[x][1][i];
      }
    }
}

template <bool BF16>
static int launch_wgrad(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* grad_out, int Cout,
                        int64_t no_cap, const int32_t* no_dev, int rows_per_chunk, float* slabs, void* stream) {
  if (Cin <= 0 || Cout <= 0 || K <= 0 || no_cap <= 0 || rows_per_chunk <= 0 || (rows_per_chunk & 1) || slabs == nullptr)
    return CNRMA_EINVAL;
  WgradMap map;
  map.chunks = (int)ceil_div(no_cap, rows_per_chunk);
  map.tiles_co = (int)ceil_div(Cout, 64);
  map.tiles = (int)ceil_div(Cin, 64) * map.tiles_co;
  map.per = K * map.tiles;
  map.by_chunk = map.chunks >= 32;
  const int64_t blocks = map.by_chunk ? ceil_div(map.chunks, 8) * (int64_t)map.per * 8
                                      : ceil_div((int64_t)map.chunks * K, 8) * map.tiles * 8;
  if (blocks > 0x7fffffffLL) return CNRMA_EINVAL;
  const bool v2 = !(Cin & 1) && !(Cout & 1);
  auto kern = v2 ? conv_wgrad_block_kernel<BF16, true> : conv_wgrad_block_kernel<BF16, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WG_WAVES), 0, as_stream(stream), in_feats, Cin, nbr, K, grad_out,
                     Cout, no_cap, no_dev, rows_per_chunk, slabs, map);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// ---- weight gradient on the gather-once structure (bf16 operands, same-coordinates 3x3x3 convolutions) -------------------------
// The block kernel above reads an input row once per (output row, offset) pair it takes part in -- ~9 times on a surface --
// and grad_out once per offset: 1 GB through the vector memory path for a 73 k-row 64 -> 64 layer, which is what bounds it.
// The tile unions of the forward convolution (tile_union_kernel: per 64-row output tile the DISTINCT input rows of its 27
// offsets + local indices) remove both: a block stages the tile's union rows (one 64-channel slice, bf16) and its 64
// grad_out rows ONCE in LDS, transposed ([channel][row]: the MFMA's reduction dimension is the row, so a lane's 8 values of
// a fragment are 8 rows of ONE channel -- contiguous for grad_out, an indexed read per row for the input: a dword = one channel PAIR), and
// seven waves run two offsets each over the image: A[ci][row r] = image[ci][lidx[r][k]], a missing neighbour indexes a zero
// column.  A block owns (part = a range of tiles, 14 of the 27 offsets, 64 x 64 tile of the matrix) and keeps its accumulators over
// all tiles of the part: one slab per part, every element written by exactly one wave, no reduction inside the block.
constexpr int WGO_PITCH = 289;      // DWORDS per channel pair of the input image: 280 union rows + the zero column (280); odd: the 32
                                    // channel pairs a fragment read touches fall into distinct banks.  A dword holds the bf16
                                    // values of channels 2m (low half) and 2m + 1 of one union row: the two 32-channel halves
                                    // of the A operand come out of ONE indexed read per row (the reads bound the consumers:
                                    // 2-byte reads, one per half, took 2.9 us per tile against 0.4 us of MFMAs)
constexpr int WGO_BP = 72;          // u16 per channel of the grad_out image: 64 rows + 8 (144 B: 16-byte aligned fragment reads)
constexpr int WGO_KPW = 2;          // offsets per consumer wave (two accumulator sets: 128 registers)
constexpr int WGO_CW = 7;           // consumer waves: 14 offsets per block
constexpr int WGO_PW = 5;           // producer waves: 320 threads stage the next tile while the consumers run this one
constexpr int WGO_KPB = WGO_KPW * WGO_CW;
constexpr int WGO_KG = 2;           // offset groups: 2 x 14 >= 27
constexpr int WGO_PT = 64 * WGO_PW;
constexpr int WGO_IT = (GO_UMAX / 2 * 16 + WGO_PT - 1) / WGO_PT;   // (row pair, 4 channels) tasks per producer thread: 7
static_assert(WGO_PT % 16 == 0, "a producer thread keeps its channel quad over its tasks");
struct WgoMap { int parts, tiles_per_part, n_ci, n_co, by_part, ablate; };
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

// Roles: the dependent chain header -> union row numbers -> rows -> LDS costs three round trips per tile; done in place by
// the waves that also hold the accumulators it took 6-8 us per tile and block against 2 us of LDS reads + MFMAs (measured with
// the phases switched off one at a time, scripts/wgrad_go_ablate.py; a register pipeline in the same waves spilled and its
// scratch reloads wait for vmcnt(0), i.e. for the very loads they were meant to overlap).  So the block is split:
//   5 producer waves stage tile t + 1 into the other half of a double-buffered image (their registers hold nothing else);
//   7 consumer waves run their 2 offsets each over tile t;
// one barrier per stage: after barrier s the consumers read buffer s & 1 while the producers fill buffer (s + 1) & 1.
// A stage = (tile, offset group); compact tiles have one group.  The producers publish {live, offset mask} per stage; a zero
// `live` ends the consumers' loop (both sides pass the same number of barriers).
__global__ __launch_bounds__(64 * (WGO_CW + WGO_PW), 3) void conv_wgrad_go_kernel(
    const float* __restrict__ in, int Cin, const float* __restrict__ gout, int Cout, int64_t no_cap,
    const int32_t* __restrict__ no_dev, const int32_t* __restrict__ hdr, const int32_t* __restrict__ urows,
    const uint16_t* __restrict__ lidx, float* __restrict__ slab, WgoMap map) {
  __shared__ __attribute__((aligned(16))) unsigned At[2][32 * WGO_PITCH];
  __shared__ __attribute__((aligned(16))) uint16_t Bt[2][64 * WGO_BP];
  __shared__ __attribute__((aligned(16))) uint16_t Ls[2][WGO_KPB * 64];
  __shared__ int meta[2][2];
  const int per = WGO_KG * map.n_ci * map.n_co;
  const unsigned bid = blockIdx.x;
  int part, rem;
  if (map.by_part) {                                                 // a part's blocks on one XCD: they stage the same rows
    const unsigned xcd = bid & 7u, j = bid >> 3;
    part = (int)(j / (unsigned)per) * 8 + (int)xcd; rem = (int)(j % (unsigned)per);
  } else {
    part = (int)(bid / (unsigned)per); rem = (int)(bid % (unsigned)per);
  }
  if (part >= map.parts) return;
  const int g = rem / (map.n_ci * map.n_co), t2 = rem % (map.n_ci * map.n_co);
  const int ci0 = (t2 / map.n_co) * 64, co0 = (t2 % map.n_co) * 64;
  const int64_t n_live = live_rows(no_cap, no_dev);
  const int64_t n_tiles = (n_live + GO_BM - 1) / GO_BM;
  const int64_t t_begin = (int64_t)part * map.tiles_per_part;
  const int64_t t_end = min(n_tiles, t_begin + map.tiles_per_part);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned kmask_block = (((1u << WGO_KPB) - 1u) << (g * WGO_KPB)) & ((1u << 27) - 1u);
  for (int i = tid; i < 2 * 32 * (WGO_PITCH - GO_UMAX); i += 64 * (WGO_CW + WGO_PW)) {   // the zero columns (+ padding) of both images
    const int bufi = i / (32 * (WGO_PITCH - GO_UMAX)), e = i % (32 * (WGO_PITCH - GO_UMAX));
    At[bufi][(e / (WGO_PITCH - GO_UMAX)) * WGO_PITCH + GO_UMAX + e % (WGO_PITCH - GO_UMAX)] = 0u;
  }
  __syncthreads();

  if (wave >= WGO_CW) {
    // ------------------------------------------------ producers ------------------------------------------------
    // The producers run their own pipeline, one stage deep on each level of the dependent chain: while image t is being
    // written, the rows of tile t + 1 are already requested (right after the last LDS store of tile t, before the barrier:
    // their flight overlaps the wait for the consumers), and header + union row numbers of tile t + 2 behind them.
    // Only a tile's first offset group rides the pipeline; further groups (a tile without locality) are staged in place.
    const int ptid = tid - 64 * WGO_CW;
    const int aq = ptid & 15, ac = ci0 + 4 * aq, bc = co0 + 4 * aq;  // this thread's 4 channels / 4 columns
    const bool okc = ac < Cin, okb = bc < Cout;
    const int acs = okc ? ac : 0;
    int stage = 0;
    int4 h_i = make_int4(0, 0, 0, 0);      // header {groups, mask, first (0), rows} of the tile whose row numbers are in id0 / id1
    int32_t id0[WGO_IT], id1[WGO_IT];
    int4 h_r = make_int4(0, 0, 0, 0);      // header of the tile whose rows are in f0 / f1 / gb0 / gb1 / lsv
    float4 f0[WGO_IT], f1[WGO_IT], gb0[2], gb1[2];
    uint16_t lsv[3];
    auto load_ids = [&](int64_t tile) {    // header + first group's row numbers (entry 0 on: no dependency on the header)
      if (tile >= t_end) { h_i = make_int4(0, 0, 0, 0); return; }
      h_i = *reinterpret_cast<const int4*>(hdr + tile * GO_HDR);
      const int32_t* tr = urows + tile * GO_ROWS;
#pragma unroll
      for (int it = 0; it < WGO_IT; ++it) {
        const int u = 2 * ((ptid + it * WGO_PT) >> 4);               // < 280: inside the tile's GO_ROWS entries whatever its count
        id0[it] = tr[u]; id1[it] = tr[u + 1];
      }
    };
    auto load_tile_rows = [&](int64_t tile, int rows_here) {         // grad_out tile + the block's 14 columns of local indices
      const int64_t tile0 = tile * GO_BM;
#pragma unroll
      for (int e = 0; e < 2; ++e) {                                  // task = (row pair, 4 columns), 512 tasks
        const int bt = ptid + e * WGO_PT, br = 2 * (bt >> 4);
        gb0[e] = make_float4(0.f, 0.f, 0.f, 0.f); gb1[e] = gb0[e];
        if (bt < 512 && okb && br < rows_here) gb0[e] = *reinterpret_cast<const float4*>(gout + (tile0 + br) * Cout + bc);
        if (bt < 512 && okb && br + 1 < rows_here) gb1[e] = *reinterpret_cast<const float4*>(gout + (tile0 + br + 1) * Cout + bc);
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int idx = ptid + e * WGO_PT, kk = idx >> 6, r = idx & 63, kg = g * WGO_KPB + kk;
        lsv[e] = (uint16_t)GO_UMAX;
        if (kk < WGO_KPB && kg < 27 && r < rows_here) lsv[e] = lidx[tile * (GO_BM * 27) + r * 27 + kg];
      }
    };
    auto load_rows = [&](int64_t tile) {   // the rows of `tile` by the numbers in id0 / id1 (h_i -> h_r)
      h_r = h_i;
      if (tile >= t_end) return;
      const int cnt = h_i.w;
      if (!(map.ablate & 1024)) {
#pragma unroll
        for (int it = 0; it < WGO_IT; ++it) {                        // entries beyond the list: row 0 (a valid address), zeroed at the store
          const int u = 2 * ((ptid + it * WGO_PT) >> 4);
          const int32_t ra = u < cnt ? id0[it] : 0, rb = u + 1 < cnt ? id1[it] : 0;
          f0[it] = *reinterpret_cast<const float4*>(in + (int64_t)ra * Cin + acs);
          f1[it] = *reinterpret_cast<const float4*>(in + (int64_t)rb * Cin + acs);
        }
      }
      load_tile_rows(tile, (int)min((int64_t)GO_BM, n_live - tile * GO_BM));
    };
    // LDS stores off ONE base per image and constant offsets (a thread's tasks are 20 row pairs apart, its 4 channels one
    // channel pitch): the compiler otherwise keeps 39 precomputed addresses in registers -- and spills
    static_assert(WGO_PT / 16 == 20 && WGO_BP % 2 == 0, "store offsets");
    auto store_tile_rows = [&](int buf) {
      unsigned* bb = reinterpret_cast<unsigned*>(Bt[buf]) + (4 * aq) * (WGO_BP / 2) + (ptid >> 4);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (ptid + e * WGO_PT < 512) {
          bb[0 * (WGO_BP / 2) + 20 * e] = pack_bf16x2(gb0[e].x, gb1[e].x);
          bb[1 * (WGO_BP / 2) + 20 * e] = pack_bf16x2(gb0[e].y, gb1[e].y);
          bb[2 * (WGO_BP / 2) + 20 * e] = pack_bf16x2(gb0[e].z, gb1[e].z);
          bb[3 * (WGO_BP / 2) + 20 * e] = pack_bf16x2(gb0[e].w, gb1[e].w);
        }
      }
      uint16_t* lb = Ls[buf] + ptid;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (ptid + e * WGO_PT < WGO_KPB * 64) lb[e * WGO_PT] = lsv[e];                       // [offset][row]
    };
    auto store_rows = [&](int buf) {
      const int cnt = h_r.w;
      store_tile_rows(buf);
      unsigned* ab = At[buf] + (2 * aq) * WGO_PITCH + 2 * (ptid >> 4);    // channel pairs 2 aq, 2 aq + 1; rows u, u + 1
      const int u_base = 2 * (ptid >> 4);
#pragma unroll
      for (int it = 0; it < WGO_IT; ++it) {
        const int u = u_base + 40 * it;
        if (u < cnt) {
          const bool k1 = okc && u + 1 < cnt;
          ab[40 * it] = pack_bf16x2(okc ? f0[it].x : 0.f, okc ? f0[it].y : 0.f);
          ab[40 * it + 1] = pack_bf16x2(k1 ? f1[it].x : 0.f, k1 ? f1[it].y : 0.f);
          ab[WGO_PITCH + 40 * it] = pack_bf16x2(okc ? f0[it].z : 0.f, okc ? f0[it].w : 0.f);
          ab[WGO_PITCH + 40 * it + 1] = pack_bf16x2(k1 ? f1[it].z : 0.f, k1 ? f1[it].w : 0.f);
        }
      }
    };
    load_ids(t_begin);
    load_rows(t_begin);
    load_ids(t_begin + 1);
    for (int64_t tile = t_begin; tile < t_end; ++tile) {
      const int ng = h_r.x;
      const unsigned mask0 = (unsigned)h_r.y;
      const bool live0 = ng > 0 && (mask0 & kmask_block) != 0u;
      if (live0 && !(map.ablate & 512)) store_rows(stage & 1);
      load_rows(tile + 1);                                           // requested before the barrier: in flight while we wait
      load_ids(tile + 2);
      if (live0) {
        if (ptid == 0) { meta[stage & 1][0] = 1; meta[stage & 1][1] = (int)mask0; }
        __syncthreads();                                             // barrier `stage`: this image is complete
        ++stage;
      }
      if (ng > 1) {                                                  // the other groups of a tile without locality: in place
        const int64_t tile0 = tile * GO_BM;
        const int rows_here = (int)min((int64_t)GO_BM, n_live - tile0);
        const int32_t* th = hdr + tile * GO_HDR;
        for (int gi = 1; gi < ng; ++gi) {
          const unsigned mask = (unsigned)th[1 + 3 * gi];
          if ((mask & kmask_block) == 0u) continue;                  // block-uniform
          const int first = th[2 + 3 * gi], cnt = th[3 + 3 * gi];
          const int32_t* tr = urows + tile * GO_ROWS + first;
          const int buf = stage & 1;
          unsigned* b32 = reinterpret_cast<unsigned*>(Bt[buf]);
          for (int bt = ptid; bt < 512; bt += WGO_PT) {
            const int br = 2 * (bt >> 4);
            float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
            if (okb && br < rows_here) x0 = *reinterpret_cast<const float4*>(gout + (tile0 + br) * Cout + bc);
            if (okb && br + 1 < rows_here) x1 = *reinterpret_cast<const float4*>(gout + (tile0 + br + 1) * Cout + bc);
            b32[((4 * aq + 0) * WGO_BP + br) >> 1] = pack_bf16x2(x0.x, x1.x);
            b32[((4 * aq + 1) * WGO_BP + br) >> 1] = pack_bf16x2(x0.y, x1.y);
            b32[((4 * aq + 2) * WGO_BP + br) >> 1] = pack_bf16x2(x0.z, x1.z);
            b32[((4 * aq + 3) * WGO_BP + br) >> 1] = pack_bf16x2(x0.w, x1.w);
          }
          for (int idx = ptid; idx < WGO_KPB * 64; idx += WGO_PT) {
            const int kk = idx >> 6, r = idx & 63, kg = g * WGO_KPB + kk;
            Ls[buf][idx] = kg < 27 && r < rows_here ? lidx[tile * (GO_BM * 27) + r * 27 + kg] : (uint16_t)GO_UMAX;
          }
          unsigned* a32 = At[buf];
          const int tasks = ((cnt + 1) >> 1) * 16;
          for (int task = ptid; task < tasks; task += WGO_PT) {
            const int u = 2 * (task >> 4);
            const int32_t r0 = tr[u], r1 = u + 1 < cnt ? tr[u + 1] : -1;
            const float4 x0 = *reinterpret_cast<const float4*>(in + (int64_t)r0 * Cin + acs);
            const float4 x1 = *reinterpret_cast<const float4*>(in + (int64_t)(r1 >= 0 ? r1 : r0) * Cin + acs);
            const bool k1 = okc && r1 >= 0;
            a32[(2 * aq) * WGO_PITCH + u] = pack_bf16x2(okc ? x0.x : 0.f, okc ? x0.y : 0.f);
            a32[(2 * aq) * WGO_PITCH + u + 1] = pack_bf16x2(k1 ? x1.x : 0.f, k1 ? x1.y : 0.f);
            a32[(2 * aq + 1) * WGO_PITCH + u] = pack_bf16x2(okc ? x0.z : 0.f, okc ? x0.w : 0.f);
            a32[(2 * aq + 1) * WGO_PITCH + u + 1] = pack_bf16x2(k1 ? x1.z : 0.f, k1 ? x1.w : 0.f);
          }
          if (ptid == 0) { meta[buf][0] = 1; meta[buf][1] = (int)mask; }
          __syncthreads();
          ++stage;
        }
      }
    }
    if (ptid == 0) meta[stage & 1][0] = 0;
    __syncthreads();
    return;
  }

  // -------------------------------------------------- consumers --------------------------------------------------
  const int lane = tid & 63, m = lane & 31, kh = lane >> 5;
  const int k0 = g * WGO_KPB + WGO_KPW * wave;
  const bool has0 = k0 < 27, has1 = k0 + 1 < 27;
  f32x16 acc[WGO_KPW][2][2];
#pragma unroll
  for (int e = 0; e < WGO_KPW; ++e)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[e][a][b][i] = 0.0f;
  for (int stage = 0;; ++stage) {
    __syncthreads();                                                 // barrier `stage`
    const int buf = stage & 1;
    if (meta[buf][0] == 0) break;
    const unsigned mask = (unsigned)meta[buf][1];
    const bool do0 = has0 && ((mask >> k0) & 1u), do1 = has1 && ((mask >> (k0 + 1)) & 1u);
    if ((!do0 && !do1) || (map.ablate & 256)) continue;
    const unsigned* A = At[buf] + m * WGO_PITCH;
    const uint16_t* B = Bt[buf];
    const uint16_t* L = Ls[buf];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const bf16x8_t fb0 = *reinterpret_cast<const bf16x8_t*>(&B[m * WGO_BP + 16 * t4 + 8 * kh]);
      const bf16x8_t fb1 = *reinterpret_cast<const bf16x8_t*>(&B[(32 + m) * WGO_BP + 16 * t4 + 8 * kh]);
#pragma unroll
      for (int e = 0; e < WGO_KPW; ++e) {
        if (!(e == 0 ? do0 : do1)) continue;
        const u16x8_t lv = *reinterpret_cast<const u16x8_t*>(&L[(WGO_KPW * wave + e) * 64 + 16 * t4 + 8 * kh]);
        unsigned w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = A[lv[j]];                 // channels 2m | 2m + 1 of row j's neighbour
        if (map.ablate & 2048) continue;                             // diagnostics: the LDS reads without the MFMAs
        u32x4_t ra0, ra1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          ra0[jj] = __builtin_amdgcn_perm(w[2 * jj + 1], w[2 * jj], 0x05040100u);     // low halves
          ra1[jj] = __builtin_amdgcn_perm(w[2 * jj + 1], w[2 * jj], 0x07060302u);     // high halves
        }
        acc[e][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ra0), fb0, acc[e][0][0], 0, 0, 0);
        acc[e][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ra0), fb1, acc[e][0][1], 0, 0, 0);
        acc[e][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ra1), fb0, acc[e][1][0], 0, 0, 0);
        acc[e][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ra1), fb1, acc[e][1][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < WGO_KPW; ++e) {
    if (!(e == 0 ? has0 : has1)) continue;
    float* dst = slab + ((int64_t)part * 27 + k0 + e) * Cin * Cout;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int co = co0 + y * 32 + m;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = ci0 + 2 * (8 * (i >> 2) + 4 * kh + (i & 3)) + x;        // matrix row r of tile x = channel 2 r + x
          if (ci < Cin && co < Cout) dst[(int64_t)ci * Cout + co] = acc[e][x][y][i];
        }
      }
  }
}

extern "C" int cnrma_sparse_kernel_map_transpose(const int32_t* nbr, int64_t no_cap, const int32_t* no_dev, int K,
                                                 int64_t n_in, int32_t* nbr_t, void* stream) {
  if (no_cap <= 0 || K <= 0 || n_in <= 0 || nbr == nullptr || nbr_t == nullptr) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  hipError_t e = cnrma_fill_bytes(nbr_t, 0xFF, (size_t)n_in * K * sizeof(int32_t), st);
  if (e != hipSuccess) return -(int)e;
  hipLaunchKernelGGL(kernel_map_transpose_kernel, dim3((unsigned)ceil_div(no_cap * K, 256)), dim3(256), 0, st, nbr, no_cap,
                     no_dev, K, n_in, nbr_t);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_wgrad_chunks(int64_t no_cap, int rows_per_chunk) {
  return rows_per_chunk > 0 ? (int)ceil_div(no_cap, rows_per_chunk) : 0;
}

extern "C" int cnrma_sparse_conv_wgrad_f32(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* grad_out,
                                           int Cout, int64_t no_cap, const int32_t* no_dev, int rows_per_chunk,
                                           float* slabs, void* stream) {
  return launch_wgrad<false>(in_feats, Cin, nbr, K, grad_out, Cout, no_cap, no_dev, rows_per_chunk, slabs, stream);
}

extern "C" int cnrma_sparse_conv_wgrad_bf16(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* grad_out,
                                            int Cout, int64_t no_cap, const int32_t* no_dev, int rows_per_chunk,
                                            float* slabs, void* stream) {
  return launch_wgrad<true>(in_feats, Cin, nbr, K, grad_out, Cout, no_cap, no_dev, rows_per_chunk, slabs, stream);
}

extern "C" size_t cnrma_voxelize_workspace_bytes(int64_t M) {
  int64_t n4 = (M + 3) / 4 * 4;
  return (size_t)(n4 * 33) + sort_temp_bytes(M) + cnrma_scan_workspace_bytes(M) + 1024 + 256;
}

extern "C" int cnrma_voxelize_f32(const float* coords, const float* feats, int64_t M, const int32_t* m_dev, int C,
                                  float voxel_size, int batch_id, int row_order, uint64_t* hash_keys, int32_t* hash_vals,
                                  int64_t hash_cap, int32_t* out_coords, float* out_feats, int32_t* out_src,
                                  int64_t out_cap, int32_t* n_out, void* workspace, void* stream) {
  if (out_cap <= 0 || out_cap > M) out_cap = M;
  if (C <= 0 || !(voxel_size > 0.0f) || out_src == nullptr || row_order < 0 || row_order > 1) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  int rc = run_unique<0>(coords, M, m_dev, voxel_size, 1, batch_id, hash_keys, hash_vals, hash_cap, out_coords,
                         out_src, n_out, workspace, st, row_order == 1, out_cap);
  if (rc) return rc;
  if (feats && out_feats) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(out_cap * (int64_t)(C / 4 + 1))), dim3(256), 0, st, feats, out_src,
                       out_cap, n_out, C, out_feats);
    CNRMA_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int cnrma_sparse_build_map(const int32_t* coords, int64_t n_cap, const int32_t* n_dev, uint64_t* hash_keys,
                                      int32_t* hash_vals, int64_t hash_cap, int precleared, void* stream) {
  if (n_cap <= 0 || hash_cap < 2 || (hash_cap & (hash_cap - 1)) != 0) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (!precleared) {                                         // precleared: the caller's arena holds 0xFF bytes already (cnrma_fill_bytes_u8)
    hipError_t e = cnrma_fill_bytes(hash_keys, 0xFF, (size_t)hash_cap * sizeof(uint64_t), st);
    if (e != hipSuccess) return -(int)e;
  }
  hipLaunchKernelGGL(build_map_kernel, dim3((unsigned)ceil_div(n_cap, 256)), dim3(256), 0, st, coords, n_cap, n_dev,
                     hash_keys, hash_vals, hash_cap);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_stride_coords(const int32_t* in_coords, int64_t n_cap, const int32_t* n_dev,
                                          int new_stride, uint64_t* hash_keys, int32_t* hash_vals, int64_t hash_cap,
                                          int32_t* out_coords, int64_t out_cap, int32_t* n_out, void* workspace,
                                          void* stream) {
  if (new_stride <= 0) return CNRMA_EINVAL;
  return run_unique<1>(in_coords, n_cap, n_dev, 1.0f, new_stride, 0, hash_keys, hash_vals, hash_cap, out_coords,
                       nullptr, n_out, workspace, as_stream(stream), false, out_cap);
}

// ---- strided coordinates of a set whose rows are SORTED by morton_key (the voxeliser's row order and everything strided from
// it): the parents' keys are non-decreasing along the rows, so "first row of its parent" is an adjacent comparison -- no
// hash table, no atomics (cnrma_sparse_stride_coords inserts every input row, ~6 per parent, with an atomic minimum each)
__global__ __launch_bounds__(256) void stride_sorted_flag_kernel(const int32_t* __restrict__ in_coords, int64_t n_cap,
                                                                 const int32_t* __restrict__ n_dev, int new_stride,
                                                                 uint8_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  uint8_t f = 0;
  if (i < live_rows(n_cap, n_dev)) {
    int b, x, y, z;
    quantise<1>(in_coords, i, 1.0f, new_stride, 0, &b, &x, &y, &z);
    f = 1;
    if (i > 0) {
      int b0, x0, y0, z0;
      quantise<1>(in_coords, i - 1, 1.0f, new_stride, 0, &b0, &x0, &y0, &z0);
      f = (b != b0 || x != x0 || y != y0 || z != z0) ? 1 : 0;
    }
  }
  flag[i] = f;
}

__global__ __launch_bounds__(256) void stride_sorted_write_kernel(const int32_t* __restrict__ in_coords, int64_t n_cap,
                                                                  const int32_t* __restrict__ n_dev, int new_stride,
                                                                  const int32_t* __restrict__ idx,
                                                                  int32_t* __restrict__ out_coords, int64_t out_cap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= live_rows(n_cap, n_dev)) return;
  const int32_t j = idx[i];
  if (j < 0 || j >= out_cap) return;              // beyond the planned capacity: dropped (n_out says so)
  int b, x, y, z;
  quantise<1>(in_coords, i, 1.0f, new_stride, 0, &b, &x, &y, &z);
  reinterpret_cast<int4*>(out_coords)[j] = make_int4(b, x, y, z);
}

// the same for short sets (n_cap <= STRIDE_SMALL_MAX: the coarse levels) as ONE block and ONE launch instead of four (flag,
// two scan launches, write): a thread owns a contiguous run of rows, flags them against their predecessors, the block scans
// the counts, the run's first-of-parent rows are written in order.  Same rows, same order, same n_out.
constexpr int STRIDE_SMALL_MAX = 16384;
__global__ __launch_bounds__(1024) void stride_sorted_small_kernel(const int32_t* __restrict__ in_coords, int n_cap,
                                                                   const int32_t* __restrict__ n_dev, int new_stride,
                                                                   int32_t* __restrict__ out_coords, int out_cap,
                                                                   int32_t* __restrict__ n_out) {
  __shared__ int smem[1024 / 64 + 1];
  const int n = (int)live_rows(n_cap, n_dev);
  const int per = (n_cap + 1023) / 1024;                   // <= 16
  const int lo = threadIdx.x * per, hi = min(lo + per, n);
  unsigned flags = 0;
  int pb = 0, px = 0, py = 0, pz = 0;
  if (lo < hi && lo > 0) quantise<1>(in_coords, lo - 1, 1.0f, new_stride, 0, &pb, &px, &py, &pz);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) {
    int b, x, y, z;
    quantise<1>(in_coords, i, 1.0f, new_stride, 0, &b, &x, &y, &z);
    const bool f = i == 0 || b != pb || x != px || y != py || z != pz;
    flags |= (f ? 1u : 0u) << (i - lo);
    cnt += f ? 1 : 0;
    pb = b; px = x; py = y; pz = z;
  }
  int total;
  int run = block_excl_scan<1024>(cnt, smem, &total);
  for (int i = lo; i < hi; ++i) {
    if ((flags >> (i - lo)) & 1u) {
      if (run < out_cap) {                                 // beyond the planned capacity: dropped (n_out says so)
        int b, x, y, z;
        quantise<1>(in_coords, i, 1.0f, new_stride, 0, &b, &x, &y, &z);
        reinterpret_cast<int4*>(out_coords)[run] = make_int4(b, x, y, z);
      }
      ++run;
    }
  }
  if (threadIdx.x == 0) n_out[0] = total;
}

extern "C" int cnrma_sparse_stride_coords_sorted(const int32_t* in_coords, int64_t n_cap, const int32_t* n_dev, int new_stride,
                                                 int32_t* out_coords, int64_t out_cap, int32_t* n_out, void* workspace,
                                                 void* stream) {
  if (new_stride <= 0 || n_cap <= 0 || in_coords == nullptr || out_coords == nullptr || n_out == nullptr) return CNRMA_EINVAL;
  if (out_cap <= 0 || out_cap > n_cap) out_cap = n_cap;
  hipStream_t st = as_stream(stream);
  if (n_cap <= STRIDE_SMALL_MAX) {
    hipLaunchKernelGGL(stride_sorted_small_kernel, dim3(1), dim3(1024), 0, st, in_coords, (int)n_cap, n_dev, new_stride, out_coords,
                       (int)out_cap, n_out);
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
  UniqueWs w = carve_unique_ws(workspace, n_cap);
  const unsigned nb = (unsigned)ceil_div(n_cap, 256);
  hipLaunchKernelGGL(stride_sorted_flag_kernel, dim3(nb), dim3(256), 0, st, in_coords, n_cap, n_dev, new_stride, w.flag);
  int rc = cnrma_mask_to_index(w.flag, w.idx, n_out, n_cap, w.scan, st);
  if (rc) return rc;
  hipLaunchKernelGGL(stride_sorted_write_kernel, dim3(nb), dim3(256), 0, st, in_coords, n_cap, n_dev, new_stride, w.idx,
                     out_coords, out_cap);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_kernel_map(const int32_t* out_coords, int64_t no_cap, const int32_t* no_dev,
                                       const uint64_t* in_hash_keys, const int32_t* in_hash_vals, int64_t hash_cap,
                                       const int32_t* offsets, int K, int32_t* nbr, void* stream) {
  if (no_cap <= 0 || K <= 0) return CNRMA_EINVAL;
  hipLaunchKernelGGL(kernel_map_kernel, dim3((unsigned)ceil_div(no_cap * K, 256)), dim3(256), 0, as_stream(stream),
                     out_coords, no_cap, no_dev, in_hash_keys, in_hash_vals, hash_cap, offsets, K, nbr);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_kernel_map_symmetric(const int32_t* coords, int64_t n_cap, const int32_t* n_dev,
                                                 const uint64_t* hash_keys, const int32_t* hash_vals, int64_t hash_cap,
                                                 const int32_t* offsets, int K, int32_t* nbr, int precleared, void* stream) {
  if (n_cap <= 0 || K <= 0 || (K & 1) == 0) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  if (!precleared) {
    hipError_t e = cnrma_fill_bytes(nbr, 0xFF, (size_t)n_cap * K * sizeof(int32_t), st);
    if (e != hipSuccess) return -(int)e;
  }
  hipLaunchKernelGGL(kernel_map_symmetric_kernel, dim3((unsigned)ceil_div(n_cap * (K / 2 + 1), 256)), dim3(256), 0, st,
                     coords, n_cap, n_dev, hash_keys, hash_vals, hash_cap, offsets, K, nbr);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// 3x3x3 table of a GENERATED child set (conv_transpose_generative: all 8 children of every parent, child m of parent p at
// row 8 p + m, m = x << 2 | y << 1 | z) from the parents' own 3x3x3 table: the neighbour of child d at offset o is child
// (d + o) mod 2 of the parent at offset floor((d + o) / 2), per axis -- no hash table, no probes, one streaming pass
__global__ __launch_bounds__(256) void kernel_map_children_kernel(const int32_t* __restrict__ parent_nbr, int64_t np_cap,
                                                                  const int32_t* __restrict__ np_dev,
                                                                  int32_t* __restrict__ nbr) {
  const int64_t n = live_rows(np_cap, np_dev) * 8;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * 27) return;
  const int64_t t = e / 27;
  const int k = (int)(e - t * 27);
  const int64_t p = t >> 3;
  const int m = (int)(t & 7);
  const int sx = ((m >> 2) & 1) + k % 3 - 1, sy = ((m >> 1) & 1) + (k / 3) % 3 - 1, sz = (m & 1) + k / 9 - 1;   // in {-1, 0, 1, 2}
  const int px = (sx + 2) / 2 - 1, py = (sy + 2) / 2 - 1, pz = (sz + 2) / 2 - 1;                                  // floor(s / 2)
  const int kp = (pz + 1) * 9 + (py + 1) * 3 + (px + 1);
  const int mc = ((sx & 1) << 2) | ((sy & 1) << 1) | (sz & 1);
  const int32_t j = parent_nbr[p * 27 + kp];
  nbr[e] = j >= 0 ? 8 * j + mc : -1;
}

extern "C" int cnrma_sparse_kernel_map_children(const int32_t* parent_nbr, int64_t np_cap, const int32_t* np_dev, int32_t* nbr,
                                                void* stream) {
  if (parent_nbr == nullptr || nbr == nullptr || np_cap <= 0 || np_cap * 8 * 27 >= ((int64_t)1 << 40)) return CNRMA_EINVAL;
  hipLaunchKernelGGL(kernel_map_children_kernel, dim3((unsigned)ceil_div(np_cap * 8 * 27, 256)), dim3(256), 0, as_stream(stream),
                     parent_nbr, np_cap, np_dev, nbr);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_kernel_map_strided(const int32_t* in_coords, int64_t n_cap, const int32_t* n_dev,
                                               int in_stride, int kernel_size, const uint64_t* out_hash_keys,
                                               const int32_t* out_hash_vals, int64_t hash_cap, int32_t* nbr,
                                               int64_t no_cap, int precleared, void* stream) {
  if (n_cap <= 0 || no_cap <= 0 || in_stride <= 0 || kernel_size < 1 || kernel_size > 3) return CNRMA_EINVAL;
  const int K = kernel_size * kernel_size * kernel_size;
  hipStream_t st = as_stream(stream);
  if (!precleared) {
    hipError_t e = cnrma_fill_bytes(nbr, 0xFF, (size_t)no_cap * K * sizeof(int32_t), st);
    if (e != hipSuccess) return -(int)e;
  }
  hipLaunchKernelGGL(kernel_map_strided_kernel, dim3((unsigned)ceil_div(n_cap, 256)), dim3(256), 0, st, in_coords,
                     n_cap, n_dev, in_stride, kernel_size, out_hash_keys, out_hash_vals, hash_cap, nbr);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cnrma_sparse_conv_workspace_bytes(int64_t no_cap, int Cout, int K) {
  // room for a full split over the kernel offsets of a short layer; long layers are never split
  if (K <= 1 || no_cap >= 65536) return 0;
  return (size_t)K * (size_t)no_cap * (size_t)Cout * sizeof(float);
}

#ifdef CNRMA_EXPERIMENTS
extern "C" int cnrma_debug_conv_tuning(const int* v, int n) {
  // v = {tile shape (0 128x128, 1 128x64, 2 64x64, 3 128x32, 4 64x128), splits over the kernel offsets, prefetch depth}; -1 or
  // missing = the launcher's own choice; n == 0 restores the product configuration.  Host-side global state: A/B runs only.
  ConvTune t;
  // ... {..., ablation mask, warp-specialised ring, XCD order (0 plain, 1 groups -> XCDs, 2 row tiles -> XCDs), gather-once form
  // (0 first form, 1 / 2 = second form with 1 / 2 row tiles per block), weight offsets in flight (2 / 4)}
  int* f[] = {&t.shape, &t.splits, &t.pf, &t.ablate, &t.ws, &t.xcd, &t.go, &t.nb};
  if (n < 0 || n > 8 || (n > 0 && v == nullptr)) return CNRMA_EINVAL;
  for (int i = 0; i < n; ++i) *f[i] = v[i];
  g_conv_tune = t;
  return 0;
}
#endif

extern "C" int cnrma_sparse_conv_plan(int64_t no_cap, int Cin, int Cout, int K, int mode, int slices, size_t workspace_bytes,
                                      int* out6) {
  // what launch_conv will run for these sizes: out6 = {tile rows, tile columns, splits, offsets per split, prefetch depth,
  // shape id}; mode 0 fp32 / bf16x6 weights absent = fp32 MFMA kernel, 1 = f16x3, 2 = bf16, 3 = bf16x6
  if (out6 == nullptr || Cin <= 0 || Cout <= 0 || K <= 0 || K > 27 || no_cap <= 0 || mode < 0 || mode > 3) return CNRMA_EINVAL;
  const bool six = mode != 0 && Cin % 32 == 0;
  const ConvPlan pl = plan_conv(no_cap, Cin, Cout, K, mode == 3 ? 0 : mode, six, slices > 1 ? slices : 1, workspace_bytes > 0,
                                workspace_bytes);
  out6[0] = pl.bm; out6[1] = pl.bn; out6[2] = pl.splits; out6[3] = pl.k_per_split; out6[4] = pl.pf; out6[5] = pl.shape;
  return 0;
}

extern "C" int cnrma_sparse_conv_f32(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* weight,
                                     int Cout, const float* scale, const float* shift, const float* residual, int act,
                                     float* out_feats, int64_t no_cap, const int32_t* no_dev, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return launch_conv(in_feats, Cin, nbr, K, weight, Cout, scale, shift, residual, act, out_feats, no_cap, no_dev, 1,
                     workspace, workspace_bytes, as_stream(stream));
}

extern "C" int cnrma_sparse_conv_prepare_weights(const float* weight, int K, int Cin, int Cout, void* weight_split,
                                                 void* stream) {
  if (K <= 0 || Cin <= 0 || Cout <= 0) return CNRMA_EINVAL;
  int64_t total = (int64_t)K * Cin * conv_cout_padded(Cout);
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight,
                     reinterpret_cast<__bf16*>(weight_split), K, Cin, Cout);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_bf16x6(const float* in_feats, const void* in_split, int64_t in_zero_row, int Cin,
                                        const int32_t* nbr, int K, const void* weight_split, int Cout,
                                        const float* scale, const float* shift, const float* residual, int act,
                                        float* out_feats, void* out_split, int64_t no_cap, const int32_t* no_dev,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (weight_split == nullptr || Cin % 32 != 0 || (in_feats == nullptr && in_split == nullptr)) return CNRMA_EINVAL;
  return launch_conv(in_feats, Cin, nbr, K, nullptr, Cout, scale, shift, residual, act, out_feats, no_cap, no_dev, 1,
                     workspace, workspace_bytes, as_stream(stream), weight_split, in_split, in_zero_row, out_split,
                     no_cap);
}

extern "C" int cnrma_rma_emit_features_f32(const float* feat_nhwc, const float* const* feat_nhwc_ref, int C, const void* records,
                                           int64_t n_cap, const int32_t* n_dev, const float* w_div, float* out_feat,
                                           int feat_stride, float* out_amax, void* stream) {
  if ((feat_nhwc == nullptr && feat_nhwc_ref == nullptr) || records == nullptr || out_feat == nullptr || C <= 0 || n_cap <= 0 ||
      feat_stride < C)
    return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int4* rec = reinterpret_cast<const int4*>(records);
  if (C % 256 == 0)
    hipLaunchKernelGGL((emit_features_kernel<64>), dim3((unsigned)ceil_div(n_cap * 64, 256)), dim3(256), 0, st, feat_nhwc,
                       feat_nhwc_ref, C, rec, n_cap, n_dev, w_div, out_feat, feat_stride, out_amax);
  else if (C % 32 == 0)
    hipLaunchKernelGGL((emit_features_kernel<8>), dim3((unsigned)ceil_div(n_cap * 8, 256)), dim3(256), 0, st, feat_nhwc,
                       feat_nhwc_ref, C, rec, n_cap, n_dev, w_div, out_feat, feat_stride, out_amax);
  else
    hipLaunchKernelGGL((emit_features_kernel<2>), dim3((unsigned)ceil_div(n_cap * 2, 256)), dim3(256), 0, st, feat_nhwc,
                       feat_nhwc_ref, C, rec, n_cap, n_dev, w_div, out_feat, feat_stride, out_amax);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_absmax_f32(const float* in, int64_t n_cap, const int32_t* n_dev, int C, float* out_amax,
                                void* stream) {
  if (n_cap <= 0 || C <= 0 || out_amax == nullptr) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  hipError_t e = cnrma_fill_bytes(out_amax, 0, sizeof(float) * AMAX_SLOTS * AMAX_STRIDE, st);
  if (e != hipSuccess) return -(int)e;
  int64_t blocks = ceil_div(n_cap * C / 4 + 1, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, n_cap, n_dev, C, out_amax);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cnrma_sparse_conv_weight_bytes(int K, int Cin, int Cout) {
  return (size_t)3 * K * Cin * conv_cout_padded(Cout) * sizeof(uint16_t);
}

extern "C" size_t cnrma_amax_bytes(void) { return sizeof(float) * AMAX_SLOTS * AMAX_STRIDE; }

extern "C" size_t cnrma_sparse_conv_f16_weight_bytes(int K, int Cin, int Cout) {
  return (size_t)2 * K * Cin * conv_cout_padded(Cout) * sizeof(uint16_t) + 64 + sizeof(float) * AMAX_SLOTS * AMAX_STRIDE;
}

extern "C" int cnrma_sparse_conv_prepare_weights_f16(const float* weight, int K, int Cin, int Cout, void* weight_split,
                                                     void* stream) {
  if (K <= 0 || Cin <= 0 || Cout <= 0 || weight_split == nullptr) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t total_src = (int64_t)K * Cin * Cout;
  const int64_t total = (int64_t)K * Cin * conv_cout_padded(Cout);
  uint16_t* wt = reinterpret_cast<uint16_t*>(weight_split);
  float* amax = reinterpret_cast<float*>(wt + 2 * total) + 16;       // slot scratch behind the 64-byte trailer
  hipError_t e = cnrma_fill_bytes(amax, 0, sizeof(float) * AMAX_SLOTS * AMAX_STRIDE, st);
  if (e != hipSuccess) return -(int)e;
  int64_t blocks = ceil_div(total_src / 4 + 1, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, weight, total_src, nullptr, 1, amax);
  blocks = ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, weight, wt, K, Cin, Cout, amax);
  CNRMA_LAUNCH_CHECK();
  return 0;
}


// ---- pair-list convolution: layers whose kernel map is nearly empty -----------------------------------------------------
// The output-stationary kernel multiplies a whole 64/128-row tile by W[k] as soon as ONE row of the tile has a neighbour
// at offset k.  In the stem (stride 2 on a point sample: 1.3-1.5 pairs per output row, 5 % of the 27 x N_out table) every
// tile has every offset, i.e. 95 % of the matrix work multiplies zero rows.  Here the table is regrouped instead: the
// valid (output o, offset k) entries become rows of a pair list sorted by k (runs padded to 128 rows with -1), the same
// MFMA kernel runs over that list as a K = 1 convolution whose weight slice is chosen per 128-row run (tile_tap), and a
// second kernel adds every output row's pair products in ascending k -- a fixed order, so results are deterministic --
// and applies the fused epilogue.  Slots inside a run are handed out by atomics: the layout of the list varies from run to
// run, the values do not (a GEMM row does not depend on its neighbours).
constexpr int PAIR_HDR = 128;      // ints: cnt[32] | fill[32] | base[33] | n_vrows | overflow flag

constexpr int PAIR_EPT = 32;                    // table entries per thread and chunk (chunk = 8192 entries per block)

// k of flat table index t (t < 2^31: the pair slots are int32)
__device__ __forceinline__ int tap_of(unsigned t, int K) { return K == 27 ? (int)(t % 27u) : (int)(t % (unsigned)K); }

__global__ __launch_bounds__(256) void pairs_count_kernel(const int32_t* __restrict__ nbr, int64_t no_cap,
                                                          const int32_t* __restrict__ no_dev, int K, int32_t* __restrict__ hdr) {
  __shared__ int hist[32];
  if (threadIdx.x < 32) hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned total = (unsigned)(live_rows(no_cap, no_dev) * K);
  const unsigned base = blockIdx.x * (256u * PAIR_EPT);
#pragma unroll 8
  for (int j = 0; j < PAIR_EPT; ++j) {
    const unsigned t = base + j * 256u + threadIdx.x;
    if (t < total && nbr[t] >= 0) atomicAdd(&hist[tap_of(t, K)], 1);
  }
  __syncthreads();
  if (threadIdx.x < K && hist[threadIdx.x]) atomicAdd(&hdr[threadIdx.x], hist[threadIdx.x]);
}

// one block: run bases (multiples of 128), the tap of every 128-row run, the padded row count; padding rows get -1
__global__ __launch_bounds__(256) void pairs_plan_kernel(int K, int32_t* __restrict__ hdr, int32_t* __restrict__ tile_tap,
                                                         int32_t* __restrict__ pair_src, int64_t pair_cap) {
  __shared__ int base[33];
  if (threadIdx.x == 0) {
    int b = 0;
    for (int k = 0; k < K; ++k) {
      base[k] = b;
      b += (hdr[k] + 127) & ~127;
    }
    base[K] = b;
    for (int k = 0; k <= K; ++k) hdr[64 + k] = base[k];
    hdr[64 + 33] = (int64_t)b <= pair_cap ? b : (int)pair_cap;       // b <= pair_cap by construction of the capacity
  }
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    const int lo = base[k], hi = base[k + 1], live = lo + hdr[k];
    // slots at or beyond pair_cap do not exist (a caller-chosen capacity below the provable bound): never written, and
    // the fill / reduce kernels treat such entries as absent
    for (int t = lo / 128 + threadIdx.x; t < hi / 128 && (int64_t)t < pair_cap / 128; t += 256) tile_tap[t] = k;
    for (int r = live + threadIdx.x; r < hi && (int64_t)r < pair_cap; r += 256) pair_src[r] = -1;
  }
}

__global__ __launch_bounds__(256) void pairs_fill_kernel(const int32_t* __restrict__ nbr, int64_t no_cap,
                                                         const int32_t* __restrict__ no_dev, int K, int32_t* __restrict__ hdr,
                                                         int32_t* __restrict__ pair_src, int32_t* __restrict__ pos,
                                                         int64_t pair_cap) {
  // slot = run base + one global reservation per offset and 8192-entry chunk + rank inside the chunk (LDS atomics):
  // per-entry global atomics on 27 words serialise at the memory side (3 ms for 660 k entries)
  __shared__ int hist[32], gbase[32];
  if (threadIdx.x < 32) hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned total = (unsigned)(live_rows(no_cap, no_dev) * K);
  const unsigned base = blockIdx.x * (256u * PAIR_EPT);
  int32_t src[PAIR_EPT];
#pragma unroll
  for (int j = 0; j < PAIR_EPT; ++j) {
    const unsigned t = base + j * 256u + threadIdx.x;
    src[j] = t < total ? nbr[t] : -1;
    if (src[j] >= 0) atomicAdd(&hist[tap_of(t, K)], 1);
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int c = hist[threadIdx.x];
    gbase[threadIdx.x] = c ? hdr[64 + threadIdx.x] + atomicAdd(&hdr[32 + threadIdx.x], c) : 0;
    hist[threadIdx.x] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PAIR_EPT; ++j) {
    const unsigned t = base + j * 256u + threadIdx.x;
    if (t < total) {
      int32_t slot = -1;
      if (src[j] >= 0) {
        const int k = tap_of(t, K);
        slot = gbase[k] + atomicAdd(&hist[k], 1);
        if ((int64_t)slot < pair_cap) pair_src[slot] = src[j];
        else { slot = -1; hdr[64 + 34] = 1; }                  // capacity below the bound: entry dropped, flagged
      }
      pos[t] = slot;
    }
  }
}

// out[o] = epilogue(sum over k ascending of prod[pos[o][k]]); 4 columns per thread
template <int KT>
__global__ __launch_bounds__(256) void pairs_reduce_kernel(ConvArgs p, const int32_t* __restrict__ pos,
                                                           const float* __restrict__ prod) {
  const int64_t n_live = live_rows(p.no_cap, p.no_dev);
  const int C4 = p.Cout >> 2;
  const int64_t total = n_live * C4;
  const int K = KT > 0 ? KT : p.K;
  float mx = 0.0f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / C4;
    const int col = (int)(t % C4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t* po = pos + o * K;
    if constexpr (KT > 0) {
      int32_t r[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) r[k] = po[k];            // independent loads first, then the (few) product rows
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (r[k] >= 0) {
          const float4 q = *reinterpret_cast<const float4*>(prod + (int64_t)r[k] * p.Cout + col);
          s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
    } else {
      for (int k = 0; k < K; ++k) {
        const int32_t r = po[k];
        if (r >= 0) {
          const float4 q = *reinterpret_cast<const float4*>(prod + (int64_t)r * p.Cout + col);
          s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
      }
    }
    float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j];
      if (p.scale) x = x * p.scale[col + j];
      if (p.shift) x = x + p.shift[col + j];
      if (p.residual) x = x + p.residual[o * p.Cout + col + j];
      v[j] = apply_act(x, p.act);
      mx = fmaxf(mx, fabsf(v[j]));
    }
    *reinterpret_cast<float4*>(p.out + o * p.Cout + col) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (p.out_amax != nullptr) {
    __shared__ float sh4[4];
    block_amax_publish(p.out_amax, mx, sh4);
  }
}

static size_t pairs_align(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t cnrma_sparse_conv_pairs_workspace_bytes(int64_t no_cap, int K, int Cout, int64_t pair_cap) {
  if (no_cap <= 0 || K <= 0 || Cout <= 0 || pair_cap <= 0) return 0;
  return pairs_align(PAIR_HDR * 4) + pairs_align((size_t)(pair_cap / 128 + 1) * 4) + pairs_align((size_t)pair_cap * 4) +
         pairs_align((size_t)no_cap * K * 4) + pairs_align((size_t)pair_cap * Cout * 4);
}

extern "C" int cnrma_sparse_conv_pairs_f16x3(const float* in_feats, const float* in_amax, int Cin, const int32_t* nbr, int K,
                                             const void* weight_split, int Cout, const float* scale, const float* shift,
                                             const float* residual, int act, float* out_feats, float* out_amax,
                                             int64_t no_cap, const int32_t* no_dev, int64_t pair_cap, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  if (weight_split == nullptr || Cin % 32 != 0 || Cout % 4 != 0 || in_feats == nullptr || in_amax == nullptr || nbr == nullptr ||
      K <= 1 || K > 27 || no_cap <= 0 || pair_cap <= 0 || pair_cap % 128 != 0 || pair_cap > 0x7fffff00LL)
    return CNRMA_EINVAL;
  if (workspace == nullptr || workspace_bytes < cnrma_sparse_conv_pairs_workspace_bytes(no_cap, K, Cout, pair_cap))
    return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  int32_t* hdr = reinterpret_cast<int32_t*>(w);        w += pairs_align(PAIR_HDR * 4);
  int32_t* tile_tap = reinterpret_cast<int32_t*>(w);   w += pairs_align((size_t)(pair_cap / 128 + 1) * 4);
  int32_t* pair_src = reinterpret_cast<int32_t*>(w);   w += pairs_align((size_t)pair_cap * 4);
  int32_t* pos = reinterpret_cast<int32_t*>(w);        w += pairs_align((size_t)no_cap * K * 4);
  float* prod = reinterpret_cast<float*>(w);
  const hipError_t fe = cnrma_fill_bytes(hdr, 0, PAIR_HDR * 4, st);
  if (fe != hipSuccess) return -(int)fe;
  if (no_cap * K >= 0x7fffffffLL) return CNRMA_EINVAL;
  const int64_t blocks = ceil_div(no_cap * K, 256 * PAIR_EPT);
  hipLaunchKernelGGL(pairs_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nbr, no_cap, no_dev, K, hdr);
  hipLaunchKernelGGL(pairs_plan_kernel, dim3(1), dim3(256), 0, st, K, hdr, tile_tap, pair_src, pair_cap);
  hipLaunchKernelGGL(pairs_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nbr, no_cap, no_dev, K, hdr, pair_src, pos,
                     pair_cap);
  CNRMA_LAUNCH_CHECK();
  const int rc = launch_conv(in_feats, Cin, pair_src, 1, nullptr, Cout, nullptr, nullptr, nullptr, 0, prod, pair_cap, hdr + 64 + 33,
                             1, nullptr, 0, st, weight_split, nullptr, 0, nullptr, 0, 1, in_amax, nullptr, tile_tap, K);
  if (rc != 0) return rc;
  ConvArgs p{};
  p.K = K; p.Cout = Cout; p.scale = scale; p.shift = shift; p.residual = residual; p.act = act; p.out = out_feats;
  p.no_cap = no_cap; p.no_dev = no_dev; p.out_amax = out_amax;
  int64_t rb = ceil_div(no_cap * (Cout / 4), 256);
  if (rb > 16384) rb = 16384;
  if (K == 27) hipLaunchKernelGGL(pairs_reduce_kernel<27>, dim3((unsigned)rb), dim3(256), 0, st, p, pos, prod);
  else hipLaunchKernelGGL(pairs_reduce_kernel<0>, dim3((unsigned)rb), dim3(256), 0, st, p, pos, prod);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// the same regrouping in exact fp32: the fp32 MFMA stage kernel over the pair runs (the north-star stem at CONV_PRECISION = "f32"
// ran the tile kernel over a 5 %-full table: 3.1 ms of the 11.1 ms of convolutions per scene)
extern "C" int cnrma_sparse_conv_pairs_f32(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* weight, int Cout,
                                           const float* scale, const float* shift, const float* residual, int act, float* out_feats,
                                           int64_t no_cap, const int32_t* no_dev, int64_t pair_cap, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (weight == nullptr || Cin % 32 != 0 || Cout % 4 != 0 || in_feats == nullptr || nbr == nullptr || K <= 1 || K > 27 || no_cap <= 0 ||
      pair_cap <= 0 || pair_cap % 128 != 0 || pair_cap > 0x7fffff00LL)
    return CNRMA_EINVAL;
  if (workspace == nullptr || workspace_bytes < cnrma_sparse_conv_pairs_workspace_bytes(no_cap, K, Cout, pair_cap))
    return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  int32_t* hdr = reinterpret_cast<int32_t*>(w);        w += pairs_align(PAIR_HDR * 4);
  int32_t* tile_tap = reinterpret_cast<int32_t*>(w);   w += pairs_align((size_t)(pair_cap / 128 + 1) * 4);
  int32_t* pair_src = reinterpret_cast<int32_t*>(w);   w += pairs_align((size_t)pair_cap * 4);
  int32_t* pos = reinterpret_cast<int32_t*>(w);        w += pairs_align((size_t)no_cap * K * 4);
  float* prod = reinterpret_cast<float*>(w);
  const hipError_t fe = cnrma_fill_bytes(hdr, 0, PAIR_HDR * 4, st);
  if (fe != hipSuccess) return -(int)fe;
  if (no_cap * K >= 0x7fffffffLL) return CNRMA_EINVAL;
  const int64_t blocks = ceil_div(no_cap * K, 256 * PAIR_EPT);
  hipLaunchKernelGGL(pairs_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nbr, no_cap, no_dev, K, hdr);
  hipLaunchKernelGGL(pairs_plan_kernel, dim3(1), dim3(256), 0, st, K, hdr, tile_tap, pair_src, pair_cap);
  hipLaunchKernelGGL(pairs_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, nbr, no_cap, no_dev, K, hdr, pair_src, pos,
                     pair_cap);
  CNRMA_LAUNCH_CHECK();
  const int rc = launch_conv(in_feats, Cin, pair_src, 1, weight, Cout, nullptr, nullptr, nullptr, 0, prod, pair_cap, hdr + 64 + 33,
                             1, nullptr, 0, st, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, tile_tap, K);
  if (rc != 0) return rc;
  ConvArgs p{};
  p.K = K; p.Cout = Cout; p.scale = scale; p.shift = shift; p.residual = residual; p.act = act; p.out = out_feats;
  p.no_cap = no_cap; p.no_dev = no_dev; p.out_amax = nullptr;
  int64_t rb = ceil_div(no_cap * (Cout / 4), 256);
  if (rb > 16384) rb = 16384;
  if (K == 27) hipLaunchKernelGGL(pairs_reduce_kernel<27>, dim3((unsigned)rb), dim3(256), 0, st, p, pos, prod);
  else hipLaunchKernelGGL(pairs_reduce_kernel<0>, dim3((unsigned)rb), dim3(256), 0, st, p, pos, prod);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// ---- second form of the gather-once kernel: instantiation table, work order, splits ---------------------------------------------
constexpr int GO_FORM_DEFAULT = 1;      // the second form everywhere (scripts/go_forms.py, profiles/r05_go_forms2_*.log: 0.93x summed over
                                        // the layer classes of an S scene, never slower than the first form in its best work order)
static int go_form_default(int64_t no_cap, int Cin, int Cout) { (void)no_cap; (void)Cin; (void)Cout; return GO_FORM_DEFAULT; }

template <int WAVES_N, int KS, bool HAS_RES, int NB, bool STAMP = false, bool BF1 = false, bool APF = false>
static int launch_go2_one(unsigned blocks, const ConvArgs& p, const GoArgs& g, const uint16_t* wfrag, const Go2Map& mp, hipStream_t st) {
  hipLaunchKernelGGL((sparse_conv_go2_kernel<WAVES_N, KS, HAS_RES, NB, STAMP, BF1, APF>), dim3(blocks), dim3(256), GO2_LDS, st, p, g, wfrag, mp);
  return 0;
}
// fragment look-ahead instantiation (APF): MEASURED AND REJECTED (round 6, scripts/go_forms.py AB=f2,f2a,
// profiles/r06_go_apf_{S,NS}.log): equal within 1 % on every short layer class (11 k rows 42.8 vs 43.0 us, 2.4 k 37.3 vs 38.3, 541
// 36.4 vs 37.0), 2-4 % slower on the 200-500 k-row layers (three blocks per CU).  The blocks of a short layer all start together,
// so their offset phases coincide and the ~2.8 waves per SIMD saturate the matrix pipe INSIDE that phase (2.8 x 384 MFMA cycles per
// offset step = the 0.6-0.74 us per step the stamps show): the fragment-read latency was already covered by the other waves.
// Experiments library only (conv_tuning(nb=12)).
constexpr unsigned GO_APF_MAX_BLOCKS = 0;
static_assert(GO2_LDS <= 48 * 1024 && 4 * GO2_LDS <= 160 * 1024, "four blocks per CU without raising the dynamic LDS limit");

// what the gather-once launcher runs for a layer: a pure function of the CAPACITY of the output, the widths and the workspace
// (a captured launch sequence replays the same kernels; cnrma_sparse_conv_go_plan exposes it to the tests)
struct Go2Plan { int form, bn, ks, splits, slices_per_split, mode, nb, apf; int64_t tiles; unsigned blocks; Go2Map mp; };
static Go2Plan go2_plan(int64_t no_cap, int Cin, int Cout, bool has_ws, size_t workspace_bytes) {
  const ConvTune tune = CNRMA_CONV_TUNE;
  Go2Plan pl{};
  pl.form = tune.go >= 0 ? tune.go : go_form_default(no_cap, Cin, Cout);
  pl.bn = Cout >= 128 ? 128 : 64;
  pl.ks = pl.bn == 128 ? 1 : 2;
  pl.tiles = ceil_div(no_cap, GO_BM);
  const int ns = Cin / BK;
  Go2Map& mp = pl.mp;
  mp.tiles = (int)pl.tiles;
  mp.ncol = (int)ceil_div(Cout, pl.bn);
  // short layers: split over the 32-channel slices (every block runs all 27 offsets of its slices); conv_reduce_kernel adds the
  // partial slabs in slab order
  int splits = 1;
  const int64_t blocks0 = pl.tiles * mp.ncol;
  if (has_ws && ns > 1 && (blocks0 < 384 || tune.splits > 0)) {
    splits = tune.splits > 0 ? tune.splits : (int)ceil_div(768, blocks0);
    if (splits > ns) splits = ns;
    const size_t per = (size_t)no_cap * Cout * sizeof(float);
    if (per > 0 && (size_t)splits * per > workspace_bytes) splits = (int)(workspace_bytes / per);
    if (splits < 2) splits = 1;
  }
  pl.slices_per_split = (int)ceil_div(ns, splits);
  pl.splits = (int)ceil_div(ns, pl.slices_per_split);
  mp.ng = mp.ncol * pl.splits;
  mp.per = (int)ceil_div(pl.tiles, 8);
  // work order, measured per layer class (profiles/r05_go_forms2_*.log): a handful of tiles under a 7-28 MB weight tensor (the
  // 541-966-row level): groups -> XCDs (35 vs 69 us for tiles -> XCDs); the 200-500 k-row layers: plain order (188 vs 205 us);
  // everything between: each XCD one contiguous eighth of the tiles (38-105 us classes: 3-10 % under the plain order)
  mp.mode = tune.xcd >= 0 ? tune.xcd : (pl.tiles < 32 && mp.ng >= 8 ? 1 : (pl.tiles >= 2048 ? 0 : 2));
  pl.mode = mp.mode;
  if (mp.mode == 1) pl.blocks = 8u * (unsigned)mp.tiles * (unsigned)ceil_div(mp.ng, 8);
  else if (mp.mode == 2) pl.blocks = 8u * (unsigned)mp.per * (unsigned)mp.ng;
  else pl.blocks = (unsigned)mp.tiles * (unsigned)mp.ng;
  pl.nb = tune.nb == 4 ? 4 : 2;
  // tune.nb 12 / 10 (experiments library): fragment look-ahead forced on / off
#ifdef CNRMA_EXPERIMENTS
  pl.apf = tune.nb == 12 ? 1 : (tune.nb == 10 ? 0 : (pl.blocks <= GO_APF_MAX_BLOCKS ? 1 : 0));
#else
  pl.apf = 0;
#endif
  return pl;
}

static int launch_go2(const Go2Plan& pl, ConvArgs p, GoArgs g, const uint16_t* wfrag, bool residual, void* stamps, hipStream_t st,
                      bool bf1 = false) {
  const Go2Map& mp = pl.mp;
  g.slices_per_split = pl.slices_per_split;
  p.splits = pl.splits;
  g.counters = reinterpret_cast<unsigned*>(stamps);          // diagnostic build: the stamp buffer (16 x 8 bytes per block)
  if ((p.ablate & 64) && stamps == nullptr) return CNRMA_EINVAL;
  const unsigned blocks = pl.blocks;
  const bool has_res = residual && pl.splits == 1;
  int rc = CNRMA_EINVAL;
  if (bf1) {                                                 // training: no fused residual (conv_reduce_kernel adds one after a split)
    if (has_res || stamps != nullptr) return CNRMA_EINVAL;
    rc = pl.bn == 128 ? launch_go2_one<4, 1, false, 2, false, true>(blocks, p, g, wfrag, mp, st)
                      : launch_go2_one<2, 2, false, 2, false, true>(blocks, p, g, wfrag, mp, st);
    if (rc != 0) return rc;
    if (pl.splits > 1) {
      int64_t rb = ceil_div(p.no_cap * p.Cout / 4 + 1, 256);
      if (rb > 4096) rb = 4096;
      hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, p);
    }
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
#ifdef CNRMA_EXPERIMENTS
  if (pl.apf && stamps == nullptr) {                          // fragment look-ahead (measured: no gain; see GO_APF_MAX_BLOCKS)
#define CNRMA_GO2A(WN, KS_)                                                                                             \
  rc = has_res ? launch_go2_one<WN, KS_, true, 2, false, false, true>(blocks, p, g, wfrag, mp, st)                      \
               : launch_go2_one<WN, KS_, false, 2, false, false, true>(blocks, p, g, wfrag, mp, st)
    if (pl.bn == 128) CNRMA_GO2A(4, 1);
    else CNRMA_GO2A(2, 2);
#undef CNRMA_GO2A
  } else
#endif
  {
#ifdef CNRMA_EXPERIMENTS            // + the s_memtime-stamped instantiation and four weight offsets in flight (measured 0-5 % slower)
#define CNRMA_GO2(WN, KS_, NB_)                                                                              \
  rc = stamps != nullptr && !has_res ? launch_go2_one<WN, KS_, false, NB_, true>(blocks, p, g, wfrag, mp, st) \
       : has_res ? launch_go2_one<WN, KS_, true, NB_>(blocks, p, g, wfrag, mp, st)                           \
                 : launch_go2_one<WN, KS_, false, NB_>(blocks, p, g, wfrag, mp, st)
  if (pl.bn == 128) {
    if (pl.nb == 2) CNRMA_GO2(4, 1, 2);
    else CNRMA_GO2(4, 1, 4);
  } else {
    if (pl.nb == 2) CNRMA_GO2(2, 2, 2);
    else CNRMA_GO2(2, 2, 4);
  }
#else
#define CNRMA_GO2(WN, KS_, NB_)                                                                              \
  rc = has_res ? launch_go2_one<WN, KS_, true, NB_>(blocks, p, g, wfrag, mp, st)                             \
               : launch_go2_one<WN, KS_, false, NB_>(blocks, p, g, wfrag, mp, st)
  if (stamps != nullptr) return CNRMA_EINVAL;
  if (pl.bn == 128) CNRMA_GO2(4, 1, 2);
  else CNRMA_GO2(2, 2, 2);
#endif
#undef CNRMA_GO2
  }
  if (rc != 0) return rc;
  if (pl.splits > 1) {
    int64_t rb = ceil_div(p.no_cap * p.Cout / 4 + 1, 256);
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, p);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

#ifdef CNRMA_EXPERIMENTS
#include "sparse_exp_go3_launch.inc"
#endif

extern "C" int cnrma_sparse_conv_go_plan(int64_t no_cap, int Cin, int Cout, size_t workspace_bytes, int has_residual, int* out8) {
  // what cnrma_sparse_conv_go_f16x3 launches for these sizes: out8 = {form (0 first, 1 second), tile columns (64: the four waves
  // are 2 column tiles x 2 offset halves, 128: 4 column tiles), splits over channel slices, slices per split, work order (0
  // plain, 1 groups -> XCDs, 2 tiles -> XCDs), residual fused in the kernel (0: in conv_reduce_kernel), blocks, weight offsets in flight}
  if (out8 == nullptr || no_cap <= 0 || Cin <= 0 || Cin % BK != 0 || Cout < 64) return CNRMA_EINVAL;
  const Go2Plan pl = go2_plan(no_cap, Cin, Cout, workspace_bytes > 0, workspace_bytes);
  out8[0] = pl.form >= 2 ? 2 : (pl.form >= 1 ? 1 : 0); out8[1] = pl.bn; out8[2] = pl.splits; out8[3] = pl.slices_per_split;
  out8[4] = pl.form >= 1 ? pl.mode : 0; out8[5] = has_residual && pl.splits == 1; out8[6] = (int)pl.blocks; out8[7] = pl.nb + (pl.apf ? 10 : 0);      // + 10: fragment look-ahead instantiation
  return 0;
}

static size_t go_align(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- exact-fp32 gather-once convolution: weight image, launcher -------------------------------------------------------------
extern "C" size_t cnrma_sparse_conv_f32_frag_weight_bytes(int K, int Cin, int Cout) {
  return (size_t)K * Cin * conv_cout_padded(Cout) * sizeof(float);
}

extern "C" int cnrma_sparse_conv_prepare_weights_f32_frag(const float* weight, int K, int Cin, int Cout, void* weight_frag,
                                                          void* stream) {
  if (weight == nullptr || weight_frag == nullptr || K <= 0 || Cin <= 0 || Cin % BK != 0 || Cout <= 0) return CNRMA_EINVAL;
  const int64_t total = (int64_t)K * Cin * conv_cout_padded(Cout);
  int64_t blocks = ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_f32_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight,
                     reinterpret_cast<float*>(weight_frag), K, Cin, Cout);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_go_f32(const float* in_feats, int Cin, const void* tile_union, const void* weight_frag, int Cout,
                                        const float* scale, const float* shift, const float* residual, int act, float* out_feats,
                                        int64_t no_cap, const int32_t* no_dev, void* workspace, size_t workspace_bytes, void* stream) {
  if (in_feats == nullptr || tile_union == nullptr || weight_frag == nullptr || out_feats == nullptr || Cin <= 0 || Cin % BK != 0 ||
      Cout < 64 || no_cap <= 0)
    return CNRMA_EINVAL;
  const int K = 27;
  hipStream_t st = as_stream(stream);
  const size_t tiles = (size_t)ceil_div(no_cap, GO_BM);
  const char* w = reinterpret_cast<const char*>(tile_union);
  GoArgs g;
  g.hdr = reinterpret_cast<const int32_t*>(w);      w += go_align(tiles * GO_HDR * 4);
  g.rows = reinterpret_cast<const int32_t*>(w);     w += go_align(tiles * GO_ROWS * 4);
  g.lidx = reinterpret_cast<const uint16_t*>(w);
  g.counters = nullptr;
  ConvArgs p{in_feats, Cin, nullptr, K, nullptr, Cout, scale, shift, residual, act, out_feats, no_cap, no_dev, 1, 1, K,
             reinterpret_cast<float*>(workspace), nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
  const Go2Plan pl = go2_plan(no_cap, Cin, Cout, workspace != nullptr, workspace_bytes);
  g.slices_per_split = pl.slices_per_split;
  p.splits = pl.splits;
  const bool has_res = residual != nullptr && pl.splits == 1;
  const float* wf = reinterpret_cast<const float*>(weight_frag);
  if (pl.bn == 128) {
    if (has_res) hipLaunchKernelGGL((sparse_conv_gof_kernel<4, 1, true>), dim3(pl.blocks), dim3(256), GOF_LDS, st, p, g, wf, pl.mp);
    else hipLaunchKernelGGL((sparse_conv_gof_kernel<4, 1, false>), dim3(pl.blocks), dim3(256), GOF_LDS, st, p, g, wf, pl.mp);
  } else {
    if (has_res) hipLaunchKernelGGL((sparse_conv_gof_kernel<2, 2, true>), dim3(pl.blocks), dim3(256), GOF_LDS, st, p, g, wf, pl.mp);
    else hipLaunchKernelGGL((sparse_conv_gof_kernel<2, 2, false>), dim3(pl.blocks), dim3(256), GOF_LDS, st, p, g, wf, pl.mp);
  }
  if (pl.splits > 1) {
    int64_t rb = ceil_div(no_cap * Cout / 4 + 1, 256);
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, p);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// ---- gather-once convolution: tile unions, fragment-order weights, launcher -----------------------------------------------

extern "C" size_t cnrma_sparse_tile_union_bytes(int64_t no_cap) {
  if (no_cap <= 0) return 0;
  const size_t tiles = (size_t)ceil_div(no_cap, GO_BM);
  return go_align(tiles * GO_HDR * 4) + go_align(tiles * GO_ROWS * 4) + go_align(tiles * GO_BM * 27 * 2);
}

extern "C" int cnrma_sparse_tile_union_build(const int32_t* nbr, int64_t no_cap, const int32_t* no_dev, int K, void* tile_union,
                                             void* stream) {
  if (nbr == nullptr || tile_union == nullptr || no_cap <= 0 || K != 27) return CNRMA_EINVAL;
  const size_t tiles = (size_t)ceil_div(no_cap, GO_BM);
  char* w = reinterpret_cast<char*>(tile_union);
  int32_t* hdr = reinterpret_cast<int32_t*>(w);      w += go_align(tiles * GO_HDR * 4);
  int32_t* rows = reinterpret_cast<int32_t*>(w);     w += go_align(tiles * GO_ROWS * 4);
  uint16_t* lidx = reinterpret_cast<uint16_t*>(w);
  hipLaunchKernelGGL(tile_union_kernel, dim3((unsigned)ceil_div((int64_t)tiles, 4)), dim3(256), 0, as_stream(stream), nbr, no_cap,
                     no_dev, K, hdr, rows, lidx, CNRMA_CONV_TUNE.ablate);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_wgrad_go_bf16(const float* in_feats, int Cin, const void* tile_union, const float* grad_out,
                                               int Cout, int64_t no_cap, const int32_t* no_dev, int parts, float* slabs,
                                               void* stream) {
  if (in_feats == nullptr || tile_union == nullptr || grad_out == nullptr || slabs == nullptr || Cin <= 0 || Cout <= 0 ||
      (Cin & 3) || (Cout & 3) || no_cap <= 0 || parts <= 0)
    return CNRMA_EINVAL;
  const size_t tiles = (size_t)ceil_div(no_cap, GO_BM);
  const char* w = reinterpret_cast<const char*>(tile_union);
  const int32_t* hdr = reinterpret_cast<const int32_t*>(w);      w += go_align(tiles * GO_HDR * 4);
  const int32_t* rows = reinterpret_cast<const int32_t*>(w);     w += go_align(tiles * GO_ROWS * 4);
  const uint16_t* lidx = reinterpret_cast<const uint16_t*>(w);
  WgoMap map;
  map.parts = parts;
  map.tiles_per_part = (int)ceil_div((int64_t)tiles, parts);
  map.n_ci = (int)ceil_div(Cin, 64);
  map.n_co = (int)ceil_div(Cout, 64);
  map.by_part = parts >= 16;
  map.ablate = CNRMA_CONV_TUNE.ablate;
  const int64_t per = (int64_t)WGO_KG * map.n_ci * map.n_co;
  const int64_t blocks = (map.by_part ? ceil_div(parts, 8) * 8 : (int64_t)parts) * per;
  if (blocks > 0x7fffffffLL) return CNRMA_EINVAL;
  hipLaunchKernelGGL(conv_wgrad_go_kernel, dim3((unsigned)blocks), dim3(64 * (WGO_CW + WGO_PW)), 0, as_stream(stream), in_feats, Cin, grad_out, Cout,
                     no_cap, no_dev, hdr, rows, lidx, slabs, map);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_prepare_weights_f16_frag(const float* weight, int K, int Cin, int Cout, void* weight_frag,
                                                          void* stream) {
  if (K <= 0 || Cin <= 0 || Cin % BK != 0 || Cout <= 0 || weight_frag == nullptr) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t total_src = (int64_t)K * Cin * Cout;
  const int64_t total = (int64_t)K * Cin * conv_cout_padded(Cout);
  uint16_t* wt = reinterpret_cast<uint16_t*>(weight_frag);
  float* amax = reinterpret_cast<float*>(wt + 2 * total) + 16;       // slot scratch behind the 64-byte trailer
  hipError_t e = cnrma_fill_bytes(amax, 0, sizeof(float) * AMAX_SLOTS * AMAX_STRIDE, st);
  if (e != hipSuccess) return -(int)e;
  int64_t blocks = ceil_div(total_src / 4 + 1, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, weight, total_src, nullptr, 1, amax);
  blocks = ceil_div(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_f16_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, st, weight, wt, K, Cin, Cout, amax);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// W fp32 [K][Cin][Cout] -> bf16 fragment-order image [k][slice][column tile of 32][k-step (2)][lane (64)][8] of
//   transpose == 0:  W itself;   transpose == 1: the data gradient's weights W'[k] = W[flip ? K - 1 - k : k]^T ([Cout] -> [Cin])
// (lane = 32 * (kk / 8 % 2) + column % 32, kk = channel inside the slice = 16 * k-step + 8 * (lane / 32) + j: the B operand
// registers of v_mfma_f32_32x32x16_bf16 -- one contiguous 1-KB load per wave and k-step)
__global__ __launch_bounds__(256) void prep_weights_bf16_frag_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int K,
                                                                     int Cin, int Cout, int transpose, int flip) {
  const int Ci = transpose ? Cout : Cin, Co = transpose ? Cin : Cout;       // the image's input / output channels
  const int Cp = conv_cout_padded(Co);
  const int64_t total = (int64_t)K * Ci * Cp;
  const int ns = Ci / BK, nt = Cp / 32;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(t & 7), lane = (int)((t >> 3) & 63), ks = (int)((t >> 9) & 1);
    int64_t q = t >> 10;
    const int tile = (int)(q % nt); q /= nt;
    const int slice = (int)(q % ns);
    const int k = (int)(q / ns);
    const int cin = slice * BK + ks * 16 + (lane >> 5) * 8 + j, co = tile * 32 + (lane & 31);
    const int ksrc = transpose && flip ? K - 1 - k : k;
    float v = 0.0f;
    if (co < Co) v = transpose ? w[((int64_t)ksrc * Cin + co) * Cout + cin] : w[((int64_t)ksrc * Cin + cin) * Cout + co];
    wt[t] = (__bf16)v;
  }
}

// both images of a weight tensor in ONE launch (training: the forward's and the mirrored-transposed one of its data gradient)
__global__ __launch_bounds__(256) void prep_weights_bf16_frag_pair_kernel(const float* __restrict__ w, __bf16* __restrict__ wf,
                                                                          __bf16* __restrict__ wtr, int K, int Cin, int Cout, int flip) {
  const int64_t total_f = (int64_t)K * Cin * conv_cout_padded(Cout), total_t = (int64_t)K * Cout * conv_cout_padded(Cin);
  for (int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < total_f + total_t; t0 += (int64_t)gridDim.x * blockDim.x) {
    const bool tr = t0 >= total_f;
    const int64_t t = tr ? t0 - total_f : t0;
    const int Ci = tr ? Cout : Cin, Co = tr ? Cin : Cout;
    const int ns = Ci / BK, nt = conv_cout_padded(Co) / 32;
    const int j = (int)(t & 7), lane = (int)((t >> 3) & 63), ks = (int)((t >> 9) & 1);
    int64_t q = t >> 10;
    const int tile = (int)(q % nt); q /= nt;
    const int slice = (int)(q % ns);
    const int k = (int)(q / ns);
    const int cin = slice * BK + ks * 16 + (lane >> 5) * 8 + j, co = tile * 32 + (lane & 31);
    const int ksrc = tr && flip ? K - 1 - k : k;
    float v = 0.0f;
    if (co < Co) v = tr ? w[((int64_t)ksrc * Cin + co) * Cout + cin] : w[((int64_t)ksrc * Cin + cin) * Cout + co];
    (tr ? wtr : wf)[t] = (__bf16)v;
  }
}

extern "C" int cnrma_sparse_conv_prepare_weights_bf16_frag_pair(const float* weight, int K, int Cin, int Cout, int flip,
                                                                void* frag_forward, void* frag_transposed, void* stream) {
  if (K <= 0 || Cin <= 0 || Cout <= 0 || Cin % BK != 0 || Cout % BK != 0 || weight == nullptr || frag_forward == nullptr ||
      frag_transposed == nullptr)
    return CNRMA_EINVAL;
  int64_t blocks = ceil_div((int64_t)K * Cin * conv_cout_padded(Cout) + (int64_t)K * Cout * conv_cout_padded(Cin), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(prep_weights_bf16_frag_pair_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight,
                     reinterpret_cast<__bf16*>(frag_forward), reinterpret_cast<__bf16*>(frag_transposed), K, Cin, Cout, flip ? 1 : 0);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cnrma_sparse_conv_bf16_frag_weight_bytes(int K, int Cin, int Cout) {
  return (size_t)K * Cin * conv_cout_padded(Cout) * 2;
}

extern "C" int cnrma_sparse_conv_prepare_weights_bf16_frag(const float* weight, int K, int Cin, int Cout, int transpose, int flip,
                                                           void* weight_frag, void* stream) {
  if (K <= 0 || Cin <= 0 || Cout <= 0 || (transpose ? Cout : Cin) % BK != 0 || weight == nullptr || weight_frag == nullptr)
    return CNRMA_EINVAL;
  const int Ci = transpose ? Cout : Cin, Co = transpose ? Cin : Cout;
  int64_t blocks = ceil_div((int64_t)K * Ci * conv_cout_padded(Co), 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_bf16_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight,
                     reinterpret_cast<__bf16*>(weight_frag), K, Cin, Cout, transpose ? 1 : 0, flip ? 1 : 0);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_go_bf16(const float* in_feats, int Cin, const void* tile_union, const void* weight_frag, int Cout,
                                         float* out_feats, int64_t no_cap, const int32_t* no_dev, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (in_feats == nullptr || tile_union == nullptr || weight_frag == nullptr || out_feats == nullptr || Cin <= 0 || Cin % BK != 0 ||
      Cout < 64 || no_cap <= 0)
    return CNRMA_EINVAL;
  const int K = 27;
  const size_t tiles = (size_t)ceil_div(no_cap, GO_BM);
  const char* w = reinterpret_cast<const char*>(tile_union);
  GoArgs g;
  g.hdr = reinterpret_cast<const int32_t*>(w);      w += go_align(tiles * GO_HDR * 4);
  g.rows = reinterpret_cast<const int32_t*>(w);     w += go_align(tiles * GO_ROWS * 4);
  g.lidx = reinterpret_cast<const uint16_t*>(w);
  ConvArgs p{in_feats, Cin, nullptr, K, nullptr, Cout, nullptr, nullptr, nullptr, 0, out_feats, no_cap, no_dev, 1, 1, K,
             reinterpret_cast<float*>(workspace), nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};
  p.ablate = 0;
  const Go2Plan pl = go2_plan(no_cap, Cin, Cout, workspace != nullptr, workspace_bytes);
  return launch_go2(pl, p, g, reinterpret_cast<const uint16_t*>(weight_frag), false, nullptr, as_stream(stream), true);
}

extern "C" int cnrma_sparse_conv_go_f16x3(const float* in_feats, const float* in_amax, int Cin, const void* tile_union,
                                          const void* weight_frag, int Cout, const float* scale, const float* shift,
                                          const float* residual, int act, float* out_feats, float* out_amax, int64_t no_cap,
                                          const int32_t* no_dev, void* workspace, size_t workspace_bytes,
                                          void* tile_counters, void* stream) {
  if (in_feats == nullptr || in_amax == nullptr || tile_union == nullptr || weight_frag == nullptr || out_feats == nullptr ||
      Cin <= 0 || Cin % BK != 0 || Cout < 64 || no_cap <= 0)
    return CNRMA_EINVAL;
  const int K = 27;
  hipStream_t st = as_stream(stream);
  const size_t tiles = (size_t)ceil_div(no_cap, GO_BM);
  const char* w = reinterpret_cast<const char*>(tile_union);
  GoArgs g;
  g.hdr = reinterpret_cast<const int32_t*>(w);      w += go_align(tiles * GO_HDR * 4);
  g.rows = reinterpret_cast<const int32_t*>(w);     w += go_align(tiles * GO_ROWS * 4);
  g.lidx = reinterpret_cast<const uint16_t*>(w);
  ConvArgs p{in_feats, Cin, nullptr, K, nullptr, Cout, scale, shift, residual, act, out_feats, no_cap, no_dev, 1, 1, K,
             reinterpret_cast<float*>(workspace), nullptr, 0, nullptr, 0, in_amax, nullptr, out_amax, nullptr, 0};
  p.ablate = CNRMA_CONV_TUNE.ablate;
  const uint16_t* wfrag = reinterpret_cast<const uint16_t*>(weight_frag);
  p.w_amax = reinterpret_cast<const float*>(wfrag + 2 * (int64_t)K * Cin * conv_cout_padded(Cout));
  const int bn = Cout >= 128 ? 128 : 64;
  const int ns = Cin / BK;
  {
    const Go2Plan pl = go2_plan(no_cap, Cin, Cout, workspace != nullptr, workspace_bytes);
#ifndef CNRMA_EXPERIMENTS
    (void)ns; (void)bn; (void)tiles;
    return launch_go2(pl, p, g, wfrag, residual != nullptr, nullptr, st);       // the product library: the second form, nothing else
  }
}
#else
    // the other ablation masks (diagnostic kernels with phases switched off) exist in the first form only
    if (pl.form >= 2 && (CNRMA_CONV_TUNE.ablate & ~64) == 0 && (uint64_t)no_cap * (uint64_t)Cin * 4u < (1ull << 32))
      return launch_go3(pl, p, g, wfrag, residual != nullptr, (CNRMA_CONV_TUNE.ablate & 64) ? tile_counters : nullptr, st);
    if (pl.form >= 1 && (CNRMA_CONV_TUNE.ablate & ~64) == 0)
      return launch_go2(pl, p, g, wfrag, residual != nullptr, (CNRMA_CONV_TUNE.ablate & 64) ? tile_counters : nullptr, st);
  }
  // short layers: split over the 32-channel slices (every block still runs all 27 offsets of its slices); partial slabs are
  // reduced by conv_reduce_kernel in a fixed order
  int splits = 1;
  const int64_t blocks = (int64_t)tiles * ceil_div(Cout, bn);
  const int force = CNRMA_CONV_TUNE.splits;
  if (workspace != nullptr && ns > 1 && (blocks < 384 || force > 0)) {
    splits = force > 0 ? force : (int)ceil_div(768, blocks);
    if (splits > ns) splits = ns;
    const size_t per = (size_t)no_cap * Cout * sizeof(float);
    if (per > 0 && (size_t)splits * per > workspace_bytes) splits = (int)(workspace_bytes / per);
    if (splits < 2) splits = 1;
  }
  g.slices_per_split = (int)ceil_div(ns, splits);
  splits = (int)ceil_div(ns, g.slices_per_split);
  p.splits = splits;
  g.counters = splits > 1 ? reinterpret_cast<unsigned*>(tile_counters) : nullptr;
  const bool has_res = residual != nullptr && splits == 1;
  dim3 grid((unsigned)tiles, (unsigned)ceil_div(Cout, bn), (unsigned)splits);
  if (p.ablate != 0 && !has_res) {                         // diagnostic instantiations (timing experiments only)
    if (bn == 128) hipLaunchKernelGGL((sparse_conv_go_kernel<1, 4, 2, 1, false, 1, true>), grid, dim3(256), 0, st, p, g, wfrag);
    else hipLaunchKernelGGL((sparse_conv_go_kernel<1, 2, 2, 1, false, 2, true>), grid, dim3(256), 0, st, p, g, wfrag);
  } else if (bn == 128) {
    if (has_res) hipLaunchKernelGGL((sparse_conv_go_kernel<1, 4, 2, 1, true>), grid, dim3(256), 0, st, p, g, wfrag);
    else hipLaunchKernelGGL((sparse_conv_go_kernel<1, 4, 2, 1, false>), grid, dim3(256), 0, st, p, g, wfrag);
  } else {
    if (CNRMA_CONV_TUNE.pf == 3) {                             // A/B aid: the 2 x 2 waves-over-rows-x-columns form
      if (has_res) hipLaunchKernelGGL((sparse_conv_go_kernel<2, 2, 1, 1, true>), grid, dim3(256), 0, st, p, g, wfrag);
      else hipLaunchKernelGGL((sparse_conv_go_kernel<2, 2, 1, 1, false>), grid, dim3(256), 0, st, p, g, wfrag);
    } else if (has_res) hipLaunchKernelGGL((sparse_conv_go_kernel<1, 2, 2, 1, true, 2>), grid, dim3(256), 0, st, p, g, wfrag);
    else hipLaunchKernelGGL((sparse_conv_go_kernel<1, 2, 2, 1, false, 2>), grid, dim3(256), 0, st, p, g, wfrag);
  }
  if (splits > 1 && g.counters == nullptr) {
    int64_t rb = ceil_div(no_cap * Cout / 4 + 1, 256);
    if (rb > 4096) rb = 4096;
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, p);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}
#endif

extern "C" int cnrma_sparse_conv_f16x3(const float* in_feats, const float* in_amax, int Cin, const int32_t* nbr, int K,
                                       const void* weight_split, int Cout, const float* scale, const float* shift,
                                       const float* residual, int act, float* out_feats, float* out_amax,
                                       int64_t no_cap, const int32_t* no_dev, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  if (weight_split == nullptr || Cin % 32 != 0 || in_feats == nullptr || in_amax == nullptr) return CNRMA_EINVAL;
  return launch_conv(in_feats, Cin, nbr, K, nullptr, Cout, scale, shift, residual, act, out_feats, no_cap, no_dev, 1,
                     workspace, workspace_bytes, as_stream(stream), weight_split, nullptr, 0, nullptr, 0, 1, in_amax,
                     out_amax);
}

extern "C" size_t cnrma_sparse_conv_bf16_weight_bytes(int K, int Cin, int Cout) {
  return (size_t)K * Cin * conv_cout_padded(Cout) * sizeof(uint16_t);
}

extern "C" int cnrma_sparse_conv_prepare_weights_bf16(const float* weight, int K, int Cin, int Cout, void* weight_bf16,
                                                      void* stream) {
  if (K <= 0 || Cin <= 0 || Cout <= 0 || weight_bf16 == nullptr) return CNRMA_EINVAL;
  int64_t blocks = ceil_div((int64_t)K * Cin * conv_cout_padded(Cout), 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight,
                     reinterpret_cast<__bf16*>(weight_bf16), K, Cin, Cout);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_prepare_weights_bf16_t(const float* weight, int K, int Cin, int Cout, int flip,
                                                        void* weight_bf16, void* stream) {
  if (K <= 0 || Cin <= 0 || Cout <= 0 || Cout % 32 != 0 || weight == nullptr || weight_bf16 == nullptr) return CNRMA_EINVAL;
  int64_t blocks = ceil_div((int64_t)K * Cout * conv_cout_padded(Cin), 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(prep_weights_bf16_t_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weight,
                     reinterpret_cast<__bf16*>(weight_bf16), K, Cin, Cout, flip ? 1 : 0);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_conv_bf16(const float* in_feats, int Cin, const int32_t* nbr, int K, const void* weight_bf16,
                                      int Cout, const float* scale, const float* shift, const float* residual, int act,
                                      float* out_feats, int64_t no_cap, const int32_t* no_dev, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (weight_bf16 == nullptr || Cin % 32 != 0 || in_feats == nullptr) return CNRMA_EINVAL;
  return launch_conv(in_feats, Cin, nbr, K, nullptr, Cout, scale, shift, residual, act, out_feats, no_cap, no_dev, 1,
                     workspace, workspace_bytes, as_stream(stream), weight_bf16, nullptr, 0, nullptr, 0, 2);
}

extern "C" int cnrma_sparse_convtr_gen_f16x3(const int32_t* in_coords, const float* in_feats, const float* in_amax,
                                             int64_t n_cap, const int32_t* n_dev, int Cin, int half_stride,
                                             const void* weight_split, int Cout, const float* scale,
                                             const float* shift, int act, int32_t* out_coords, float* out_feats,
                                             float* out_amax, void* stream) {
  if (n_cap <= 0 || half_stride <= 0 || weight_split == nullptr || Cin % 32 != 0 || in_amax == nullptr) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(convtr_coords_kernel, dim3((unsigned)ceil_div(n_cap * 8, 256)), dim3(256), 0, st, in_coords,
                     n_cap, n_dev, half_stride, out_coords);
  return launch_conv(in_feats, Cin, nullptr, 1, nullptr, Cout, scale, shift, nullptr, act, out_feats, n_cap, n_dev, 8,
                     nullptr, 0, st, weight_split, nullptr, 0, nullptr, 0, 1, in_amax, out_amax);
}

extern "C" int cnrma_sparse_split_features(const float* feats, int64_t n_cap, const int32_t* n_dev, int C,
                                           void* out_split, void* stream) {
  if (n_cap <= 0 || C <= 0 || C % 8 != 0) return CNRMA_EINVAL;
  const int64_t work = n_cap * (C / 8);
  hipLaunchKernelGGL(split_features_kernel, dim3((unsigned)ceil_div(work > C / 8 ? work : C / 8, 256)), dim3(256), 0,
                     as_stream(stream), feats, n_cap, n_dev, C, reinterpret_cast<uint16_t*>(out_split));
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_convtr_gen_bf16x6(const int32_t* in_coords, const float* in_feats, const void* in_split,
                                              int64_t n_cap, const int32_t* n_dev, int Cin, int half_stride,
                                              const void* weight_split, int Cout, const float* scale,
                                              const float* shift, int act, int32_t* out_coords, float* out_feats,
                                              void* out_split, void* stream) {
  if (n_cap <= 0 || half_stride <= 0 || weight_split == nullptr || Cin % 32 != 0) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(convtr_coords_kernel, dim3((unsigned)ceil_div(n_cap * 8, 256)), dim3(256), 0, st, in_coords,
                     n_cap, n_dev, half_stride, out_coords);
  return launch_conv(in_feats, Cin, nullptr, 1, nullptr, Cout, scale, shift, nullptr, act, out_feats, n_cap, n_dev, 8,
                     nullptr, 0, st, weight_split, in_split, n_cap, out_split, 8 * n_cap);
}

extern "C" int cnrma_sparse_convtr_gen_f32(const int32_t* in_coords, const float* in_feats, int64_t n_cap,
                                           const int32_t* n_dev, int Cin, int half_stride, const float* weight,
                                           int Cout, const float* scale, const float* shift, int act,
                                           int32_t* out_coords, float* out_feats, void* stream) {
  if (n_cap <= 0 || half_stride <= 0) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(convtr_coords_kernel, dim3((unsigned)ceil_div(n_cap * 8, 256)), dim3(256), 0, st, in_coords,
                     n_cap, n_dev, half_stride, out_coords);
  // 8 weight slices; slice k reads in[i] (identity map, K = 1) and writes row k*n + i
  return launch_conv(in_feats, Cin, nullptr, 1, weight, Cout, scale, shift, nullptr, act, out_feats, n_cap, n_dev, 8,
                     nullptr, 0, st);
}

extern "C" int cnrma_sparse_maxpool_f32(const float* in_feats, int C, const int32_t* nbr, int K, float* out_feats,
                                        int64_t no_cap, const int32_t* no_dev, void* stream) {
  if (no_cap <= 0 || C <= 0 || K <= 0) return CNRMA_EINVAL;
  hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)ceil_div(no_cap * C, 256)), dim3(256), 0, as_stream(stream),
                     in_feats, C, nbr, K, out_feats, no_cap, no_dev);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cnrma_instnorm_workspace_bytes(int C) { return (size_t)(1024 * 2 * C + 2 * C) * sizeof(double); }

extern "C" int cnrma_sparse_instnorm_f32(const float* in_feats, int64_t n_cap, const int32_t* n_dev,
                                         const int32_t* row0_dev, int C, const float* weight, const float* bias, float eps,
                                         int relu, float* out_feats, double* stats_ws, void* stream) {
  if (n_cap <= 0 || C <= 0 || C > 256) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int nblk = 1024;
  double* part = stats_ws + 2 * C;
  hipLaunchKernelGGL(colstats_partial_kernel, dim3(nblk), dim3(256), 0, st, in_feats, n_cap, n_dev, row0_dev, C, part);
  hipLaunchKernelGGL(colstats_final_kernel, dim3((unsigned)C), dim3(256), 0, st, part, nblk, C, n_cap, n_dev, stats_ws);
  if (out_feats != nullptr)            // NULL: statistics only (cnrma_sparse_instnorm_maxpool_f32 applies them)
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3(grid_for(n_cap * C, 256, 4096)), dim3(256), 0, st, in_feats, n_cap,
                       n_dev, row0_dev, C, stats_ws, weight, bias, eps, relu, out_feats);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_instnorm_maxpool_f32(const float* in_feats, int C, const double* stats, const float* weight,
                                                 const float* bias, float eps, int relu, const int32_t* nbr, int K,
                                                 float* out_feats, int64_t no_cap, const int32_t* no_dev, float* out_amax,
                                                 void* stream) {
  if (no_cap <= 0 || C <= 0 || (C & 3) != 0 || K <= 0 || in_feats == nullptr || stats == nullptr || nbr == nullptr ||
      out_feats == nullptr)
    return CNRMA_EINVAL;
  hipLaunchKernelGGL(instnorm_maxpool_kernel, dim3((unsigned)ceil_div(no_cap * (C / 4), 256)), dim3(256), 0, as_stream(stream),
                     in_feats, C, stats, weight, bias, eps, relu, nbr, K, out_feats, no_cap, no_dev, out_amax);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// BatchNorm1d training backward for the rows of x [n][C] (C <= 256): stats = {mean[C], biased var[C]} in fp64 as the forward
// (cnrma_sparse_instnorm_f32 with the layer's eps) left them; ws: cnrma_instnorm_workspace_bytes(C).
extern "C" int cnrma_bn_backward_f32(const float* grad_out, const float* x, int64_t n, int C, const double* stats,
                                     const float* weight, float eps, float* grad_in, float* grad_weight, float* grad_bias,
                                     double* ws, void* stream) {
  if (n <= 0 || C <= 0 || C > 256 || grad_out == nullptr || x == nullptr || stats == nullptr || grad_in == nullptr ||
      ws == nullptr)
    return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int nblk = 1024;
  double* sums = ws;                  // [2C]
  double* part = ws + 2 * C;          // [nblk][2C]
  hipLaunchKernelGGL(colsum2_partial_kernel, dim3(nblk), dim3(256), 0, st, grad_out, x, n, C, part);
  hipLaunchKernelGGL(bn_backward_final_kernel, dim3((unsigned)C), dim3(256), 0, st, part, nblk, C, n, stats, eps, grad_weight,
                     grad_bias, sums);
  hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(grid_for(n * C, 256, 4096)), dim3(256), 0, st, grad_out, x, n, C, stats,
                     sums, weight, eps, grad_in);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// BatchNorm1d in training mode over the rows of x [n][C] (C <= 256, C % 4 == 0, n >= 2), fused with what follows it in a
// residual block: out = act( (x - mean) / sqrt(var + eps) * weight + bias [+ residual] ), act = `relu`: 0 none, 1 ReLU, 2 ELU;
// running statistics and batch
// counter updated in the statistics kernel (momentum given).  stats_ws: cnrma_instnorm_workspace_bytes(C); on return
// stats_ws[0..C) = mean, [C..2C) = biased variance (fp64): the backward's `stats`.
extern "C" int cnrma_bn_train_forward_f32(const float* x, int64_t n, int C, const float* weight, const float* bias, float eps,
                                          const float* residual, int relu, float momentum, float* running_mean,
                                          float* running_var, int64_t* num_batches_tracked, float* out, double* stats_ws,
                                          void* stream) {
  if (n < 2 || C <= 0 || C > 256 || (C & 3) || x == nullptr || out == nullptr || stats_ws == nullptr) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int nblk = 1024;
  double* part = stats_ws + 2 * C;
  hipLaunchKernelGGL(colstats_partial_kernel, dim3(nblk), dim3(256), 0, st, x, n, nullptr, nullptr, C, part);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((unsigned)C), dim3(256), 0, st, part, nblk, C, n, stats_ws, momentum, running_mean,
                     running_var, num_batches_tracked);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(n * C / 4, 256, 4096)), dim3(256), 0, st, x, n, C, stats_ws, weight, bias, eps,
                     residual, relu, out);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// its backward: y != NULL: the forward ended in an activation (act 1 ReLU: grad_out is masked by y > 0; 2 ELU: times y + 1 where
// y <= 0; y = the forward's output);
// grad_residual != NULL: receives the masked gradient (the residual branch's).  ws: cnrma_instnorm_workspace_bytes(C).
extern "C" int cnrma_bn_train_backward_f32(const float* grad_out, const float* x, const float* y, int act, int64_t n, int C,
                                           const double* stats, const float* weight, float eps, float* grad_in,
                                           float* grad_residual, float* grad_weight, float* grad_bias, double* ws, void* stream) {
  if (n <= 0 || C <= 0 || C > 256 || (C & 3) || grad_out == nullptr || x == nullptr || stats == nullptr || grad_in == nullptr ||
      ws == nullptr)
    return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  const int nblk = 1024;
  double* sums = ws;                  // [2C]
  double* part = ws + 2 * C;          // [nblk][2C]
  hipLaunchKernelGGL(bn_colsum2_partial_kernel, dim3(nblk), dim3(256), 0, st, grad_out, x, y, act, n, C, part);
  hipLaunchKernelGGL(bn_backward_final_kernel, dim3((unsigned)C), dim3(256), 0, st, part, nblk, C, n, stats, eps, grad_weight,
                     grad_bias, sums);
  hipLaunchKernelGGL(bn_backward_apply2_kernel, dim3(grid_for(n * C / 4, 256, 4096)), dim3(256), 0, st, grad_out, x, y, act, n, C, stats,
                     sums, weight, eps, grad_in, grad_residual);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_union_add_f32(const int32_t* a_coords, const float* a_feats, int64_t na_cap,
                                          const int32_t* na_dev, const int32_t* b_coords, const float* b_feats,
                                          int64_t nb_cap, const int32_t* nb_dev, int C, uint64_t* a_hash_keys,
                                          int32_t* a_hash_vals, int64_t hash_cap, int32_t* out_coords,
                                          float* out_feats, int64_t out_cap, int32_t* n_out, void* workspace,
                                          void* stream) {
  if (na_cap <= 0 || nb_cap <= 0 || C <= 0 || out_cap < na_cap) return CNRMA_EINVAL;
  hipStream_t st = as_stream(stream);
  // workspace: match[nb] int32, idx[nb] int32, flag[nb] u8, n_new int32, scan ws
  char* p = reinterpret_cast<char*>(workspace);
  int64_t n4 = (nb_cap + 3) / 4 * 4;
  int32_t* match = reinterpret_cast<int32_t*>(p); p += n4 * 4;
  int32_t* idx = reinterpret_cast<int32_t*>(p); p += n4 * 4;
  uint8_t* flag = reinterpret_cast<uint8_t*>(p); p += n4;
  int32_t* n_new = reinterpret_cast<int32_t*>(p); p += 16;
  hipLaunchKernelGGL(union_flag_kernel, dim3((unsigned)ceil_div(nb_cap, 256)), dim3(256), 0, st, b_coords, nb_cap,
                     nb_dev, a_hash_keys, a_hash_vals, hash_cap, match, flag);
  int rc = cnrma_mask_to_index(flag, idx, n_new, nb_cap, p, st);
  if (rc) return rc;
  hipLaunchKernelGGL(union_copy_a_kernel, dim3(grid_for(na_cap * C, 256, 4096)), dim3(256), 0, st, a_coords, a_feats,
                     na_cap, na_dev, C, out_coords, out_feats);
  hipLaunchKernelGGL(union_merge_b_kernel, dim3(grid_for(nb_cap * C, 256, 4096)), dim3(256), 0, st, b_coords, b_feats,
                     nb_cap, nb_dev, C, match, idx, na_cap, na_dev, a_hash_keys, a_hash_vals, hash_cap, out_coords,
                     out_feats, n_new, n_out, out_cap);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cnrma_union_workspace_bytes(int64_t nb) {
  int64_t n4 = (nb + 3) / 4 * 4;
  return (size_t)(n4 * 9 + 16) + cnrma_scan_workspace_bytes(nb) + 256;
}

extern "C" int cnrma_sparse_interp_f32(const int32_t* q_coords, int64_t n_cap, const int32_t* n_dev,
                                       const float* score, const uint64_t* s_hash_keys, const int32_t* s_hash_vals,
                                       int64_t hash_cap, int score_stride, float* out, void* stream) {
  if (n_cap <= 0 || score_stride <= 0) return CNRMA_EINVAL;
  hipLaunchKernelGGL(interp_kernel, dim3((unsigned)ceil_div(n_cap, 256)), dim3(256), 0, as_stream(stream), q_coords,
                     n_cap, n_dev, score, s_hash_keys, s_hash_vals, hash_cap, score_stride, out);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_sparse_prune_f32(const int32_t* in_coords, const float* in_feats, int64_t n_cap,
                                      const int32_t* n_dev, int C, const int32_t* sel_index, int32_t* out_coords,
                                      float* out_feats, void* stream) {
  if (n_cap <= 0 || C <= 0) return CNRMA_EINVAL;
  hipLaunchKernelGGL(prune_kernel, dim3(grid_for(n_cap * C, 256, 4096)), dim3(256), 0, as_stream(stream), in_coords,
                     in_feats, n_cap, n_dev, C, sel_index, out_coords, out_feats);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rowmax_f32(const float* in, int64_t n_cap, const int32_t* n_dev, int C, float* out, void* stream) {
  if (n_cap <= 0 || C <= 0) return CNRMA_EINVAL;
  hipLaunchKernelGGL(rowmax_kernel, dim3((unsigned)ceil_div(n_cap, 256)), dim3(256), 0, as_stream(stream), in, n_cap,
                     n_dev, C, out);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
