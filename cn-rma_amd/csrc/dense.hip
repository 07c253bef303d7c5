// Dense unprojection + accumulate + mean in ONE pass over the voxel grid.
//
// The reference makes V full passes: zero-fill + scatter a C x G volume per view, add it to a running sum,
// and finally divides by the view count (ray_marching.py:21-69, :220-257) -- 15 GB of writes per scene at the
// ScanNet shape.  Here each lane owns one voxel, loops over the V views with the accumulators in registers and
// writes the mean volume exactly once (compulsory traffic: read the feature planes, write C x G floats).
//
// Lane -> voxel mapping: linear index with z fastest, so the stores of one channel are coalesced; the gathers hit
// channels-last pixels (one 128-B line per pixel at C = 32).
// Bit-exactness: world = fl(fl(i*vs)+o); cam = fma chain (MKL bmm); px,py = rint(cam0/cam2) (IEEE division);
// sum in view order; mean = sum / (float)count.  Compiled with -ffp-contract=off.
#include "common.h"

#include <stdlib.h>

namespace {

struct DenseParams {
  int V, C, H, W, X, Y, Z;
  float vs, ox, oy, oz;
};

__device__ __forceinline__ void voxel_world(const DenseParams& p, int64_t g, float* wx, float* wy, float* wz) {
  const int z = (int)(g % p.Z);
  const int64_t t = g / p.Z;
  const int y = (int)(t % p.Y);
  const int x = (int)(t / p.Y);
  *wx = (float)x * p.vs + p.ox;  // ray_marching.py:48  (two roundings)
  *wy = (float)y * p.vs + p.oy;
  *wz = (float)z * p.vs + p.oz;
}

// Brick order of the voxels (cooperative kernel).  Voxels that share a pixel lie on one camera ray, 8-64 voxels apart;
// a gathered pixel line is re-used only if two of them are processed by the SAME XCD (private 4 MB L2) at about the same
// time.  With the plain z-fastest order the voxels in flight on an XCD are a pile of thin vertical columns that a ray
// crosses once (TCC hit rate 22 %).  Here the grid is cut into bricks of st x st columns x zt layers (default 16 x 16 x 32
// = 8192 voxels = 32 workgroups), z fastest inside a column, columns in tt x tt tiles; every brick is a chunk that one XCD
// group owns (chunks c, c + 8, ... as before), so the ~8 bricks an XCD has in flight are compact 0.64 x 0.64 x 1.28 m
// boxes in which a ray meets several of its voxels.  Measured at the north-star shape (scripts/dense_ab.py, results
// bit-identical): 13.3 -> 10.7 ms; 64 x 64 x 16 slabs 12.8, 32 x 32 x 16 11.5, 16 x 16 x 64 10.8, 8 x 8 x 32 11.5,
// 128 x 128 x 16 19.0 ms.  Stores stay 32-byte runs along z per lane group, merged in L2 (zt >= 16).
struct SlabOrder { int on, nsx, nsy, nsz, zt, st, tt, zi; };  // zt: z-layers per brick, st: brick side, tt: tile side (columns), zi: inner z run

__device__ __forceinline__ bool slab_decode(const DenseParams& p, const SlabOrder& o, int64_t gv, int* x, int* y, int* z) {
  const int64_t per = (int64_t)o.st * o.st * o.zt;
  const int64_t sv = gv / per;
  const int r = (int)(gv - sv * per);
  if (sv >= (int64_t)o.nsx * o.nsy * o.nsz) return false;
  const int sy = (int)(sv % o.nsy), sx = (int)((sv / o.nsy) % o.nsx), sz = (int)(sv / ((int64_t)o.nsy * o.nsx));
  // order inside a brick: [tile of tt x tt columns][outer z][column in tile][inner z run of zi]
  const int tcols = o.tt * o.tt, zo_n = o.zt / o.zi;
  const int zin = r % o.zi, q = r / o.zi;
  const int in = q % tcols, q2 = q / tcols;
  const int zo = q2 % zo_n, tile = q2 / zo_n;
  const int zl = zo * o.zi + zin;
  const int tpr = o.st / o.tt;
  *x = sx * o.st + (tile / tpr) * o.tt + in / o.tt;
  *y = sy * o.st + (tile % tpr) * o.tt + in % o.tt;
  *z = sz * o.zt + zl;
  return *x < p.X && *y < p.Y && *z < p.Z;
}

// project one voxel into one view: returns validity, pixel in (*px,*py) (ray_marching.py:51-58)
__device__ __forceinline__ bool project(const float* __restrict__ P, float wx, float wy, float wz, int H, int W,
                                        float* rx, float* ry) {
  float cam[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = P[r * 4 + 0] * wx;
    acc = fmaf(P[r * 4 + 1], wy, acc);
    acc = fmaf(P[r * 4 + 2], wz, acc);
    acc = fmaf(P[r * 4 + 3], 1.0f, acc);
    cam[r] = acc;
  }
  *rx = rintf(cam[0] / cam[2]);
  *ry = rintf(cam[1] / cam[2]);
  // comparisons in float are equivalent to the reference's int64 ones: NaN / out-of-range casts land on
  // INT64_MIN there (negative -> invalid), and are rejected here as well
  return (*rx >= 0.0f) && (*ry >= 0.0f) && (*rx < (float)W) && (*ry < (float)H) && (cam[2] > 0.0f);
}

// CT channels per lane kept in registers; grid.y walks channel chunks
template <int CT>
__global__ __launch_bounds__(256) void backproject_accum_kernel(DenseParams p, const float* __restrict__ feat,
                                                                const float* __restrict__ proj,
                                                                float* __restrict__ volume,
                                                                int32_t* __restrict__ count) {
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const int c0 = blockIdx.y * CT;
  float wx, wy, wz;
  voxel_world(p, g, &wx, &wy, &wz);
  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.0f;
  int cnt = 0;
  const int64_t plane = (int64_t)p.H * p.W * p.C;
  for (int v = 0; v < p.V; ++v) {
    float rx, ry;
    if (!project(proj + v * 12, wx, wy, wz, p.H, p.W, &rx, &ry)) continue;
    ++cnt;
    const float* f = feat + v * plane + ((int64_t)(int)ry * p.W + (int)rx) * p.C + c0;
    if constexpr (CT % 4 == 0) {
#pragma unroll
      for (int c = 0; c < CT; c += 4) {
        float4 q = *reinterpret_cast<const float4*>(f + c);
        acc[c] += q.x; acc[c + 1] += q.y; acc[c + 2] += q.z; acc[c + 3] += q.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[c] += f[c];
    }
  }
  const float denom = (float)cnt;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if (c0 + c < p.C) volume[(int64_t)(c0 + c) * G + g] = cnt > 0 ? acc[c] / denom : 0.0f;
  }
  if (blockIdx.y == 0) count[g] = cnt;
}

// ---- cooperative-gather variant (C % 4 == 0) -------------------------------------------------------------------
// A wave owns 64 consecutive voxels.  Per view every lane projects ITS voxel; then the wave walks the voxels in
// groups of 64/LPV and LPV lanes read one pixel's channel vector together (LPV x 16 B = one full 128-B line at
// C = 32), so a gather instruction touches 64/LPV lines instead of 64 and every fetched byte is used.  Lane l ends
// up holding channels 4*(l % LPV) .. +3 of voxels (g * 64/LPV + l / LPV), g = 0 .. LPV-1.
template <int LPV>
__global__ __launch_bounds__(256) void backproject_accum_coop_kernel(DenseParams p, const float* __restrict__ feat,
                                                                     const float* __restrict__ proj,
                                                                     float* __restrict__ volume,
                                                                     int32_t* __restrict__ count, int chunk_blocks,
                                                                     int64_t n_phys, SlabOrder ord) {
  constexpr int VPG = 64 / LPV;              // voxels served per gather instruction
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  const int lane = threadIdx.x & 63;
  // XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the XCD group), each with a
  // private L2.  A pixel's channel vector is re-read by the ~9 voxels along its ray; with consecutive blocks on
  // different XCDs every one of those reads misses its L2 (profiles/r02: 22 % hit rate, 80 GB over the fabric per launch
  // at the north-star shape against 20 GB of compulsory traffic).  Here an XCD group owns whole chunks of `chunk_blocks`
  // consecutive blocks (one x-plane of the grid) -- chunks c, c + 8, c + 16, ... --, so rays that run inside a plane find
  // their pixels in the group's L2; the chunks stay interleaved over the whole grid, which keeps the groups evenly
  // loaded (8 contiguous slabs, one per XCD, measured 17 % slower: the frustum makes slabs unequal).
  // persistent form: the grid may be smaller than the number of logical blocks (gridDim.x a multiple of 8, so that
  // b % 8 -- the XCD group -- is the same for every logical block a workgroup takes); each workgroup walks
  // pb = blockIdx.x, blockIdx.x + gridDim.x, ...
  for (int64_t pb = blockIdx.x; pb < n_phys; pb += gridDim.x) {
  int64_t lb = pb;
  if (chunk_blocks > 0) {
    const int64_t grp = pb & 7, k = pb >> 3;
    lb = (grp + 8 * (k / chunk_blocks)) * chunk_blocks + k % chunk_blocks;
  }
  const int64_t wave_base = (lb * blockDim.x + threadIdx.x) - lane;
  const int64_t g = wave_base + lane;
  const int c0 = blockIdx.y * (4 * LPV);
  float wx = 0.f, wy = 0.f, wz = 0.f;
  bool in_grid;
  int64_t lin = -1;                               // this lane's voxel as an index into [X][Y][Z] (-1: none)
  if (ord.on) {
    int x = 0, y = 0, z = 0;
    in_grid = slab_decode(p, ord, g, &x, &y, &z);
    if (__ballot(in_grid) == 0ull) continue;
    if (in_grid) {
      lin = ((int64_t)x * p.Y + y) * p.Z + z;
      wx = (float)x * p.vs + p.ox; wy = (float)y * p.vs + p.oy; wz = (float)z * p.vs + p.oz;      // as voxel_world()
    }
  } else {
    if (wave_base >= G) continue;
    in_grid = g < G;
    if (in_grid) { voxel_world(p, g, &wx, &wy, &wz); lin = g; }
  }
  float4 acc[LPV];
#pragma unroll
  for (int i = 0; i < LPV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int cnt = 0;
  const int sub = lane % LPV, vsel = lane / LPV;
  const int64_t plane = (int64_t)p.H * p.W * p.C;
  for (int v = 0; v < p.V; ++v) {
    float rx, ry;
    const bool ok = in_grid && project(proj + v * 12, wx, wy, wz, p.H, p.W, &rx, &ry);
    const unsigned long long any = __ballot(ok);
    if (any == 0ull) continue;
    cnt += ok ? 1 : 0;
    const int pix = ok ? ((int)ry * p.W + (int)rx) : -1;
    const float* fv = feat + v * plane + c0 + 4 * sub;
#pragma unroll
    for (int grp = 0; grp < LPV; ++grp) {
      if (((any >> (grp * VPG)) & ((VPG == 64) ? ~0ull : ((1ull << VPG) - 1ull))) == 0ull) continue;   // wave-uniform
      const int pq = __shfl(pix, grp * VPG + vsel, 64);
      if (pq >= 0) {
        const float4 q = *reinterpret_cast<const float4*>(fv + (int64_t)pq * p.C);
        acc[grp].x += q.x; acc[grp].y += q.y; acc[grp].z += q.z; acc[grp].w += q.w;
      }
    }
  }
  // the count of the voxel a lane accumulates for (group grp) lives in lane grp*VPG + vsel
#pragma unroll
  for (int grp = 0; grp < LPV; ++grp) {
    const int cv = __shfl(cnt, grp * VPG + vsel, 64);
    const int lo = __shfl((int)(lin & 0xffffffffLL), grp * VPG + vsel, 64), hi = __shfl((int)(lin >> 32), grp * VPG + vsel, 64);
    const int64_t gv = ((int64_t)hi << 32) | (uint32_t)lo;
    if (gv >= 0) {
      const float denom = (float)cv;
      const float4 a = acc[grp];
      const int c = c0 + 4 * sub;
      volume[(int64_t)(c + 0) * G + gv] = cv > 0 ? a.x / denom : 0.0f;
      volume[(int64_t)(c + 1) * G + gv] = cv > 0 ? a.y / denom : 0.0f;
      volume[(int64_t)(c + 2) * G + gv] = cv > 0 ? a.z / denom : 0.0f;
      volume[(int64_t)(c + 3) * G + gv] = cv > 0 ? a.w / denom : 0.0f;
    }
  }
  if (blockIdx.y == 0 && in_grid) count[lin] = cnt;
  }
}

template <int LPV>
int launch_accum_coop(const DenseParams& p, const float* feat, const float* proj, float* volume, int32_t* count,
                      hipStream_t st) {
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  int64_t nb = ceil_div(G, 256);
  // one chunk = the blocks of one x-plane (at least 32: keeps a group's L2 working set a compact slab piece)
  int64_t cb = ceil_div((int64_t)p.Y * p.Z, 256);
  if (cb < 32) cb = 32;
  SlabOrder ord{0, 0, 0, 0, 32, 16, 8, 32};
  const char* so = getenv("CNRMA_DENSE_SLAB");             // tuning / A-B aid: 0 = z-fastest linear order (x-plane chunks)
  if (so != nullptr ? so[0] == '1' : G < ((int64_t)1 << 40)) {
    ord.on = 1;
    const char* e;
    if ((e = getenv("CNRMA_SLAB_Z")) != nullptr) ord.zt = atoi(e);
    if ((e = getenv("CNRMA_SLAB_S")) != nullptr) ord.st = atoi(e);
    if ((e = getenv("CNRMA_SLAB_T")) != nullptr) ord.tt = atoi(e);
    ord.zi = ord.zt;
    if ((e = getenv("CNRMA_SLAB_ZI")) != nullptr) ord.zi = atoi(e);
    if (ord.zi < 1 || ord.zt % ord.zi != 0) return CNRMA_EINVAL;
    if (ord.zt < 1 || ord.tt < 1 || ord.st < ord.tt || ord.st % ord.tt != 0 || ((int64_t)ord.st * ord.st * ord.zt) % 256 != 0) return CNRMA_EINVAL;
    ord.nsx = (int)ceil_div(p.X, ord.st); ord.nsy = (int)ceil_div(p.Y, ord.st); ord.nsz = (int)ceil_div(p.Z, ord.zt);
    cb = (int64_t)ord.st * ord.st * ord.zt / 256;          // one supertile per chunk
    nb = (int64_t)ord.nsx * ord.nsy * ord.nsz * cb;
  }
  const char* env = getenv("CNRMA_DENSE_CHUNK");           // tuning / A-B aid: 0 = plain round-robin order
  if (env != nullptr) cb = atoll(env);
  int64_t gx = nb;
  if (cb > 0 && (nb >= 16 * cb || ord.on)) gx = ceil_div(ceil_div(nb, cb), 8) * 8 * cb;      // whole chunks for every XCD group
  else cb = 0;
  int64_t launch_x = gx;
  const char* pe = getenv("CNRMA_DENSE_PERSIST");          // tuning aid: workgroups per CU and channel sweep of a persistent grid
  const int per_cu = pe != nullptr ? atoi(pe) : 0;
  if (per_cu > 0 && gx > (int64_t)256 * per_cu) launch_x = (int64_t)256 * per_cu;     // a multiple of 8
  dim3 grid((unsigned)launch_x, (unsigned)ceil_div(p.C, 4 * LPV));
  hipLaunchKernelGGL((backproject_accum_coop_kernel<LPV>), grid, dim3(256), 0, st, p, feat, proj, volume, count, (int)cb, gx, ord);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void backproject_index_kernel(DenseParams p, const float* __restrict__ proj,
                                                                int32_t* __restrict__ px, int32_t* __restrict__ py,
                                                                uint8_t* __restrict__ valid) {
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  float wx, wy, wz, rx, ry;
  voxel_world(p, g, &wx, &wy, &wz);
  bool ok = project(proj, wx, wy, wz, p.H, p.W, &rx, &ry);
  const float lim = 2147483000.0f;
  px[g] = (rx > -lim && rx < lim) ? (int32_t)rx : INT32_MIN;
  py[g] = (ry > -lim && ry < lim) ? (int32_t)ry : INT32_MIN;
  valid[g] = ok ? 1 : 0;
}

template <int CT>
int launch_accum(const DenseParams& p, const float* feat, const float* proj, float* volume, int32_t* count,
                 hipStream_t st) {
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  dim3 grid((unsigned)ceil_div(G, 256), (unsigned)ceil_div(p.C, CT));
  hipLaunchKernelGGL((backproject_accum_kernel<CT>), grid, dim3(256), 0, st, p, feat, proj, volume, count);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int cnrma_backproject_accum_f32(const float* feat_nhwc, const float* proj, int V, int C, int H, int W,
                                           int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                           float* volume, int32_t* count, void* stream) {
  if (V <= 0 || C <= 0 || H <= 0 || W <= 0 || X <= 0 || Y <= 0 || Z <= 0) return CNRMA_EINVAL;
  DenseParams p{V, C, H, W, X, Y, Z, voxel_size, ox, oy, oz};
  hipStream_t st = as_stream(stream);
  const char* lpv = getenv("CNRMA_DENSE_LPV");             // tuning / A-B aid: lanes (x 4 channels) per voxel and sweep
  if (lpv != nullptr) {
    const int l = atoi(lpv);
    if (l == 16 && C % 64 == 0) return launch_accum_coop<16>(p, feat_nhwc, proj, volume, count, st);
    if (l == 32 && C % 128 == 0) return launch_accum_coop<32>(p, feat_nhwc, proj, volume, count, st);
    if (l == 4 && C % 16 == 0) return launch_accum_coop<4>(p, feat_nhwc, proj, volume, count, st);
  }
  if (C % 32 == 0) return launch_accum_coop<8>(p, feat_nhwc, proj, volume, count, st);
  if (C % 16 == 0) return launch_accum_coop<4>(p, feat_nhwc, proj, volume, count, st);
  if (C % 8 == 0) return launch_accum_coop<2>(p, feat_nhwc, proj, volume, count, st);
  if (C % 4 == 0) return launch_accum_coop<1>(p, feat_nhwc, proj, volume, count, st);
  return launch_accum<1>(p, feat_nhwc, proj, volume, count, st);
}

// Backward of the accumulate + mean w.r.t. the feature maps (training): volume[c][g] = sum_v feat[v][pix_v(g)][c] / count[g],
// so every valid (voxel, view) pair adds grad_volume[c][g] / count[g] to grad_feat[v][pix][c].  ~58 voxels share a pixel:
// float atomics (sum order not fixed: training only).  One lane per (voxel, 4 channels): a wave adds 128-byte channel
// vectors, the shape the memory-side atomic units run at full rate for.
template <int LPV>
__global__ __launch_bounds__(256) void backproject_backward_kernel(DenseParams p, const float* __restrict__ grad_volume,
                                                                   const int32_t* __restrict__ count,
                                                                   const float* __restrict__ proj,
                                                                   float* __restrict__ grad_feat) {
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t g = t / LPV;
  const int sub = (int)(t % LPV);
  if (g >= G) return;
  const int cnt = count[g];
  if (cnt <= 0) return;
  float wx, wy, wz;
  voxel_world(p, g, &wx, &wy, &wz);
  const float denom = (float)cnt;
  const int64_t plane = (int64_t)p.H * p.W * p.C;
  for (int c = 4 * sub; c < p.C; c += 4 * LPV) {
    float gq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gq[j] = (c + j < p.C) ? grad_volume[(int64_t)(c + j) * G + g] / denom : 0.0f;
    for (int v = 0; v < p.V; ++v) {
      float rx, ry;
      if (!project(proj + v * 12, wx, wy, wz, p.H, p.W, &rx, &ry)) continue;
      float* q = grad_feat + v * plane + ((int64_t)(int)ry * p.W + (int)rx) * p.C + c;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < p.C) atomicAdd(q + j, gq[j]);
    }
  }
}

extern "C" int cnrma_backproject_backward_f32(const float* grad_volume, const int32_t* count, const float* proj, int V,
                                              int C, int H, int W, int X, int Y, int Z, float voxel_size, float ox,
                                              float oy, float oz, float* grad_feat_nhwc, void* stream) {
  if (V <= 0 || C <= 0 || H <= 0 || W <= 0 || X <= 0 || Y <= 0 || Z <= 0 || grad_volume == nullptr || count == nullptr ||
      grad_feat_nhwc == nullptr)
    return CNRMA_EINVAL;
  DenseParams p{V, C, H, W, X, Y, Z, voxel_size, ox, oy, oz};
  hipStream_t st = as_stream(stream);
  hipError_t e = cnrma_fill_bytes(grad_feat_nhwc, 0, (size_t)V * H * W * C * sizeof(float), st);
  if (e != hipSuccess) return -(int)e;
  const int64_t G = (int64_t)X * Y * Z;
  if (C >= 32) {
    hipLaunchKernelGGL((backproject_backward_kernel<8>), dim3((unsigned)ceil_div(G * 8, 256)), dim3(256), 0, st, p,
                       grad_volume, count, proj, grad_feat_nhwc);
  } else {
    hipLaunchKernelGGL((backproject_backward_kernel<1>), dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, st, p, grad_volume,
                       count, proj, grad_feat_nhwc);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_backproject_index_f32(const float* proj_view, int H, int W, int X, int Y, int Z, float voxel_size,
                                           float ox, float oy, float oz, int32_t* px, int32_t* py, uint8_t* valid,
                                           void* stream) {
  if (H <= 0 || W <= 0 || X <= 0 || Y <= 0 || Z <= 0) return CNRMA_EINVAL;
  DenseParams p{1, 1, H, W, X, Y, Z, voxel_size, ox, oy, oz};
  const int64_t G = (int64_t)X * Y * Z;
  hipLaunchKernelGGL(backproject_index_kernel, dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, as_stream(stream), p,
                     proj_view, px, py, valid);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
