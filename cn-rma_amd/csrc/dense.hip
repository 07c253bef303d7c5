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


namespace {

static __device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

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
struct SlabOrder { int on, nsx, nsy, nsz, zt, st, tt, zi, nt, own, stagger, groups; };  // zt: z-layers per brick, st: brick side, tt: tile side (columns), zi: inner z run

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

// Lattice assignment of a brick's z-columns to its workgroups (lockstep schedule; zt == 32, st % 8 == 0): workgroup w of
// the st*st/8 takes 8 whole columns on a (2 x 4) lattice with pitch (st/2, st/4) -- column j at
// (w / (st/4) + (st/2) * (j / 4),  w % (st/4) + (st/4) * (j % 4)) --, so every workgroup samples the whole brick and meets
// each view's frustum boundary in the same proportion as its neighbours: equal work per view, the precondition for staying in
// phase.  A wave still owns 2 columns x 32 consecutive z (full 128-byte store runs per channel plane).
__device__ __forceinline__ bool lattice_decode(const DenseParams& p, const SlabOrder& o, int64_t lb, int tid, int* x, int* y, int* z) {
  const int cb = o.st * o.st / 8;
  const int64_t sv = lb / cb;
  if (sv >= (int64_t)o.nsx * o.nsy * o.nsz) return false;
  const int w = (int)(lb - sv * cb), j = tid >> 5, q4 = o.st / 4;
  const int sy = (int)(sv % o.nsy), sx = (int)((sv / o.nsy) % o.nsx), sz = (int)(sv / ((int64_t)o.nsy * o.nsx));
  *x = sx * o.st + w / q4 + (o.st / 2) * (j >> 2);
  *y = sy * o.st + w % q4 + q4 * (j & 3);
  *z = sz * 32 + (tid & 31);
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

// ---- pipelined cooperative-gather kernel (round 3; the product kernel) ------------------------------------------------
// Same lane <-> voxel / channel mapping and the same arithmetic as backproject_accum_coop_kernel (results bit-identical),
// but built for memory-level parallelism.  The round-2 loop put every gather inside its own conditional block, and the
// compiler closed each with `s_waitcnt vmcnt(0)`: ONE 1-KB gather in flight per wave, 24 KB per CU -- the kernel was bound
// by the latency of a single L2 / Infinity-Cache / HBM round trip (its run time scaled 1:1 with the number of resident
// waves), not by bandwidth.  Here all LPV gathers of a view are issued back to back (exec-masked loads,
// no branch in between), the projection of the NEXT view is computed while they fly, and only then are they added; with
// PIPE = 2 the gathers of view v + 1 are issued before the sums of view v.
//
// Schedule (SlabOrder + lockstep): the grid is cut into bricks; a brick is the unit one XCD group works on.  In lockstep
// mode the launch is a persistent grid of 8 x `bpc` workgroups (bpc = workgroups per brick, all resident): group g =
// blockIdx.x % 8 walks bricks g, g + 8, ... and meets at a group-local barrier after every brick, so that the workgroups
// sharing an L2 are in the same 32^3 box AND at about the same view at the same time -- the condition under which a
// pixel line fetched for one voxel is still in L2 when the next voxel on that camera ray asks for it
// (scripts/dense_l2sim.cpp: read hit rate 34 % -> 54 %, fabric reads 51 -> 35 GB per launch at the north-star shape).
// The barrier spins a bounded number of times: placement (b % 8 = XCD) and co-residency are performance assumptions,
// never correctness ones.
// Group-local barrier on ONE monotonic arrival counter (never reset, so a call that timed out leaves nothing to clean up):
// the k-th arrival waits until the counter reaches the next multiple of n.  Bounded spin: if the workgroups of a group
// are not all resident (another kernel holds CUs) the wait gives up after ~1 ms and the kernel merely loses its phase lock.
__device__ __forceinline__ void group_barrier(unsigned int* w, unsigned int n) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int goal = (old / n + 1u) * n;
    for (int spin = 0; spin < 1024; ++spin) {             // sparse polls: every poll is an uncached memory transaction that
      if ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - goal) >= 0) break;     // competes with the
      __builtin_amdgcn_s_sleep(64);                        // gathers of the workgroups still at work (~1.7 us apart)
    }
  }
  __syncthreads();
}

template <int LPV, int PIPE, int EPI>
__device__ __forceinline__ void accum_block(const DenseParams& p, const float* __restrict__ feat,
                                            const float* __restrict__ proj, float* __restrict__ volume,
                                            int32_t* __restrict__ count, int64_t lb, int c0, bool write_count,
                                            const SlabOrder& ord, float* __restrict__ lds_wave, int tid) {
  constexpr int VPG = 64 / LPV;
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  const int lane = tid & 63;
  const int64_t g = lb * 256 + tid;
  float wx = 0.f, wy = 0.f, wz = 0.f;
  bool in_grid;
  int64_t lin = -1;
  if (ord.on) {
    int x = 0, y = 0, z = 0;
    in_grid = ord.on == 2 ? lattice_decode(p, ord, lb, tid, &x, &y, &z) : slab_decode(p, ord, g, &x, &y, &z);
    if (in_grid) {
      lin = ((int64_t)x * p.Y + y) * p.Z + z;
      wx = (float)x * p.vs + p.ox; wy = (float)y * p.vs + p.oy; wz = (float)z * p.vs + p.oz;      // as voxel_world()
    }
  } else {
    in_grid = g < G;
    if (in_grid) { voxel_world(p, g, &wx, &wy, &wz); lin = g; }
  }
  if (__ballot(in_grid) == 0ull) return;
  float4 acc[LPV];
#pragma unroll
  for (int i = 0; i < LPV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int cnt = 0;
  const int sub = lane % LPV, vsel = lane / LPV;
  const int64_t plane = (int64_t)p.H * p.W * p.C;
  const float* fbase = feat + c0 + 4 * sub;

  auto pixel_of = [&](int v) -> int {
    float rx, ry;
    const bool ok = in_grid && project(proj + v * 12, wx, wy, wz, p.H, p.W, &rx, &ry);
    return ok ? ((int)ry * p.W + (int)rx) : -1;
  };
  auto gather = [&](int v, int pix, float4* q) {
    const float* fv = fbase + v * plane;
#pragma unroll
    for (int grp = 0; grp < LPV; ++grp) {
      const int pq = __shfl(pix, grp * VPG + vsel, 64);
      q[grp] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pq >= 0) q[grp] = *reinterpret_cast<const float4*>(fv + (int64_t)pq * p.C);
    }
  };
  auto add = [&](const float4* q) {
#pragma unroll
    for (int grp = 0; grp < LPV; ++grp) {
      acc[grp].x += q[grp].x; acc[grp].y += q[grp].y; acc[grp].z += q[grp].z; acc[grp].w += q[grp].w;
    }
  };

  if constexpr (PIPE == 1) {
    int pix_n = pixel_of(0);
    for (int v = 0; v < p.V; ++v) {
      const int pix = pix_n;
      const bool any = __ballot(pix >= 0) != 0ull;
      float4 q[LPV];
      if (any) gather(v, pix, q);
      cnt += pix >= 0 ? 1 : 0;
      pix_n = (v + 1 < p.V) ? pixel_of(v + 1) : -1;        // overlaps with the gathers in flight
      if (any) add(q);
    }
  } else {
    // two views in flight: q0 holds view v (issued one iteration ago), q1 is issued for view v + 1 before q0 is consumed
    float4 q0[LPV], q1[LPV];
    int pix0 = pixel_of(0);
    bool any0 = __ballot(pix0 >= 0) != 0ull;
    if (any0) gather(0, pix0, q0);
    cnt += pix0 >= 0 ? 1 : 0;
    for (int v = 0; v < p.V; ++v) {
      bool any1 = false;
      if (v + 1 < p.V) {
        const int pix1 = pixel_of(v + 1);
        any1 = __ballot(pix1 >= 0) != 0ull;
        if (any1) gather(v + 1, pix1, q1);
        cnt += pix1 >= 0 ? 1 : 0;
      }
      if (any0) add(q0);
      any0 = any1;
#pragma unroll
      for (int grp = 0; grp < LPV; ++grp) q0[grp] = q1[grp];
    }
  }

  // means.  The count of the voxel a lane accumulates for (group grp) lives in lane grp*VPG + vsel.
  if constexpr (EPI == 0) {
#pragma unroll
    for (int grp = 0; grp < LPV; ++grp) {
      const int cv = __shfl(cnt, grp * VPG + vsel, 64);
      const int lo = __shfl((int)(lin & 0xffffffffLL), grp * VPG + vsel, 64), hi = __shfl((int)(lin >> 32), grp * VPG + vsel, 64);
      const int64_t gv = ((int64_t)hi << 32) | (uint32_t)lo;
      if (gv >= 0) {
        const float denom = (float)cv;
        const float4 a = acc[grp];
        const int c = c0 + 4 * sub;
        const float m0 = cv > 0 ? a.x / denom : 0.0f, m1 = cv > 0 ? a.y / denom : 0.0f, m2 = cv > 0 ? a.z / denom : 0.0f,
                    m3 = cv > 0 ? a.w / denom : 0.0f;
        float* o = volume + (int64_t)c * G + gv;
        if (ord.nt) {                           // streaming stores: the volume is written once and not read here
          __builtin_nontemporal_store(m0, o); __builtin_nontemporal_store(m1, o + G);
          __builtin_nontemporal_store(m2, o + 2 * G); __builtin_nontemporal_store(m3, o + 3 * G);
        } else { o[0] = m0; o[G] = m1; o[2 * G] = m2; o[3 * G] = m3; }
      }
    }
  } else {
    // transpose through the wave's own LDS tile [4*LPV channels][64 voxels (+1 pad)] so that every store instruction
    // writes ONE channel plane for the wave's 64 voxels (two full 128-B lines in brick order) instead of 8 planes x 32 B
    constexpr int LD = 65;
    // everything the epilogue needs is re-derived here from a laundered thread index, so that nothing of it is live
    // across the view loop (kept live it pushed the loop over its 128 registers: scratch reloads = a vmcnt(0) per view)
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    const int lane2 = t2 & 63, sub2 = lane2 % LPV, vsel2 = lane2 / LPV;
    float* ldsw = lds_wave + 0 * (t2 & 0);
#pragma unroll
    for (int grp = 0; grp < LPV; ++grp) {
      const int cv = __shfl(cnt, grp * VPG + vsel2, 64);
      const float denom = (float)cv;
      const float4 a = acc[grp];
      float* t = ldsw + (4 * sub2) * LD + grp * VPG + vsel2;
      t[0 * LD] = cv > 0 ? a.x / denom : 0.0f;
      t[1 * LD] = cv > 0 ? a.y / denom : 0.0f;
      t[2 * LD] = cv > 0 ? a.z / denom : 0.0f;
      t[3 * LD] = cv > 0 ? a.w / denom : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    if (in_grid) {
      float* o = volume + (int64_t)c0 * G + lin;
#pragma unroll
      for (int c = 0; c < 4 * LPV; ++c) o[(int64_t)c * G] = ldsw[c * LD + lane2];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (write_count && in_grid) count[lin] = cnt;
}

template <int LPV, int PIPE, int EPI, int LOCK>
__global__ __launch_bounds__(256, PIPE == 1 ? 4 : 3) void backproject_accum_pipe_kernel(DenseParams p, const float* feat,
                                                                     const float* __restrict__ proj,
                                                                     float* __restrict__ volume,
                                                                     int32_t* __restrict__ count, int chunk_blocks,
                                                                     int64_t n_phys, SlabOrder ord,
                                                                     unsigned int* __restrict__ bar,
                                                                     const float* const* __restrict__ feat_ref) {
  // feat_ref != NULL: the feature maps are handed over BY REFERENCE -- a device word holds their address (one scalar load);
  // a captured launch sequence can then read whatever tensor the producer wrote, with no copy into a static buffer
  if (feat_ref != nullptr) feat = *feat_ref;
  __shared__ float lds[EPI ? 4 * (4 * LPV) * 65 : 1];
  float* lds_wave = lds + (EPI ? (threadIdx.x >> 6) * (4 * LPV) * 65 : 0);
  if constexpr (LOCK != 0) {
    // persistent grid: gridDim.x = 8 * chunk_blocks; group = blockIdx.x & 7 walks chunks group, group + 8, ...; all channel
    // sweeps inside (blockIdx.y unused)
    // ord.groups = 8: one brick per XCD at a time; 16: two independent groups per XCD (b and b + 8 share an XCD), each on its
    // own brick behind its own barrier -- while one group drains / decodes / projects, the other one's gathers keep the
    // memory pipeline busy
    const int NG = ord.groups;
    const int grp = blockIdx.x % NG, slot = blockIdx.x / NG;
    const int64_t n_chunks = n_phys / chunk_blocks;
    const int n_sweeps = (int)ceil_div_dev(p.C, 4 * LPV);
    for (int sw = 0; sw < n_sweeps; ++sw)
      for (int64_t ch = grp; ch < n_chunks; ch += NG) {
        // stagger: released together, the waves of a CU would project at the same time and then wait for memory at the
        // same time (lockstep 2x slower than free-running); wave w of every workgroup starts w x stagger x 64 cycles late,
        // so that at any moment a quarter of the waves is in each quarter of the view step
        for (int i = 0; i < ord.stagger * (int)(threadIdx.x >> 6); ++i) __builtin_amdgcn_s_sleep(1);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));            // per-thread values are recomputed per brick instead of being hoisted out of
                                                 // the brick loop (hoisted they spilled, and a scratch access in the view loop
                                                 // puts a vmcnt(0) in front of every gather)
        accum_block<LPV, PIPE, EPI>(p, feat, proj, volume, count, ch * chunk_blocks + slot, sw * 4 * LPV, sw == 0, ord, lds_wave, tid);
        group_barrier(bar + grp * 16, (unsigned)chunk_blocks);
      }
  } else {
    int64_t lb = blockIdx.x;
    int sweep = blockIdx.y;
    if (ord.own > 0) {
      // one channel sweep per XCD group (8 sweeps <-> 8 groups): every group walks ALL bricks in order for its own 128-byte
      // slice of the pixel rows, so the workgroups resident on an XCD are one contiguous run of neighbouring bricks (not
      // every 8th brick) and no feature line is fetched by two XCDs (dense_l2sim: read hit rate 34 % -> 41 %)
      sweep = (int)(lb & 7);
      lb >>= 3;
    } else if (chunk_blocks > 0) {
      const int64_t grp = lb & 7, k = lb >> 3;
      lb = (grp + 8 * (k / chunk_blocks)) * chunk_blocks + k % chunk_blocks;
    }
    accum_block<LPV, PIPE, EPI>(p, feat, proj, volume, count, lb, sweep * (4 * LPV), sweep == 0, ord, lds_wave, (int)threadIdx.x);
  }
}

#ifdef CNRMA_EXPERIMENTS
#include "dense_exp.inc"
#endif

// Debug / A-B switches of the dense kernel.  Product code never reads the environment: the defaults below ARE the shipped
// configuration, and only cnrma_debug_dense_tuning() (scripts/dense_ab.py, tests of the alternative orders) changes them.
struct DenseTune {
  int variant = 1;      // 0: round-2 kernel (one gather in flight per wave)   1: pipelined kernel   2: hoisted kernel
  int slab = 1;         // 1: brick order, 0: z-fastest linear order with x-plane chunks
  int st = 16, zt = 32, tt = 8, zi = 32;   // brick side (columns), brick layers, tile side, inner z run
  int chunk = -1;       // blocks per XCD chunk in linear order (-1: one x-plane, 0: plain round-robin)
  int persist = 0;      // variant 0 only: workgroups per CU of its persistent form
  int lpv = 0;          // 0: by channel count; 4 / 8 / 16 / 32 lanes (x 4 channels) per voxel
  int pipe = 1;         // variant 1: views in flight (1 | 2)
  int epi = 0;          // variant 1: 0 direct stores, 1 LDS-transposed full-line stores
  int nt = 0;           // variant 1: non-temporal stores of the volume
  int own = 0;          // variant 1: one channel sweep per XCD group (needs exactly 8 sweeps)
  int stagger = 0;      // variant 1 lockstep: start delay per wave index, in units of 64 cycles
  int ldspad = 0;       // variant 1: KB of unused dynamic LDS per workgroup (caps the workgroups per CU: leaves room for co-resident kernels)
  int groups = 8;       // variant 1 lockstep: 8 = one brick per XCD at a time, 16 = two independent half-size groups per XCD
  int lattice = 0;      // variant 1 lockstep: 1 = lattice assignment of columns to workgroups (balanced), 0 = compact tiles
  int lockstep = 0;     // variants 1, 2: persistent grid, every XCD group walks one brick at a time (1: behind a barrier, 2: no barrier)
};
#ifdef CNRMA_EXPERIMENTS
static DenseTune g_tune;                     // libcnrma_hip_exp.so only: written by cnrma_debug_dense_tuning
#define CNRMA_DENSE_TUNE g_tune
#else
// the product library has no tuning state: the defaults above ARE the shipped configuration (pipelined kernel, brick order)
static constexpr DenseTune k_dense_tune{};
#define CNRMA_DENSE_TUNE k_dense_tune
#endif

template <int LPV>
int launch_accum_coop(const DenseParams& p, const float* feat, const float* proj, float* volume, int32_t* count,
                      unsigned int* bar, hipStream_t st, const float* const* feat_ref = nullptr) {
  const DenseTune t = CNRMA_DENSE_TUNE;
  if (feat_ref != nullptr && t.variant != 1) return CNRMA_EINVAL;      // by-reference hand-off: the product kernel only
  const int64_t G = (int64_t)p.X * p.Y * p.Z;
  int64_t nb = ceil_div(G, 256);
  // one chunk = the blocks of one x-plane (at least 32: keeps a group's L2 working set a compact slab piece)
  int64_t cb = ceil_div((int64_t)p.Y * p.Z, 256);
  if (cb < 32) cb = 32;
  SlabOrder ord{0, 0, 0, 0, t.zt, t.st, t.tt, t.zi, t.nt, 0, t.stagger, t.groups == 16 ? 16 : 8};
  if (t.slab) {
    ord.on = 1;
    if (ord.zi < 1 || ord.zt % ord.zi != 0) return CNRMA_EINVAL;
    if (ord.zt < 1 || ord.tt < 1 || ord.st < ord.tt || ord.st % ord.tt != 0 || ((int64_t)ord.st * ord.st * ord.zt) % 256 != 0) return CNRMA_EINVAL;
    ord.nsx = (int)ceil_div(p.X, ord.st); ord.nsy = (int)ceil_div(p.Y, ord.st); ord.nsz = (int)ceil_div(p.Z, ord.zt);
    cb = (int64_t)ord.st * ord.st * ord.zt / 256;          // one brick per chunk
    nb = (int64_t)ord.nsx * ord.nsy * ord.nsz * cb;
  } else if (t.chunk >= 0) cb = t.chunk;
  int64_t gx = nb;
  if (cb > 0 && (nb >= 16 * cb || ord.on)) gx = ceil_div(ceil_div(nb, cb), 8) * 8 * cb;      // whole chunks for every XCD group
  else cb = 0;
#ifdef CNRMA_EXPERIMENTS
  if (t.variant == 0) {
    int64_t launch_x = gx;
    if (t.persist > 0 && gx > (int64_t)256 * t.persist) launch_x = (int64_t)256 * t.persist;     // a multiple of 8
    dim3 grid((unsigned)launch_x, (unsigned)ceil_div(p.C, 4 * LPV));
    hipLaunchKernelGGL((backproject_accum_coop_kernel<LPV>), grid, dim3(256), 0, st, p, feat, proj, volume, count, (int)cb, gx, ord);
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
  if (t.variant == 2 && p.V <= 64 && (int64_t)p.H * p.W * p.C * 4 < ((int64_t)1 << 31)) {
    const int lock = (t.lockstep && (bar != nullptr || t.lockstep == 2) && ord.on && ord.zt == 32 && ord.st % 8 == 0 && cb > 0 && cb <= 256) ? 1 : 0;
    const size_t lds = (size_t)4 * p.V * 64 * sizeof(int32_t);
    dim3 grid(lock ? (unsigned)(8 * cb) : (unsigned)gx);
    if (lock)
      hipLaunchKernelGGL((backproject_accum_hoist_kernel<LPV, 1>), grid, dim3(256), lds, st, p, feat, proj, volume, count, (int)cb,
                         gx, ord, bar, t.lockstep == 1 ? 1 : 0);
    else
      hipLaunchKernelGGL((backproject_accum_hoist_kernel<LPV, 0>), grid, dim3(256), lds, st, p, feat, proj, volume, count, (int)cb,
                         gx, ord, bar, 0);
    CNRMA_LAUNCH_CHECK();
    return 0;
  }
#endif
  const int lock = (t.lockstep == 1 && bar != nullptr && ord.on && cb > 0 && cb <= 256) ? 1 : 0;
  if (t.lattice && ord.on && ord.zt == 32 && ord.st % 8 == 0) ord.on = 2;
  const int n_sweeps = (int)ceil_div(p.C, 4 * LPV);
  if (t.own && !lock && ord.on && n_sweeps == 8 && nb * 8 < ((int64_t)1 << 31)) ord.own = 1;
  dim3 grid(lock ? (unsigned)(ord.groups * cb) : (ord.own ? (unsigned)(nb * 8) : (unsigned)gx), (lock || ord.own) ? 1u : (unsigned)n_sweeps);
#ifdef CNRMA_EXPERIMENTS
#define CNRMA_DENSE_LAUNCH(PIPE, EPI)                                                                                      \
  do {                                                                                                                     \
    if (lock)                                                                                                              \
      hipLaunchKernelGGL((backproject_accum_pipe_kernel<LPV, PIPE, EPI, 1>), grid, dim3(256), 0, st, p, feat, proj, volume, \
                         count, (int)cb, gx, ord, bar, feat_ref);                                                          \
    else                                                                                                                   \
      hipLaunchKernelGGL((backproject_accum_pipe_kernel<LPV, PIPE, EPI, 0>), grid, dim3(256), (size_t)t.ldspad * 1024, st, p,  \
                         feat, proj, volume, count, (int)cb, gx, ord, bar, feat_ref);                                     \
  } while (0)
  if (t.pipe == 2) { if (t.epi) CNRMA_DENSE_LAUNCH(2, 1); else CNRMA_DENSE_LAUNCH(2, 0); }
  else             { if (t.epi) CNRMA_DENSE_LAUNCH(1, 1); else CNRMA_DENSE_LAUNCH(1, 0); }
#undef CNRMA_DENSE_LAUNCH
#else                                   // the product schedule: one view in flight per wave, direct stores, free-running grid
  (void)lock;
  hipLaunchKernelGGL((backproject_accum_pipe_kernel<LPV, 1, 0, 0>), grid, dim3(256), 0, st, p, feat, proj, volume, count, (int)cb, gx,
                     ord, bar, feat_ref);
#endif
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

#ifdef CNRMA_EXPERIMENTS
extern "C" int cnrma_debug_dense_tuning(const int* v, int n) {
  // v = {variant, slab, st, zt, tt, zi, chunk, persist, lpv, pipe, epi, lockstep, lattice, nt, own, stagger, groups, ldspad}; n < 18 keeps the remaining defaults;
  // n == 0 restores the product configuration.  Host-side global state: debug / A-B runs only.
  DenseTune t;
  int* f[] = {&t.variant, &t.slab, &t.st, &t.zt, &t.tt, &t.zi, &t.chunk, &t.persist, &t.lpv, &t.pipe, &t.epi, &t.lockstep, &t.lattice, &t.nt, &t.own, &t.stagger, &t.groups, &t.ldspad};
  if (n < 0 || n > 18 || (n > 0 && v == nullptr)) return CNRMA_EINVAL;
  for (int i = 0; i < n; ++i) *f[i] = v[i];
  g_tune = t;
  return 0;
}
#endif

static int backproject_accum_any(const float* feat_nhwc, const float* const* feat_ref, const float* proj, int V, int C, int H,
                                 int W, int X, int Y, int Z, float voxel_size, float ox, float oy, float oz, float* volume,
                                 int32_t* count, void* workspace, int64_t workspace_bytes, void* stream) {
  if (V <= 0 || C <= 0 || H <= 0 || W <= 0 || X <= 0 || Y <= 0 || Z <= 0) return CNRMA_EINVAL;
  DenseParams p{V, C, H, W, X, Y, Z, voxel_size, ox, oy, oz};
  hipStream_t st = as_stream(stream);
  unsigned int* bar = (workspace != nullptr && workspace_bytes >= CNRMA_DENSE_WORKSPACE_BYTES) ? static_cast<unsigned int*>(workspace) : nullptr;
  const int l = CNRMA_DENSE_TUNE.lpv;
  if (l == 16 && C % 64 == 0) return launch_accum_coop<16>(p, feat_nhwc, proj, volume, count, bar, st, feat_ref);
  if (l == 4 && C % 16 == 0) return launch_accum_coop<4>(p, feat_nhwc, proj, volume, count, bar, st, feat_ref);
  if (C % 32 == 0) return launch_accum_coop<8>(p, feat_nhwc, proj, volume, count, bar, st, feat_ref);
  if (C % 16 == 0) return launch_accum_coop<4>(p, feat_nhwc, proj, volume, count, bar, st, feat_ref);
  if (C % 8 == 0) return launch_accum_coop<2>(p, feat_nhwc, proj, volume, count, bar, st, feat_ref);
  if (C % 4 == 0) return launch_accum_coop<1>(p, feat_nhwc, proj, volume, count, bar, st, feat_ref);
  if (feat_ref != nullptr) return CNRMA_EINVAL;
  return launch_accum<1>(p, feat_nhwc, proj, volume, count, st);
}

extern "C" int cnrma_backproject_accum_f32(const float* feat_nhwc, const float* proj, int V, int C, int H, int W,
                                           int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                           float* volume, int32_t* count, void* workspace, int64_t workspace_bytes,
                                           void* stream) {
  if (feat_nhwc == nullptr) return CNRMA_EINVAL;
  return backproject_accum_any(feat_nhwc, nullptr, proj, V, C, H, W, X, Y, Z, voxel_size, ox, oy, oz, volume, count, workspace,
                               workspace_bytes, stream);
}

extern "C" int cnrma_backproject_accum_ref_f32(const float* const* feat_nhwc_ref, const float* proj, int V, int C, int H, int W,
                                               int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                               float* volume, int32_t* count, void* workspace, int64_t workspace_bytes,
                                               void* stream) {
  if (feat_nhwc_ref == nullptr || C % 4 != 0) return CNRMA_EINVAL;
  return backproject_accum_any(nullptr, feat_nhwc_ref, proj, V, C, H, W, X, Y, Z, voxel_size, ox, oy, oz, volume, count,
                               workspace, workspace_bytes, stream);
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
