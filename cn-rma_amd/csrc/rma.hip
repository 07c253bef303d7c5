// Ray-marching aggregation (RMA) kernels: ray parameters, NeuS march (count + emit), depth variant.
//
// Arithmetic contract (bit-level, see DESIGN.md "numerics"): this file is compiled with -ffp-contract=off and
// every fused multiply-add below is an explicit fmaf(), because the reference's CPU path mixes MKL matrix
// products (FMA chains in k order) with un-fused elementwise torch ops.  sigmoid() is
// the exact operation sequence of torch's vectorised CPU kernel (Sleef expf_u10 with FMA, then 1/(1+e)), the
// running transmittance is an fp64 sequential product like the CPU cumprod (Appendix B.13).
//
// One lane per ray; all V views in one launch.  A ray marches only the step interval that can touch the grid
// (conservative slab clip; outside it every sample reads TSDF = 1 and contributes alpha = 0 exactly,
// Appendix B.4) and stops as soon as the transmittance drops below the weight threshold (w = T*alpha <= T).
#include "common.h"

#include <stdlib.h>

namespace {

struct MarchParams {
  int V, H, W, X, Y, Z, N;
  float vs, ox, oy, oz, t_one, thr;
};

struct Ray {
  float ox, oy, oz, dx, dy, dz;
};

// get_ray_parameter (ray_marching.py:71-111) for pixel (u = column, v = row) of one view.
__device__ __forceinline__ Ray make_ray(const float* __restrict__ Pi, float u, float v) {
  float far_[3], o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // bmm(Pinv, [u, v, 1, 1]) : fma chain in k order (MKL sgemm on this shape)
    float acc = Pi[r * 4 + 0] * u;
    acc = fmaf(Pi[r * 4 + 1], v, acc);
    acc = fmaf(Pi[r * 4 + 2], 1.0f, acc);
    acc = fmaf(Pi[r * 4 + 3], 1.0f, acc);
    far_[r] = acc;
    o[r] = Pi[r * 4 + 3];  // bmm(Pinv, [0,0,0,1]) is exactly column 3
  }
  float dx = far_[0] - o[0], dy = far_[1] - o[1], dz = far_[2] - o[2];
  // F.normalize(p=2, dim=1): sqrt(fma(z,z,fma(y,y,x*x))) clamped at eps = 1e-12, then true division
  float n2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  float nrm = fmaxf(sqrtf(n2), 1e-12f);
  Ray ray;
  ray.ox = o[0]; ray.oy = o[1]; ray.oz = o[2];
  ray.dx = dx / nrm; ray.dy = dy / nrm; ray.dz = dz / nrm;
  return ray;
}

// Sleef_expf_u10 (FMA flavour): the exp used by torch's vectorised CPU sigmoid.
__device__ __forceinline__ float sleef_expf_u10(float d) {
  float qf = rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
  int q = (int)qf;
  float s = fmaf(qf, -0.693145751953125f, d);
  s = fmaf(qf, -1.428606765330187045e-06f, s);
  float u = 0.000198527617612853646278381f;
  u = fmaf(u, s, 0.00139304355252534151077271f);
  u = fmaf(u, s, 0.00833336077630519866943359f);
  u = fmaf(u, s, 0.0416664853692054748535156f);
  u = fmaf(u, s, 0.166666671633720397949219f);
  u = fmaf(u, s, 0.5f);
  u = 1.0f + fmaf(s * s, u, s);
  int q1 = q >> 1, q2 = q - q1;
  u = u * __int_as_float((q1 + 127) << 23) * __int_as_float((q2 + 127) << 23);
  if (d < -104.0f) u = 0.0f;
  if (d > 104.0f) u = __int_as_float(0x7f800000);
  return u;
}

// torch.sigmoid(-sdf) on CPU: a = 0 - (-sdf); a = exp(a); a = 1 + a; a = 1 / a
__device__ __forceinline__ float sigmoid_neg(float sdf) {
  float a = 0.0f - (-sdf);
  a = sleef_expf_u10(a);
  a = 1.0f + a;
  return 1.0f / a;
}

struct Sample {
  float x, y, z, s;
  bool valid;
};

// one step of the march (ray_marching.py:713-745): place, rounded voxel, validity, sigmoid(-tsdf)
__device__ __forceinline__ Sample eval_step(const Ray& r, int n, const MarchParams& p, const float* __restrict__ tsdf) {
  Sample sm;
  float t = (float)n * p.t_one;
  sm.x = r.ox + r.dx * t;
  sm.y = r.oy + r.dy * t;
  sm.z = r.oz + r.dz * t;
  float fx = rintf((sm.x - p.ox) / p.vs);
  float fy = rintf((sm.y - p.oy) / p.vs);
  float fz = rintf((sm.z - p.oz) / p.vs);
  sm.valid = (fx >= 0.0f) && (fx < (float)p.X) && (fy >= 0.0f) && (fy < (float)p.Y) && (fz >= 0.0f) && (fz < (float)p.Z);
  float sdf = 1.0f;
  if (sm.valid) sdf = tsdf[((int64_t)(int)fx * p.Y + (int)fy) * p.Z + (int)fz];
  sm.s = sigmoid_neg(sdf);
  return sm;
}

// conservative step interval [a, b] outside which every sample is out of the grid; returns false when empty
__device__ __forceinline__ bool clip_steps(const Ray& r, const MarchParams& p, int* a, int* b) {
  float t0 = 0.0f, t1 = (float)(p.N - 1) * p.t_one;
  const float o[3] = {r.ox - p.ox, r.oy - p.oy, r.oz - p.oz};
  const float d[3] = {r.dx, r.dy, r.dz};
  const float hi[3] = {((float)p.X + 0.5f) * p.vs, ((float)p.Y + 0.5f) * p.vs, ((float)p.Z + 0.5f) * p.vs};
  const float lo = -1.5f * p.vs;
  bool any = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (!(fabsf(d[k]) > 1e-7f)) {
      if (!(o[k] >= lo && o[k] <= hi[k])) any = false;  // also rejects NaN
    } else {
      float ta = (lo - o[k]) / d[k], tb = (hi[k] - o[k]) / d[k];
      float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
      t0 = fmaxf(t0, tn);
      t1 = fminf(t1, tf);
    }
  }
  if (!any || !(t0 <= t1)) return false;
  int ia = (int)floorf(t0 / p.t_one) - 2;
  int ib = (int)ceilf(t1 / p.t_one) + 2;
  *a = ia < 0 ? 0 : ia;
  *b = ib > p.N - 1 ? p.N - 1 : ib;
  return *a <= *b;
}

// Generic NeuS march over one ray; calls keep(n, sample, w) for every kept sample in step order.
template <typename Keep>
__device__ __forceinline__ void neus_march(const Ray& r, const MarchParams& p, const float* __restrict__ tsdf,
                                           float s_out, Keep keep) {
  int a, b;
  if (!clip_steps(r, p, &a, &b)) return;
  double acc = 1.0;  // running product of (1 - alpha): fp64 like the CPU cumprod
  Sample cur = eval_step(r, a, p, tsdf);
  for (int n = a; n <= b; ++n) {
    Sample nxt;
    if (n + 1 <= b) {
      nxt = eval_step(r, n + 1, p, tsdf);
    } else {
      nxt = cur;                                  // n == N-1: s_next repeats the last sample (:758)
      if (n + 1 <= p.N - 1) { nxt.s = s_out; nxt.valid = false; }
    }
    float alpha = fmaxf((cur.s - nxt.s) / cur.s, 0.0f);          // :759
    float T = (float)acc;                                         // :760-762 exclusive product
    float w = T * alpha;                                          // :763
    if (cur.valid && w >= p.thr) keep(n, cur, w);                 // :765-767
    acc *= (double)(1.0f - alpha);
    if ((float)acc < p.thr) break;                                // every later w = T*alpha <= T < thr
    cur = nxt;
  }
}

__device__ __forceinline__ bool ray_setup(const MarchParams& p, const float* __restrict__ proj_inv, int64_t r, Ray* ray,
                                          int* view, int* pix) {
  const int64_t HW = (int64_t)p.H * p.W;
  if (r >= (int64_t)p.V * HW) return false;
  *view = (int)(r / HW);
  *pix = (int)(r - (int64_t)(*view) * HW);
  int v = *pix / p.W, u = *pix - v * p.W;
  *ray = make_ray(proj_inv + (int64_t)(*view) * 16, (float)u, (float)v);
  return true;
}

struct EmitDst {
  float* xyz; int xyz_stride;
  float* w; int w_stride;
  float* feat; int feat_stride;
  int32_t* sample;       // [M][2] = (ray, step) debug / alignment aid, may be NULL
  const int32_t* sel;    // may be NULL
  const float* w_div;    // device scalar, may be NULL
  float ax, ay, az;
  int64_t sel_cap = 0;   // > 0: rows m >= sel_cap have no selection entry (static trace: capacity of the row list) -> dropped
  int64_t out_cap = 0;   // > 0: output rows j >= out_cap do not exist -> dropped
};

// kept: optional per-ray record of the kept samples, [ray][cap] x {weight bits, step}; lets the emission run one
// thread group per OUTPUT row instead of re-marching every ray.  A ray can keep at most ~1/thr samples (the weights
// of a ray sum to 1 - prod(1-alpha) <= 1), so cap = floor(1/thr) + 2 never overflows; overflow[0] counts violations.
__global__ __launch_bounds__(256) void neus_count_kernel(MarchParams p, const float* __restrict__ proj_inv,
                                                         const float* __restrict__ tsdf, int32_t* __restrict__ count,
                                                         double* __restrict__ wsum, int2* __restrict__ kept, int cap,
                                                         int32_t* __restrict__ overflow) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  Ray ray; int view, pix;
  if (!ray_setup(p, proj_inv, r, &ray, &view, &pix)) return;
  const float s_out = sigmoid_neg(1.0f);
  int c = 0;
  double ws = 0.0;
  int2* mine = kept ? kept + r * cap : nullptr;
  neus_march(ray, p, tsdf, s_out, [&](int n, const Sample&, float w) {
    if (mine) {
      if (c < cap) mine[c] = make_int2(__float_as_int(w), n);
      else atomicAdd(overflow, 1);
    }
    ++c;
    ws += (double)w;
  });
  count[r] = c;
  wsum[r] = ws;
}

// ---- production march (single pass, kept-sample records) -----------------------------------------------------
// Same arithmetic as neus_march() above, organised for throughput (the reference path marches 12.3 M rays per scene at the
// north-star shape):
//  * sigmoid(-tsdf) comes from a per-voxel table built once per scene (cnrma_rma_sigmoid_table_f32: the SAME function
//    of the SAME value, so the weights are bit-identical) -- removes the exp and one division from every step;
//  * the three divisions by the voxel size use the correctly rounded reciprocal y = fl(1 / vs) and the quotient
//    refinement q = a*y; r = fma(-vs, q, a); q = fma(r, y, q); r = fma(-vs, q, a); q = fma(r, y, q) -- the core of the
//    IEEE division expansion without its reciprocal refinement and range scaling (5 instead of 11 instructions; equal to
//    a / vs for every sample: tests/test_rma_gpu.py::test_division_by_voxel_size_is_exact and the golden vectors);
//  * the table value of step n + 2 is fetched while step n is evaluated (the address only depends on n).
__device__ __forceinline__ float div_by_vs(float a, float vs, float y) {
  float q = a * y;
  float r = fmaf(-vs, q, a);
  q = fmaf(r, y, q);
  r = fmaf(-vs, q, a);
  return fmaf(r, y, q);
}

struct TabSample { float s; bool valid; int rad; };

// Free-space skipping (round 5).  skip[block] (one byte per 4 x 4 x 4 block of voxels, cnrma_rma_march_tables_f32) = R in {0, 4, 8,
// 12, 16}: every voxel within Chebyshev distance R of ANY voxel of the block holds the block's table value, bit for bit.  A
// sample whose successor has the same table value is a no-op of the march (alpha = 0, w = 0 < thr, transmittance x 1.0: the
// `still` test below), so a ray standing in a voxel of radius R may jump over the next j steps without evaluating them
// whenever they are certain to land inside that radius: the rounded voxel index moves by at most |delta| + 1 per axis for a
// displacement of delta voxels, the displacement of j steps is j * m with m = t_one * max|d_axis| / vs, so j < (R - 1) / m
// suffices (taken with a margin of 1 % + 0.01 voxel; the fp32 error of the positions is ~1e-5 voxel).  The records and sums
// are those of the step-by-step march, bit for bit (tests: golden vectors, oracle, skip on / off).
constexpr int SKIP_B = 4, SKIP_RMAX = 16;

__device__ __forceinline__ TabSample fetch_step(const Ray& r, int n, const MarchParams& p, float inv_vs,
                                                const float* __restrict__ tab, float s_out,
                                                const uint8_t* __restrict__ skip = nullptr, int by = 0, int bz = 0) {
  const float t = (float)n * p.t_one;
  const float x = r.ox + r.dx * t, y = r.oy + r.dy * t, z = r.oz + r.dz * t;
  const float fx = rintf(div_by_vs(x - p.ox, p.vs, inv_vs));
  const float fy = rintf(div_by_vs(y - p.oy, p.vs, inv_vs));
  const float fz = rintf(div_by_vs(z - p.oz, p.vs, inv_vs));
  TabSample o;
  o.valid = (fx >= 0.0f) && (fx < (float)p.X) && (fy >= 0.0f) && (fy < (float)p.Y) && (fz >= 0.0f) && (fz < (float)p.Z);
  o.s = s_out;
  o.rad = 0;
  if (o.valid) {
    o.s = tab[((int)fx * p.Y + (int)fy) * p.Z + (int)fz];                   // X*Y*Z < 2^31 (checked by the entry point)
    if (skip != nullptr) o.rad = skip[(((int)fx >> 2) * by + ((int)fy >> 2)) * bz + ((int)fz >> 2)];
  }
  return o;
}

__device__ __forceinline__ void neus_march_block(const MarchParams& p, const float* __restrict__ proj_inv,
                                                 const float* __restrict__ tab, int32_t* __restrict__ count,
                                                 double* __restrict__ wsum, int2* __restrict__ kept, int cap,
                                                 int32_t* __restrict__ overflow, int tiles_x, int tiles_per_view,
                                                 unsigned vblock, const uint8_t* __restrict__ skip = nullptr) {
  // thread -> ray: a wave marches an 8 x 8 pixel tile (a block a 16 x 16 tile) instead of 64 consecutive pixels of an image
  // row: the bundle stays a compact patch of voxels at every step, so one table line serves more lanes
  int64_t r;
  if (tiles_x > 0) {
    const int view_t = (int)(vblock / (unsigned)tiles_per_view), tile = (int)(vblock - (unsigned)view_t * (unsigned)tiles_per_view);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int u = tx * 16 + (w & 1) * 8 + (l & 7), v = ty * 16 + (w >> 1) * 8 + (l >> 3);
    if (u >= p.W || v >= p.H) return;
    r = ((int64_t)view_t * p.H + v) * p.W + u;
  } else {
    r = (int64_t)vblock * blockDim.x + threadIdx.x;
  }
  Ray ray; int view, pix;
  if (!ray_setup(p, proj_inv, r, &ray, &view, &pix)) return;
  const float s_out = sigmoid_neg(1.0f);
  const float inv_vs = 1.0f / p.vs;
  int c = 0;
  double ws = 0.0;
  int2* mine = kept + r * cap;
  int a, b;
  if (clip_steps(ray, p, &a, &b)) {
    double acc = 1.0;
    const int by = (p.Y + SKIP_B - 1) / SKIP_B, bz = (p.Z + SKIP_B - 1) / SKIP_B;
    // steps a jump may cover per unit of safe radius (see SKIP_B above): j <= (R - 1.01) / (1.01 m)
    const float m_vox = p.t_one * fmaxf(fmaxf(fabsf(ray.dx), fabsf(ray.dy)), fabsf(ray.dz)) * inv_vs * 1.01f;
    const float per_vox = m_vox > 1e-6f ? 1.0f / m_vox : 0.0f;
    TabSample cur = fetch_step(ray, a, p, inv_vs, tab, s_out, skip, by, bz);
    TabSample n1 = cur, n2 = cur;                       // samples a+1, a+2 (only used when inside [a, b])
    if (a + 1 <= b) n1 = fetch_step(ray, a + 1, p, inv_vs, tab, s_out, skip, by, bz);
    // a step whose successor has the same table value has alpha = max((s - s) / s, 0) = 0: w = 0 < thr (nothing kept) and
    // the transmittance is multiplied by 1.0 -- the whole body is a no-op (given 0 < thr <= 1, so that neither "w >= thr"
    // nor the exit test can change).  Rays spend most of their steps in free space (tsdf == -1: one table value); when every
    // lane of a wave is in such a step the division, the fp64 product and the bookkeeping are skipped (-25 of ~85 VALU
    // instructions per step; the kernel is VALU-issue bound).
    const bool can_skip = p.thr > 0.0f && p.thr <= 1.0f;
    for (int n = a; n <= b; ++n) {
      const bool still = can_skip && (n + 1 <= b) && (n1.s == cur.s);
      if (still && cur.rad > 0) {
        // every step up to n + j lands in a voxel holding cur.s: steps n .. n + j - 1 are no-ops; go on at n + j with the same
        // sample value (its own radius is not known: one plain step follows before the next jump)
        int j = (int)(((float)cur.rad - 1.01f) * per_vox);
        j = min(j, b - 2 - n);
        if (j >= 3) {
          n += j - 1;                                                           // the loop's ++n lands on n + j
          cur.rad = 0;
          n1 = fetch_step(ray, n + 2, p, inv_vs, tab, s_out, skip, by, bz);     // sample (n + j) + 1
          continue;
        }
      }
      if (n + 2 <= b) n2 = fetch_step(ray, n + 2, p, inv_vs, tab, s_out, skip, by, bz);     // in flight during this step
      if (!still) {
        float s_next;
        if (n + 1 <= b) s_next = n1.s;
        else s_next = (n + 1 <= p.N - 1) ? s_out : cur.s;                      // beyond the clip: outside / :758 repeat
        const float alpha = fmaxf((cur.s - s_next) / cur.s, 0.0f);            // :759
        const float T = (float)acc;                                            // :760-762
        const float w = T * alpha;                                             // :763
        if (cur.valid && w >= p.thr) {                                         // :765-767
          if (c < cap) mine[c] = make_int2(__float_as_int(w), n);
          else atomicAdd(overflow, 1);
          ++c;
          ws += (double)w;
        }
        acc *= (double)(1.0f - alpha);
        if ((float)acc < p.thr) break;
      }
      cur = n1;
      n1 = n2;
    }
  }
  count[r] = c;
  wsum[r] = ws;
}

__global__ __launch_bounds__(256) void neus_march_kernel(MarchParams p, const float* __restrict__ proj_inv,
                                                         const float* __restrict__ tab, int32_t* __restrict__ count,
                                                         double* __restrict__ wsum, int2* __restrict__ kept, int cap,
                                                         int32_t* __restrict__ overflow, int tiles_x, int tiles_per_view,
                                                         const uint8_t* __restrict__ skip) {
  neus_march_block(p, proj_inv, tab, count, wsum, kept, cap, overflow, tiles_x, tiles_per_view, blockIdx.x, skip);
}

// Layout pass + march in ONE launch.  The NCHW -> channels-last pass is a pure HBM stream (25 GB at the north-star shape);
// the march works on cache-resident data and does not read the feature maps at all.  As two launches they run one after
// the other (kernels that fill the chip do not overlap across streams here: 4.9 + 2.1 = 7.0 ms).  Here they are ONE grid
// whose every `stride`-th workgroup is a march block (stride odd: with an even stride all march blocks land on 2 of the
// 8 XCDs, blocks being dealt round-robin), so both kinds are co-resident on every CU: 6.66 ms at the north-star shape.
// Not more: both halves live off occupancy (the march waits on dependent table look-ups, the copy on HBM), so sharing
// the wave slots mostly time-slices them -- a persistent variant in which every workgroup alternated 16 layout tiles
// with one march tile (two atomic work counters) took 6.91 ms, the same as the two launches.  The two halves are
// independent, so the results are those of the two separate launches, bit for bit.
__global__ __launch_bounds__(256) void layout_march_kernel(MarchParams p, const float* __restrict__ proj_inv,
                                                           const float* __restrict__ tab, int32_t* __restrict__ count,
                                                           double* __restrict__ wsum, int2* __restrict__ kept, int cap,
                                                           int32_t* __restrict__ overflow, int tiles_x, int tiles_per_view,
                                                           unsigned n_march, unsigned stride,
                                                           const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int64_t HW, unsigned n_px_tiles, unsigned n_c_blocks,
                                                           const uint8_t* __restrict__ skip) {
  __shared__ float tile[64][65];
  const unsigned b = blockIdx.x;
  if (b % stride == stride - 1 && b / stride < n_march) {
    neus_march_block(p, proj_inv, tab, count, wsum, kept, cap, overflow, tiles_x, tiles_per_view, b / stride, skip);
    return;
  }
  const unsigned before = (b + 1) / stride < n_march ? (b + 1) / stride : n_march;       // march blocks at smaller indices
  const unsigned lb = b - before;
  // 64 channels x 64 pixels with 16-byte accesses on both sides (nchw_to_nhwc_v4_kernel of util.hip; HW % 4 == C % 4 == 0)
  const int64_t view = lb / (n_px_tiles * n_c_blocks);
  const unsigned rem = lb - (unsigned)view * (n_px_tiles * n_c_blocks);
  const int64_t p0 = (int64_t)(rem % n_px_tiles) * 64;
  const int c0 = (int)(rem / n_px_tiles) * 64;
  const float* sv = src + view * C * HW;
  float* dv = dst + view * C * HW;
  const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;     // 16 quads x 16 rows
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = rr + 16 * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + c < C && p0 + 4 * q < HW) v = *reinterpret_cast<const float4*>(sv + (int64_t)(c0 + c) * HW + p0 + 4 * q);
    tile[c][4 * q + 0] = v.x; tile[c][4 * q + 1] = v.y; tile[c][4 * q + 2] = v.z; tile[c][4 * q + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int px = rr + 16 * j;
    if (c0 + 4 * q < C && p0 + px < HW) {
      const float4 o = make_float4(tile[4 * q + 0][px], tile[4 * q + 1][px], tile[4 * q + 2][px], tile[4 * q + 3][px]);
      *reinterpret_cast<float4*>(dv + (p0 + px) * C + c0 + 4 * q) = o;
    }
  }
}

__global__ __launch_bounds__(256) void sigmoid_table_kernel(const float* __restrict__ tsdf, int64_t n, float* __restrict__ tab) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    tab[i] = sigmoid_neg(tsdf[i]);
}

// skip table, pass 1: blockval[b] = the table value shared by all voxels of the 4 x 4 x 4 block b (bit pattern), or 0xFFFFFFFF when
// they differ / the block sticks out of the grid (a sigmoid is never NaN).  One thread per block, 16-byte reads along z.
__global__ __launch_bounds__(256) void skip_blockval_kernel(const float* __restrict__ tab, int X, int Y, int Z, int bx, int by, int bz,
                                                            uint32_t* __restrict__ blockval) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (int64_t)bx * by * bz) return;
  const int kz = (int)(b % bz), ky = (int)((b / bz) % by), kx = (int)(b / ((int64_t)bz * by));
  const int x0 = kx * SKIP_B, y0 = ky * SKIP_B, z0 = kz * SKIP_B;
  uint32_t v = 0xFFFFFFFFu;
  if (x0 + SKIP_B <= X && y0 + SKIP_B <= Y && z0 + SKIP_B <= Z) {
    const uint32_t* t = reinterpret_cast<const uint32_t*>(tab);
    v = t[((int64_t)x0 * Y + y0) * Z + z0];
    bool same = true;
    for (int i = 0; i < SKIP_B; ++i)
      for (int j = 0; j < SKIP_B; ++j) {
        const uint32_t* q = t + ((int64_t)(x0 + i) * Y + (y0 + j)) * Z + z0;
#pragma unroll
        for (int k = 0; k < SKIP_B; ++k) same = same && (q[k] == v);
      }
    if (!same) v = 0xFFFFFFFFu;
  }
  blockval[b] = v;
}
// pass 2: skip[b] = 4 r for the largest r <= 4 such that every block within Chebyshev distance r of b exists and holds b's value
__global__ __launch_bounds__(256) void skip_radius_kernel(const uint32_t* __restrict__ blockval, int bx, int by, int bz,
                                                          uint8_t* __restrict__ skip) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (int64_t)bx * by * bz) return;
  const int kz = (int)(b % bz), ky = (int)((b / bz) % by), kx = (int)(b / ((int64_t)bz * by));
  const uint32_t v = blockval[b];
  int r = 0;
  if (v != 0xFFFFFFFFu) {
    constexpr int RB = SKIP_RMAX / SKIP_B;
    for (r = 0; r < RB; ++r) {                               // does the shell at distance r + 1 hold v everywhere?
      const int d = r + 1;
      if (kx - d < 0 || ky - d < 0 || kz - d < 0 || kx + d >= bx || ky + d >= by || kz + d >= bz) break;
      bool ok = true;
      for (int i = -d; i <= d && ok; ++i)
        for (int j = -d; j <= d && ok; ++j) {
          const bool face = i == -d || i == d || j == -d || j == d;
          const uint32_t* q = blockval + ((int64_t)(kx + i) * by + (ky + j)) * bz + kz;
          if (face) {
            for (int k = -d; k <= d; ++k) ok = ok && q[k] == v;
          } else {
            ok = q[-d] == v && q[d] == v;
          }
        }
      if (!ok) break;
    }
  }
  skip[b] = (uint8_t)(r * SKIP_B);
}

// Emission from the kept-sample records, two small kernels:
//  (1) one lane per ray copies the records of its SELECTED samples to their output position: rec[j] = {ray, step, w};
//  (2) LPR lanes per OUTPUT row j rebuild the place (o + d*t with the march's arithmetic) and copy the pixel's
//      channel vector (one full 128-B line read and written per row at C = 32).
__global__ __launch_bounds__(256) void neus_scatter_records_kernel(int64_t R, const int32_t* __restrict__ row_offset,
                                                                   const int2* __restrict__ kept, int cap,
                                                                   const int32_t* __restrict__ sel, int64_t sel_cap,
                                                                   int64_t rec_cap, int4* __restrict__ rec) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int64_t m0 = row_offset[r];
  const int c = (int)(row_offset[r + 1] - m0);     // 0 for rays of dropped views as well
  for (int i = 0; i < c; ++i) {
    if (sel && m0 + i >= sel_cap) break;            // rows past the capacity of the selection (flagged by the caller)
    const int64_t j = sel ? (int64_t)sel[m0 + i] : m0 + i;
    if (j < 0 || j >= rec_cap) continue;
    const int2 k = kept[r * cap + i];
    rec[j] = make_int4((int)r, k.y, k.x, 0);
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void neus_emit_rows_kernel(MarchParams p, int C, const float* __restrict__ proj_inv,
                                                            const float* feat, const float* const* __restrict__ feat_ref,
                                                            int64_t n_rows, const int32_t* __restrict__ n_rows_dev,
                                                            const int4* __restrict__ rec, EmitDst dst) {
  if (feat_ref != nullptr) feat = *feat_ref;         // feature maps handed over by reference (see cnrma_backproject_accum_ref_f32)
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j = t / LPR;
  const int sub = (int)(t % LPR);
  if (j >= live_rows(n_rows, n_rows_dev)) return;
  const int4 rc = rec[j];
  const int64_t r = rc.x;
  const int n = rc.y;
  const float w = __int_as_float(rc.z);
  const int view = (int)(r / ((int64_t)p.H * p.W));
  const int pix = (int)(r - (int64_t)view * p.H * p.W);
  if (sub == 0) {
    const int v = pix / p.W, u = pix - v * p.W;
    const Ray ray = make_ray(proj_inv + (int64_t)view * 16, (float)u, (float)v);
    const float tt = (float)n * p.t_one;
    if (dst.xyz) {
      float* q = dst.xyz + j * dst.xyz_stride;
      q[0] = (ray.ox + ray.dx * tt) + dst.ax;
      q[1] = (ray.oy + ray.dy * tt) + dst.ay;
      q[2] = (ray.oz + ray.dz * tt) + dst.az;
    }
    if (dst.w) dst.w[j * dst.w_stride] = w;
    if (dst.sample) { dst.sample[2 * j] = (int32_t)r; dst.sample[2 * j + 1] = n; }
  }
  if (dst.feat) {
    const float* f = feat + ((int64_t)view * p.H * p.W + pix) * C;
    float* q = dst.feat + j * dst.feat_stride;
    const bool scaled = dst.w_div != nullptr;
    const float scale = scaled ? w / dst.w_div[0] : 1.0f;
    const bool vec = ((C | dst.feat_stride) & 3) == 0 && ((((uintptr_t)dst.feat) | ((uintptr_t)feat)) & 15) == 0;
    if (vec) {
      for (int c = 4 * sub; c < C; c += 4 * LPR) {
        float4 x = *reinterpret_cast<const float4*>(f + c);
        if (scaled) { x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale; }
        *reinterpret_cast<float4*>(q + c) = x;
      }
    } else {
      for (int c = sub; c < C; c += LPR) q[c] = scaled ? f[c] * scale : f[c];
    }
  }
}

__device__ __forceinline__ void emit_row(const EmitDst& d, int64_t m, float x, float y, float z, float w,
                                         const float* __restrict__ f, int C, int64_t ray, int step) {
  int64_t j = m;
  if (d.sel) {
    if (d.sel_cap > 0 && m >= d.sel_cap) return;
    j = d.sel[m];
    if (j < 0) return;
  }
  if (d.out_cap > 0 && j >= d.out_cap) return;
  if (d.xyz) {
    float* q = d.xyz + j * d.xyz_stride;
    q[0] = x + d.ax; q[1] = y + d.ay; q[2] = z + d.az;
  }
  if (d.w) d.w[j * d.w_stride] = w;
  if (d.sample) { d.sample[2 * j] = (int32_t)ray; d.sample[2 * j + 1] = step; }
  if (d.feat) {
    float* q = d.feat + j * d.feat_stride;
    float scale = 1.0f;
    const bool scaled = d.w_div != nullptr;
    if (scaled) scale = w / d.w_div[0];                           // weights / mean(weights)  (:303)
    if (((C | d.feat_stride) & 3) == 0 && ((((uintptr_t)d.feat) | ((uintptr_t)f)) & 15) == 0) {
      for (int c = 0; c < C; c += 4) {
        float4 v = *reinterpret_cast<const float4*>(f + c);
        if (scaled) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }   // point_features * weights (:304)
        *reinterpret_cast<float4*>(q + c) = v;
      }
    } else {
      for (int c = 0; c < C; ++c) q[c] = scaled ? f[c] * scale : f[c];
    }
  }
}

__global__ __launch_bounds__(256) void neus_emit_kernel(MarchParams p, int C, const float* __restrict__ proj_inv,
                                                        const float* __restrict__ tsdf,
                                                        const float* __restrict__ feat,
                                                        const int32_t* __restrict__ row_offset, EmitDst dst) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  Ray ray; int view, pix;
  if (!ray_setup(p, proj_inv, r, &ray, &view, &pix)) return;
  const int64_t m0 = row_offset[r];
  if (row_offset[r + 1] == m0) return;  // nothing kept on this ray
  const float s_out = sigmoid_neg(1.0f);
  const float* f = feat + ((int64_t)view * p.H * p.W + pix) * C;
  int64_t m = m0;
  neus_march(ray, p, tsdf, s_out, [&](int n, const Sample& sm, float w) {
    emit_row(dst, m, sm.x, sm.y, sm.z, w, f, C, r, n);
    ++m;
  });
}

// ---- depth variant (ray_marching.py:809-956) -------------------------------------------------------------------
// first step n* with sdf[n]*sdf[n+1] <= 0 (the product with the appended 1 at n = N-1 is never <= 0)
__device__ __forceinline__ bool depth_first_change(const Ray& r, const MarchParams& p, const float* __restrict__ tsdf,
                                                   int* best) {
  int a, b;
  if (!clip_steps(r, p, &a, &b)) return false;
  auto sdf_at = [&](int n) -> float {
    float t = (float)n * p.t_one;
    float fx = rintf(((r.ox + r.dx * t) - p.ox) / p.vs);
    float fy = rintf(((r.oy + r.dy * t) - p.oy) / p.vs);
    float fz = rintf(((r.oz + r.dz * t) - p.oz) / p.vs);
    bool valid = (fx >= 0.0f) && (fx < (float)p.X) && (fy >= 0.0f) && (fy < (float)p.Y) && (fz >= 0.0f) && (fz < (float)p.Z);
    return valid ? tsdf[((int64_t)(int)fx * p.Y + (int)fy) * p.Z + (int)fz] : 1.0f;
  };
  float cur = sdf_at(a);
  for (int n = a; n <= b && n < p.N - 1; ++n) {
    float nxt = (n + 1 <= b) ? sdf_at(n + 1) : 1.0f;
    if (cur * nxt <= 0.0f) { *best = n; return true; }
    cur = nxt;
  }
  return false;
}

template <typename Slot>
__device__ __forceinline__ void depth_slots(const Ray& r, const MarchParams& p, int k, int best, Slot slot) {
  if (k == 0) {
    float idx = (float)best + 0.5f;                                                   // :912
    slot(r.ox + (r.dx * idx) * p.t_one, r.oy + (r.dy * idx) * p.t_one, r.oz + (r.dz * idx) * p.t_one, 1.0f, 0);
    return;
  }
  for (int j = 0; j < 2 * k; ++j) {
    int sel = best + (j - k + 1);                                                     // :889,:897
    float tri = (float)(j < k ? j + 1 : 2 * k - j) / (float)k;                        // :891-894
    if (sel < 0 || sel >= p.N) continue;                                              // :899-900 (weight 0 -> dropped)
    float idx = (float)sel;
    slot(r.ox + (r.dx * idx) * p.t_one, r.oy + (r.dy * idx) * p.t_one, r.oz + (r.dz * idx) * p.t_one, tri, j);
  }
}

__global__ __launch_bounds__(256) void depth_count_kernel(MarchParams p, int k, const float* __restrict__ proj_inv,
                                                          const float* __restrict__ tsdf, int32_t* __restrict__ count,
                                                          double* __restrict__ wsum) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  Ray ray; int view, pix;
  if (!ray_setup(p, proj_inv, r, &ray, &view, &pix)) return;
  int best, c = 0;
  double ws = 0.0;
  if (depth_first_change(ray, p, tsdf, &best))
    depth_slots(ray, p, k, best, [&](float, float, float, float w, int) { ++c; ws += (double)w; });
  count[r] = c;
  wsum[r] = ws;
}

__global__ __launch_bounds__(256) void depth_emit_kernel(MarchParams p, int k, int C, const float* __restrict__ proj_inv,
                                                         const float* __restrict__ tsdf,
                                                         const float* __restrict__ feat,
                                                         const int32_t* __restrict__ row_offset, EmitDst dst) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  Ray ray; int view, pix;
  if (!ray_setup(p, proj_inv, r, &ray, &view, &pix)) return;
  const int64_t m0 = row_offset[r];
  if (row_offset[r + 1] == m0) return;
  int best;
  if (!depth_first_change(ray, p, tsdf, &best)) return;
  const float* f = feat + ((int64_t)view * p.H * p.W + pix) * C;
  int64_t m = m0;
  depth_slots(ray, p, k, best, [&](float x, float y, float z, float w, int j) {
    emit_row(dst, m, x, y, z, w, f, C, r, j);
    ++m;
  });
}

__global__ void ray_params_kernel(const float* __restrict__ proj_inv, int V, int H, int W, float* __restrict__ o,
                                  float* __restrict__ d) {
  const int64_t HW = (int64_t)H * W;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (int64_t)V * HW) return;
  int view = (int)(r / HW);
  int pix = (int)(r - view * HW);
  int v = pix / W, u = pix - v * W;
  Ray ray = make_ray(proj_inv + (int64_t)view * 16, (float)u, (float)v);
  float* dv = d + (int64_t)view * 3 * HW;
  dv[pix] = ray.dx; dv[HW + pix] = ray.dy; dv[2 * HW + pix] = ray.dz;
  if (pix == 0) { o[view * 3] = ray.ox; o[view * 3 + 1] = ray.oy; o[view * 3 + 2] = ray.oz; }
}

__global__ void mean_weight_kernel(const double* __restrict__ wsum_total, const int32_t* __restrict__ m_total,
                                   float* __restrict__ mean_w) {
  // torch.mean(weights) (:303): fp64 sum of the fp32 weights / M, rounded once to fp32
  mean_w[0] = (float)(wsum_total[0] / (double)m_total[0]);
}

MarchParams make_params(int V, int H, int W, int X, int Y, int Z, float vs, float ox, float oy, float oz, int N,
                        float t_one, float thr) {
  MarchParams p;
  p.V = V; p.H = H; p.W = W; p.X = X; p.Y = Y; p.Z = Z; p.N = N;
  p.vs = vs; p.ox = ox; p.oy = oy; p.oz = oz; p.t_one = t_one; p.thr = thr;
  return p;
}

bool bad_dims(int V, int H, int W, int X, int Y, int Z, int N) {
  return V <= 0 || H <= 0 || W <= 0 || X <= 0 || Y <= 0 || Z <= 0 || N <= 0 ||
         (int64_t)V * H * W >= (int64_t)1 << 31;
}

}  // namespace

extern "C" int cnrma_ray_params_f32(const float* proj_inv, int V, int H, int W, float* o, float* d, void* stream) {
  if (V <= 0 || H <= 0 || W <= 0) return CNRMA_EINVAL;
  int64_t R = (int64_t)V * H * W;
  hipLaunchKernelGGL(ray_params_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), proj_inv,
                     V, H, W, o, d);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_neus_count_f32(const float* proj_inv, const float* tsdf, int V, int H, int W, int X, int Y,
                                        int Z, float voxel_size, float ox, float oy, float oz, int n_steps,
                                        float t_one, float thr, int32_t* count, double* wsum, void* stream) {
  if (bad_dims(V, H, W, X, Y, Z, n_steps)) return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, X, Y, Z, voxel_size, ox, oy, oz, n_steps, t_one, thr);
  int64_t R = (int64_t)V * H * W;
  hipLaunchKernelGGL(neus_count_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), p,
                     proj_inv, tsdf, count, wsum, (int2*)nullptr, 0, (int32_t*)nullptr);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

#ifdef CNRMA_EXPERIMENTS       // libcnrma_hip_exp.so only (tests/test_rma_gpu.py: the fast division against the compiler's IEEE expansion)
// parity aid: q_fast[i] = div_by_vs(a[i]) next to q_ref[i] = a[i] / vs (the compiler's IEEE expansion)
__global__ __launch_bounds__(256) void div_check_kernel(const float* __restrict__ a, int64_t n, float vs,
                                                        float* __restrict__ q_fast, float* __restrict__ q_ref) {
  const float y = 1.0f / vs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    q_fast[i] = div_by_vs(a[i], vs, y);
    q_ref[i] = a[i] / vs;
  }
}

extern "C" int cnrma_debug_div_by_voxel_size_f32(const float* a, int64_t n, float voxel_size, float* q_fast, float* q_ref,
                                                 void* stream) {
  if (n <= 0 || !(voxel_size > 0.0f)) return CNRMA_EINVAL;
  hipLaunchKernelGGL(div_check_kernel, dim3(4096), dim3(256), 0, as_stream(stream), a, n, voxel_size, q_fast, q_ref);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
#endif

extern "C" int cnrma_rma_sigmoid_table_f32(const float* tsdf, int64_t n, float* table, void* stream) {
  if (n <= 0 || tsdf == nullptr || table == nullptr) return CNRMA_EINVAL;
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sigmoid_table_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tsdf, n, table);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t cnrma_rma_skip_table_bytes(int X, int Y, int Z) {
  const size_t nb = (size_t)ceil_div(X, SKIP_B) * (size_t)ceil_div(Y, SKIP_B) * (size_t)ceil_div(Z, SKIP_B);
  return ((nb + 255) & ~(size_t)255) + nb * sizeof(uint32_t);               // radii, then the pass-1 scratch
}

extern "C" int cnrma_rma_march_tables_f32(const float* tsdf, int X, int Y, int Z, float* table, void* skip_table, void* stream) {
  if (tsdf == nullptr || table == nullptr || X <= 0 || Y <= 0 || Z <= 0) return CNRMA_EINVAL;
  const int64_t n = (int64_t)X * Y * Z;
  hipStream_t st = as_stream(stream);
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sigmoid_table_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tsdf, n, table);
  if (skip_table != nullptr) {
    const int bx = (int)ceil_div(X, SKIP_B), by = (int)ceil_div(Y, SKIP_B), bz = (int)ceil_div(Z, SKIP_B);
    const int64_t nb = (int64_t)bx * by * bz;
    uint8_t* skip = reinterpret_cast<uint8_t*>(skip_table);
    uint32_t* blockval = reinterpret_cast<uint32_t*>(skip + ((nb + 255) & ~(int64_t)255));
    hipLaunchKernelGGL(skip_blockval_kernel, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, st, table, X, Y, Z, bx, by, bz, blockval);
    hipLaunchKernelGGL(skip_radius_kernel, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, st, blockval, bx, by, bz, skip);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

static int launch_march(const MarchParams& p, const float* proj_inv, const float* tsdf, const float* sig_table,
                        int32_t* count, double* wsum, void* kept, int cap, int32_t* overflow, const float* feat_nchw,
                        float* feat_nhwc, int C, hipStream_t st, const uint8_t* skip = nullptr) {
  const int64_t R = (int64_t)p.V * p.H * p.W, HW = (int64_t)p.H * p.W;
  hipError_t e = cnrma_fill_bytes(overflow, 0, 4 * sizeof(int32_t), st);     // [0] violations, [1..3] reserved
  if (e != hipSuccess) return -(int)e;
  const bool table = sig_table != nullptr && (int64_t)p.X * p.Y * p.Z < ((int64_t)1 << 31);
  const int tx = (int)ceil_div(p.W, 16), ty = (int)ceil_div(p.H, 16);
  const int64_t n_march = (int64_t)p.V * tx * ty;
  if (feat_nchw != nullptr) {
    const int64_t npx = ceil_div(HW, 64), ncb = ceil_div(C, 64), n_layout = npx * ncb * p.V;
    const bool v4 = HW % 4 == 0 && C % 4 == 0 && ((((uintptr_t)feat_nchw) | ((uintptr_t)feat_nhwc)) & 15) == 0;
    // worth it when the layout pass is the larger half (at the ScanNet shape it is 0.04 of 0.28 ms: two launches)
    if (table && v4 && n_layout + n_march < ((int64_t)1 << 31) && n_layout >= 8 * n_march) {
      unsigned stride = (unsigned)((n_layout + n_march) / n_march);
      if (stride % 2 == 0) --stride;                                 // odd: march blocks on all 8 XCDs
      hipLaunchKernelGGL(layout_march_kernel, dim3((unsigned)(n_layout + n_march)), dim3(256), 0, st, p, proj_inv, sig_table,
                         count, wsum, reinterpret_cast<int2*>(kept), cap, overflow, tx, tx * ty, (unsigned)n_march, stride,
                         feat_nchw, feat_nhwc, C, HW, (unsigned)npx, (unsigned)ncb, skip);
      CNRMA_LAUNCH_CHECK();
      return 0;
    }
    const int rc = cnrma_nchw_to_nhwc_f32(feat_nchw, feat_nhwc, p.V, C, p.H, p.W, st);      // not fusable: two launches
    if (rc != 0) return rc;
  }
  if (table) {
    hipLaunchKernelGGL(neus_march_kernel, dim3((unsigned)n_march), dim3(256), 0, st, p, proj_inv, sig_table, count, wsum,
                       reinterpret_cast<int2*>(kept), cap, overflow, tx, tx * ty, skip);
  } else {
    if (tsdf == nullptr) return CNRMA_EINVAL;
    hipLaunchKernelGGL(neus_count_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, st, p, proj_inv, tsdf, count, wsum,
                       reinterpret_cast<int2*>(kept), cap, overflow);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_neus_march_f32(const float* proj_inv, const float* tsdf, const float* sig_table, int V, int H,
                                        int W, int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                        int n_steps, float t_one, float thr, int32_t* count, double* wsum, void* kept,
                                        int cap, int32_t* overflow, const void* skip_table, void* stream) {
  if (bad_dims(V, H, W, X, Y, Z, n_steps) || kept == nullptr || cap <= 0 || overflow == nullptr) return CNRMA_EINVAL;
  if (tsdf == nullptr && sig_table == nullptr) return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, X, Y, Z, voxel_size, ox, oy, oz, n_steps, t_one, thr);
  return launch_march(p, proj_inv, tsdf, sig_table, count, wsum, kept, cap, overflow, nullptr, nullptr, 0, as_stream(stream),
                      sig_table ? reinterpret_cast<const uint8_t*>(skip_table) : nullptr);
}

extern "C" int cnrma_nchw_to_nhwc_march_f32(const float* feat_nchw, float* feat_nhwc, int C, const float* proj_inv,
                                            const float* tsdf, const float* sig_table, int V, int H, int W, int X, int Y,
                                            int Z, float voxel_size, float ox, float oy, float oz, int n_steps, float t_one,
                                            float thr, int32_t* count, double* wsum, void* kept, int cap, int32_t* overflow,
                                            const void* skip_table, void* stream) {
  if (bad_dims(V, H, W, X, Y, Z, n_steps) || kept == nullptr || cap <= 0 || overflow == nullptr || C <= 0 ||
      feat_nchw == nullptr || feat_nhwc == nullptr)
    return CNRMA_EINVAL;
  if (tsdf == nullptr && sig_table == nullptr) return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, X, Y, Z, voxel_size, ox, oy, oz, n_steps, t_one, thr);
  return launch_march(p, proj_inv, tsdf, sig_table, count, wsum, kept, cap, overflow, feat_nchw, feat_nhwc, C, as_stream(stream),
                      sig_table ? reinterpret_cast<const uint8_t*>(skip_table) : nullptr);
}

static int neus_emit_rows_any(const float* proj_inv, const float* feat_nhwc, const float* const* feat_ref, int V, int C, int H, int W,
                                            int n_steps, float t_one, const int32_t* row_offset, int64_t n_out,
                                            const int32_t* n_out_dev, const void* kept, int cap,
                                            const int32_t* sel_index, int64_t sel_cap, void* records,
                                            const float* w_div, float addx, float addy, float addz, float* out_xyz,
                                            int xyz_stride, float* out_w, int w_stride, float* out_feat,
                                            int feat_stride, int32_t* out_sample, void* stream) {
  // kept == nullptr: the records are already there (cnrma_rma_select_records) -- only the row emission runs
  if (V <= 0 || C <= 0 || H <= 0 || W <= 0 || n_out <= 0 || (kept != nullptr && cap <= 0) || records == nullptr ||
      (kept == nullptr && row_offset != nullptr) || (feat_nhwc == nullptr && feat_ref == nullptr && out_feat != nullptr))
    return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, 1, 1, 1, 1.0f, 0.f, 0.f, 0.f, n_steps, t_one, 0.0f);
  EmitDst d{out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride, out_sample, nullptr, w_div, addx, addy, addz};
  const int64_t R = (int64_t)V * H * W;
  hipStream_t st = as_stream(stream);
  int4* rec = reinterpret_cast<int4*>(records);
  if (kept != nullptr)
    hipLaunchKernelGGL(neus_scatter_records_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, st, R, row_offset,
                       reinterpret_cast<const int2*>(kept), cap, sel_index, sel_cap, n_out, rec);
  if (C % 256 == 0) {      // one wave copies a row's 1-KiB channel vector per instruction
    hipLaunchKernelGGL((neus_emit_rows_kernel<64>), dim3((unsigned)ceil_div(n_out * 64, 256)), dim3(256), 0, st, p, C,
                       proj_inv, feat_nhwc, feat_ref, n_out, n_out_dev, rec, d);
  } else if (C % 32 == 0) {
    hipLaunchKernelGGL((neus_emit_rows_kernel<8>), dim3((unsigned)ceil_div(n_out * 8, 256)), dim3(256), 0, st, p, C,
                       proj_inv, feat_nhwc, feat_ref, n_out, n_out_dev, rec, d);
  } else {
    hipLaunchKernelGGL((neus_emit_rows_kernel<2>), dim3((unsigned)ceil_div(n_out * 2, 256)), dim3(256), 0, st, p, C,
                       proj_inv, feat_nhwc, feat_ref, n_out, n_out_dev, rec, d);
  }
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_neus_emit_rows_f32(const float* proj_inv, const float* feat_nhwc, int V, int C, int H, int W,
                                            int n_steps, float t_one, const int32_t* row_offset, int64_t n_out,
                                            const int32_t* n_out_dev, const void* kept, int cap,
                                            const int32_t* sel_index, int64_t sel_cap, void* records,
                                            const float* w_div, float addx, float addy, float addz, float* out_xyz,
                                            int xyz_stride, float* out_w, int w_stride, float* out_feat,
                                            int feat_stride, int32_t* out_sample, void* stream) {
  return neus_emit_rows_any(proj_inv, feat_nhwc, nullptr, V, C, H, W, n_steps, t_one, row_offset, n_out, n_out_dev, kept, cap,
                            sel_index, sel_cap, records, w_div, addx, addy, addz, out_xyz, xyz_stride, out_w, w_stride, out_feat,
                            feat_stride, out_sample, stream);
}

extern "C" int cnrma_rma_neus_emit_rows_ref_f32(const float* proj_inv, const float* const* feat_nhwc_ref, int V, int C, int H,
                                                int W, int n_steps, float t_one, const int32_t* row_offset, int64_t n_out,
                                                const int32_t* n_out_dev, const void* kept, int cap,
                                                const int32_t* sel_index, int64_t sel_cap, void* records,
                                                const float* w_div, float addx, float addy, float addz, float* out_xyz,
                                                int xyz_stride, float* out_w, int w_stride, float* out_feat,
                                                int feat_stride, int32_t* out_sample, void* stream) {
  if (feat_nhwc_ref == nullptr) return CNRMA_EINVAL;
  return neus_emit_rows_any(proj_inv, nullptr, feat_nhwc_ref, V, C, H, W, n_steps, t_one, row_offset, n_out, n_out_dev, kept, cap,
                            sel_index, sel_cap, records, w_div, addx, addy, addz, out_xyz, xyz_stride, out_w, w_stride, out_feat,
                            feat_stride, out_sample, stream);
}

// Backward of the emission w.r.t. the feature maps (the only differentiable input: the weights are computed under
// torch.no_grad() in the reference, ray_marching.py:705): out[j][c] = feat[pixel(r)][c] * (w_j / w_div), so
// grad_feat[pixel(r)][c] = sum over the SELECTED rows j of ray r of grad_out[j][c] * (w_j / w_div).  The rows of a ray
// are known from its kept-sample records: one lane group per ray walks them in step order -- no atomics, deterministic.
template <int LPR>
__global__ __launch_bounds__(256) void neus_rows_backward_kernel(int64_t R, int C, const int32_t* __restrict__ row_offset,
                                                                const int2* __restrict__ kept, int cap,
                                                                const int32_t* __restrict__ sel,
                                                                const float* __restrict__ w_div,
                                                                const float* __restrict__ grad_out, int grad_stride,
                                                                float* __restrict__ grad_feat) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = t / LPR;
  const int sub = (int)(t % LPR);
  if (r >= R) return;
  const int64_t m0 = row_offset[r];
  const int cnt = (int)(row_offset[r + 1] - m0);
  const float wd = w_div ? w_div[0] : 1.0f;
  for (int c = sub; c < C; c += LPR) {
    float acc = 0.0f;
    for (int i = 0; i < cnt; ++i) {
      const int64_t j = sel ? (int64_t)sel[m0 + i] : m0 + i;
      if (j < 0) continue;
      const float w = __int_as_float(kept[r * cap + i].x);
      const float scale = w_div ? w / wd : 1.0f;           // the forward's factor, same operations
      acc += grad_out[j * grad_stride + c] * scale;
    }
    grad_feat[r * C + c] = acc;
  }
}

extern "C" int cnrma_rma_neus_rows_backward_f32(const float* grad_out_feat, int grad_stride, int V, int C, int H, int W,
                                                const int32_t* row_offset, const void* kept, int cap,
                                                const int32_t* sel_index, const float* w_div, float* grad_feat_nhwc,
                                                void* stream) {
  if (V <= 0 || C <= 0 || H <= 0 || W <= 0 || kept == nullptr || cap <= 0 || grad_out_feat == nullptr ||
      grad_feat_nhwc == nullptr)
    return CNRMA_EINVAL;
  const int64_t R = (int64_t)V * H * W;
  const int lpr = C >= 32 ? 32 : (C >= 8 ? 8 : 1);
  const unsigned blocks = (unsigned)ceil_div(R * lpr, 256);
  hipStream_t st = as_stream(stream);
  const int2* k2 = reinterpret_cast<const int2*>(kept);
  if (lpr == 32)
    hipLaunchKernelGGL((neus_rows_backward_kernel<32>), dim3(blocks), dim3(256), 0, st, R, C, row_offset, k2, cap, sel_index,
                       w_div, grad_out_feat, grad_stride, grad_feat_nhwc);
  else if (lpr == 8)
    hipLaunchKernelGGL((neus_rows_backward_kernel<8>), dim3(blocks), dim3(256), 0, st, R, C, row_offset, k2, cap, sel_index,
                       w_div, grad_out_feat, grad_stride, grad_feat_nhwc);
  else
    hipLaunchKernelGGL((neus_rows_backward_kernel<1>), dim3(blocks), dim3(256), 0, st, R, C, row_offset, k2, cap, sel_index,
                       w_div, grad_out_feat, grad_stride, grad_feat_nhwc);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_neus_emit_f32(const float* proj_inv, const float* tsdf, const float* feat_nhwc, int V, int C,
                                       int H, int W, int X, int Y, int Z, float voxel_size, float ox, float oy,
                                       float oz, int n_steps, float t_one, float thr, const int32_t* row_offset,
                                       const int32_t* sel_index, const float* w_div, float addx, float addy,
                                       float addz, float* out_xyz, int xyz_stride, float* out_w, int w_stride,
                                       float* out_feat, int feat_stride, int32_t* out_sample, void* stream) {
  if (bad_dims(V, H, W, X, Y, Z, n_steps) || C <= 0) return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, X, Y, Z, voxel_size, ox, oy, oz, n_steps, t_one, thr);
  EmitDst d{out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride, out_sample, sel_index, w_div, addx, addy, addz};
  int64_t R = (int64_t)V * H * W;
  hipLaunchKernelGGL(neus_emit_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), p, C,
                     proj_inv, tsdf, feat_nhwc, row_offset, d);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_depth_count_f32(const float* proj_inv, const float* tsdf, int V, int H, int W, int X, int Y,
                                         int Z, float voxel_size, float ox, float oy, float oz, int n_steps,
                                         float t_one, int select_grids, int32_t* count, double* wsum, void* stream) {
  if (bad_dims(V, H, W, X, Y, Z, n_steps) || select_grids < 0) return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, X, Y, Z, voxel_size, ox, oy, oz, n_steps, t_one, 0.0f);
  int64_t R = (int64_t)V * H * W;
  hipLaunchKernelGGL(depth_count_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), p,
                     select_grids, proj_inv, tsdf, count, wsum);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_depth_emit_f32(const float* proj_inv, const float* tsdf, const float* feat_nhwc, int V, int C,
                                        int H, int W, int X, int Y, int Z, float voxel_size, float ox, float oy,
                                        float oz, int n_steps, float t_one, int select_grids,
                                        const int32_t* row_offset, const int32_t* sel_index, const float* w_div,
                                        float addx, float addy, float addz, float* out_xyz, int xyz_stride,
                                        float* out_w, int w_stride, float* out_feat, int feat_stride, int64_t sel_cap,
                                        int64_t out_cap, void* stream) {
  if (bad_dims(V, H, W, X, Y, Z, n_steps) || C <= 0 || select_grids < 0 || sel_cap < 0 || out_cap < 0) return CNRMA_EINVAL;
  MarchParams p = make_params(V, H, W, X, Y, Z, voxel_size, ox, oy, oz, n_steps, t_one, 0.0f);
  EmitDst d{out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride, nullptr, sel_index, w_div, addx, addy, addz};
  d.sel_cap = sel_cap;
  d.out_cap = out_cap;
  int64_t R = (int64_t)V * H * W;
  hipLaunchKernelGGL(depth_emit_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, as_stream(stream), p,
                     select_grids, C, proj_inv, tsdf, feat_nhwc, row_offset, d);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

// Reference quirk (ray_marching.py:781-782, :282-287): a view that keeps exactly ONE sample is dropped entirely
// (torch.squeeze() makes the index 0-dim, len() raises and the bare except skips the view).  One block per view.
__global__ __launch_bounds__(1024) void drop_single_sample_views_kernel(int32_t* __restrict__ count, double* __restrict__ wsum,
                                                                        int64_t rays_per_view) {
  __shared__ int smem[1024 / 64 + 1];
  int32_t* c = count + (int64_t)blockIdx.x * rays_per_view;
  double* w = wsum + (int64_t)blockIdx.x * rays_per_view;
  int local = 0;
  int64_t i = threadIdx.x * 4;
  if ((rays_per_view & 3) == 0 && ((reinterpret_cast<uintptr_t>(c) & 15) == 0)) {
    for (; i + 3 < rays_per_view; i += 4096) {
      const int4 q = *reinterpret_cast<const int4*>(c + i);
      local += q.x + q.y + q.z + q.w;
    }
  } else {
    for (int64_t j = threadIdx.x; j < rays_per_view; j += 1024) local += c[j];
  }
  int total;
  (void)block_excl_scan<1024>(local, smem, &total);
  if (total != 1) return;
  for (int64_t j = threadIdx.x; j < rays_per_view; j += 1024) { c[j] = 0; w[j] = 0.0; }
}

extern "C" int cnrma_rma_drop_single_sample_views(int32_t* count, double* wsum, int V, int64_t rays_per_view, void* stream) {
  if (V <= 0 || rays_per_view <= 0 || count == nullptr || wsum == nullptr) return CNRMA_EINVAL;
  hipLaunchKernelGGL(drop_single_sample_views_kernel, dim3((unsigned)V), dim3(1024), 0, as_stream(stream), count, wsum,
                     rays_per_view);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_rma_mean_weight(const double* wsum_total, const int32_t* m_total, float* mean_w, void* stream) {
  hipLaunchKernelGGL(mean_weight_kernel, dim3(1), dim3(1), 0, as_stream(stream), wsum_total, m_total, mean_w);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
