// 3D NMS for the offline post-processing step: BEV-overlap NMS as used by the reference's
// post_process/nms_bbox.py:17-58 through mmdet3d's pcdet_nms_gpu (rotated boxes) / pcdet_nms_normal_gpu (axis-aligned).
// Third-party semantics (OpenPCDet iou3d_nms, not under /root/reference): boxes are (x, y, z, dx, dy, dz, heading),
// the overlap is the 2D IoU of the bird's-eye-view rectangles, candidates are visited in descending score order and a
// candidate is dropped when its IoU with an already kept box exceeds the threshold.
//
// Kernel: one 64 x 64 tile of the pairwise suppression matrix per block -> bit mask [N][ceil(N/64)] (uint64);
// the greedy pass over the mask is a sequential scan done by the host wrapper (post-processing, not the hot path).
#include "common.h"

namespace {

struct P2 { float x, y; };

__device__ __forceinline__ float cross2(P2 a, P2 b, P2 c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

// corners of a BEV rectangle (centre cx,cy, size dx,dy, heading a), counter-clockwise
__device__ __forceinline__ void bev_corners(const float* b, P2* c) {
  const float ca = cosf(b[6]), sa = sinf(b[6]);
  const float hx = b[3] * 0.5f, hy = b[4] * 0.5f;
  const float lx[4] = {hx, -hx, -hx, hx}, ly[4] = {hy, hy, -hy, -hy};
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i].x = b[0] + lx[i] * ca - ly[i] * sa; c[i].y = b[1] + lx[i] * sa + ly[i] * ca; }
}

// area of the intersection of two convex quadrilaterals (Sutherland-Hodgman clipping of A by the edges of B)
__device__ float quad_intersection_area(const P2* A, const P2* B) {
  P2 poly[10], tmp[10];
  int n = 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) poly[i] = A[i];
  for (int e = 0; e < 4 && n > 0; ++e) {
    const P2 p = B[e], q = B[(e + 1) & 3];
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const P2 s = poly[i], t = poly[(i + 1) % n];
      const float ds = cross2(p, q, s), dt = cross2(p, q, t);
      if (ds >= 0.0f) tmp[m++] = s;
      if ((ds > 0.0f && dt < 0.0f) || (ds < 0.0f && dt > 0.0f)) {
        const float u = ds / (ds - dt);
        tmp[m].x = s.x + u * (t.x - s.x);
        tmp[m].y = s.y + u * (t.y - s.y);
        ++m;
      }
    }
    n = m;
    for (int i = 0; i < n; ++i) poly[i] = tmp[i];
  }
  float area = 0.0f;
  for (int i = 0; i < n; ++i) { const P2 a = poly[i], b = poly[(i + 1) % n]; area += a.x * b.y - a.y * b.x; }
  return fabsf(area) * 0.5f;
}

__device__ __forceinline__ float bev_iou(const float* a, const float* b, int rotated) {
  float inter;
  if (rotated) {
    P2 ca[4], cb[4];
    bev_corners(a, ca);
    bev_corners(b, cb);
    inter = quad_intersection_area(ca, cb);
  } else {
    const float lx = fmaxf(a[0] - a[3] * 0.5f, b[0] - b[3] * 0.5f), rx = fminf(a[0] + a[3] * 0.5f, b[0] + b[3] * 0.5f);
    const float ly = fmaxf(a[1] - a[4] * 0.5f, b[1] - b[4] * 0.5f), ry = fminf(a[1] + a[4] * 0.5f, b[1] + b[4] * 0.5f);
    inter = fmaxf(rx - lx, 0.0f) * fmaxf(ry - ly, 0.0f);
  }
  const float uni = a[3] * a[4] + b[3] * b[4] - inter;
  return inter / fmaxf(uni, 1e-8f);
}

// boxes [N][7] sorted by descending score; mask[i][j/64] bit (j%64) set when j > i and IoU(i, j) > thr
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thr, int rotated,
                                                      unsigned long long* __restrict__ mask, int words) {
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  if (col0 + 63 < row0) return;                      // strictly below the diagonal: never needed
  __shared__ float cb[64 * 7];
  const int tid = threadIdx.x;
  if (col0 + tid < n)
    for (int k = 0; k < 7; ++k) cb[tid * 7 + k] = boxes[(int64_t)(col0 + tid) * 7 + k];
  __syncthreads();
  const int i = row0 + tid;
  if (i >= n) return;
  float a[7];
  for (int k = 0; k < 7; ++k) a[k] = boxes[(int64_t)i * 7 + k];
  unsigned long long bits = 0ull;
  const int lim = min(64, n - col0);
  for (int j = 0; j < lim; ++j) {
    if (col0 + j <= i) continue;
    if (bev_iou(a, cb + j * 7, rotated) > thr) bits |= 1ull << j;
  }
  mask[(int64_t)i * words + blockIdx.x] = bits;
}

// pairwise IoU matrix (evaluation / tests): iou[i][j] for a [Na][7], b [Nb][7]; mode 0 = BEV, 1 = 3D (BEV x height)
__global__ __launch_bounds__(256) void iou_matrix_kernel(const float* __restrict__ a, int na, const float* __restrict__ b,
                                                         int nb, int rotated, int mode3d, float* __restrict__ iou) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)na * nb) return;
  const int i = (int)(t / nb), j = (int)(t - (int64_t)i * nb);
  const float* p = a + (int64_t)i * 7;
  const float* q = b + (int64_t)j * 7;
  if (!mode3d) { iou[t] = bev_iou(p, q, rotated); return; }
  float inter;
  if (rotated) {
    P2 ca[4], cb[4];
    bev_corners(p, ca);
    bev_corners(q, cb);
    inter = quad_intersection_area(ca, cb);
  } else {
    const float lx = fmaxf(p[0] - p[3] * 0.5f, q[0] - q[3] * 0.5f), rx = fminf(p[0] + p[3] * 0.5f, q[0] + q[3] * 0.5f);
    const float ly = fmaxf(p[1] - p[4] * 0.5f, q[1] - q[4] * 0.5f), ry = fminf(p[1] + p[4] * 0.5f, q[1] + q[4] * 0.5f);
    inter = fmaxf(rx - lx, 0.0f) * fmaxf(ry - ly, 0.0f);
  }
  const float zl = fmaxf(p[2] - p[5] * 0.5f, q[2] - q[5] * 0.5f), zh = fminf(p[2] + p[5] * 0.5f, q[2] + q[5] * 0.5f);
  const float iv = inter * fmaxf(zh - zl, 0.0f);
  const float uv = p[3] * p[4] * p[5] + q[3] * q[4] * q[5] - iv;
  iou[t] = iv / fmaxf(uv, 1e-8f);
}

}  // namespace

extern "C" int cnrma_nms_mask_f32(const float* boxes_sorted, int n, float iou_thr, int rotated, uint64_t* mask,
                                  void* stream) {
  if (n <= 0) return n == 0 ? 0 : CNRMA_EINVAL;
  const int words = (n + 63) / 64;
  hipStream_t st = as_stream(stream);
  hipError_t e = cnrma_fill_bytes(mask, 0, (size_t)n * words * sizeof(uint64_t), st);
  if (e != hipSuccess) return -(int)e;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, st, boxes_sorted, n, iou_thr, rotated,
                     reinterpret_cast<unsigned long long*>(mask), words);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_box_iou_f32(const float* a, int na, const float* b, int nb, int rotated, int mode3d, float* iou,
                                 void* stream) {
  if (na <= 0 || nb <= 0) return (na == 0 || nb == 0) ? 0 : CNRMA_EINVAL;
  hipLaunchKernelGGL(iou_matrix_kernel, dim3((unsigned)ceil_div((int64_t)na * nb, 256)), dim3(256), 0, as_stream(stream),
                     a, na, b, nb, rotated, mode3d, iou);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
