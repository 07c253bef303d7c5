// switch_pointcloud (test path) on an existing point matrix + FCAF3D box decoding.
#include "common.h"

namespace {

// lanes across the 3+C columns of a row: coalesced row copies
__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ points, int64_t M, int C,
                                                          const int32_t* __restrict__ sel, float ax, float ay, float az,
                                                          float* __restrict__ coords, float* __restrict__ feats) {
  const int W = 3 + C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < M * W; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / W;
    const int c = (int)(t - m * W);
    const int64_t j = sel ? (int64_t)sel[m] : m;
    if (j < 0) continue;
    const float v = points[t];
    if (c < 3) coords[j * 3 + c] = v + (c == 0 ? ax : (c == 1 ? ay : az));   // ray_marching.py:364
    else feats[j * C + (c - 3)] = v;
  }
}

// FCAF3DHead._bbox_pred_to_bbox  (fcaf3d_head.py:300-349)
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ pts, const float* __restrict__ reg, int R,
                                                     int64_t n, int mode, float* __restrict__ boxes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = reg + i * R;
  const float* p = pts + i * 3;
  const float xc = p[0] + (b[1] - b[0]) / 2;                 // :304-306
  const float yc = p[1] + (b[3] - b[2]) / 2;
  const float zc = p[2] + (b[5] - b[4]) / 2;
  if (mode == 0) {                                           // 6-DoF, :309-319
    float* o = boxes + i * 6;
    o[0] = xc; o[1] = yc; o[2] = zc; o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5];
    return;
  }
  float* o = boxes + i * 7;
  o[0] = xc; o[1] = yc; o[2] = zc;
  if (mode == 3) {                                           // 'naive', :321-326
    o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5]; o[6] = b[6];
  } else if (mode == 2) {                                    // 'sin-cos', :327-335
    const float norm = sqrtf(b[6] * b[6] + b[7] * b[7]);
    o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5];
    o[6] = atan2f(b[6] / norm, b[7] / norm);
  } else {                                                   // 'fcaf3d', :336-349
    const float scale = b[0] + b[1] + b[2] + b[3];
    const float q = expf(sqrtf(b[6] * b[6] + b[7] * b[7]));
    const float alpha = 0.5f * atan2f(b[6], b[7]);
    o[3] = scale / (1.0f + q);
    o[4] = scale / (1.0f + q) * q;
    o[5] = b[5] + b[4];
    o[6] = alpha;
  }
}

// scores = sigmoid(cls) * sigmoid(centerness); max over classes  (fcaf3d_head.py:249-250)
__global__ __launch_bounds__(256) void scores_kernel(const float* __restrict__ cls, const float* __restrict__ ctr,
                                                     int64_t n, int n_cls, float* __restrict__ scores,
                                                     float* __restrict__ max_score) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sc = 1.0f / (1.0f + expf(-ctr[i]));
  float m = -__builtin_inff();
  for (int c = 0; c < n_cls; ++c) {
    const float s = (1.0f / (1.0f + expf(-cls[i * n_cls + c]))) * sc;
    scores[i * n_cls + c] = s;
    m = fmaxf(m, s);
  }
  max_score[i] = m;
}

// FCAF3DHead.forward_single tail (fcaf3d_head.py:276-298) in one pass over the fused head GEMM output y[n][ldy] =
// [centerness | reg (R) | cls (n_cls) | pad]: centerness, bbox_pred = [exp(scale * reg[:6]), reg[6:]], cls_score,
// prune score = max_c cls, points = coords[:, 1:] * voxel_size.
__global__ __launch_bounds__(256) void head_post_kernel(const float* __restrict__ y, int ldy, const int32_t* __restrict__ coords,
                                                        int64_t n, int R, int n_cls, const float* __restrict__ scale,
                                                        float voxel_size, float* __restrict__ centerness,
                                                        float* __restrict__ bbox_pred, float* __restrict__ cls,
                                                        float* __restrict__ max_cls, float* __restrict__ points) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = y + i * ldy;
  const float sc = scale[0];
  centerness[i] = r[0];
  for (int j = 0; j < R; ++j) bbox_pred[i * R + j] = j < 6 ? expf(r[1 + j] * sc) : r[1 + j];   // :284-286
  float m = -__builtin_inff();
  for (int c = 0; c < n_cls; ++c) {
    const float v = r[1 + R + c];
    cls[i * n_cls + c] = v;
    m = fmaxf(m, v);
  }
  max_cls[i] = m;                                                                                // :280
  const int4 cc = reinterpret_cast<const int4*>(coords)[i];
  points[i * 3 + 0] = (float)cc.y * voxel_size;                                                  // :296
  points[i * 3 + 1] = (float)cc.z * voxel_size;
  points[i * 3 + 2] = (float)cc.w * voxel_size;
}

// max over classes of sigmoid(cls) * sigmoid(centerness) only (the ranking key of :250-253)
__global__ __launch_bounds__(256) void max_score_kernel(const float* __restrict__ cls, const float* __restrict__ ctr,
                                                        int64_t n, int n_cls, float* __restrict__ max_score) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sc = 1.0f / (1.0f + expf(-ctr[i]));
  float m = -__builtin_inff();
  for (int c = 0; c < n_cls; ++c) m = fmaxf(m, (1.0f / (1.0f + expf(-cls[i * n_cls + c]))) * sc);
  max_score[i] = m;
}

// rows ids[0..k): scores = sigmoid(cls) * sigmoid(centerness) and decoded boxes, written at out_row0 + q
__global__ __launch_bounds__(256) void select_decode_kernel(const int64_t* __restrict__ ids, int64_t k,
                                                            const float* __restrict__ cls, const float* __restrict__ ctr,
                                                            const float* __restrict__ reg, const float* __restrict__ pts,
                                                            int n_cls, int R, int mode, int box_w,
                                                            float* __restrict__ scores, float* __restrict__ boxes) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= k) return;
  const int64_t i = ids ? ids[q] : q;
  const float sc = 1.0f / (1.0f + expf(-ctr[i]));
  for (int c = 0; c < n_cls; ++c) scores[q * n_cls + c] = (1.0f / (1.0f + expf(-cls[i * n_cls + c]))) * sc;
  const float* b = reg + i * R;
  const float* p = pts + i * 3;
  float* o = boxes + q * box_w;
  o[0] = p[0] + (b[1] - b[0]) / 2;
  o[1] = p[1] + (b[3] - b[2]) / 2;
  o[2] = p[2] + (b[5] - b[4]) / 2;
  if (mode == 0 || mode == 3 || mode == 2) { o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5]; }
  if (mode == 3) o[6] = b[6];
  if (mode == 2) { const float norm = sqrtf(b[6] * b[6] + b[7] * b[7]); o[6] = atan2f(b[6] / norm, b[7] / norm); }
  if (mode == 1) {
    const float scale = b[0] + b[1] + b[2] + b[3];
    const float qq = expf(sqrtf(b[6] * b[6] + b[7] * b[7]));
    o[3] = scale / (1.0f + qq);
    o[4] = scale / (1.0f + qq) * qq;
    o[5] = b[5] + b[4];
    o[6] = 0.5f * atan2f(b[6], b[7]);
  }
}

}  // namespace

extern "C" int cnrma_fcaf3d_head_post_f32(const float* y, int ldy, const int32_t* coords, int64_t n, int R, int n_cls,
                                          const float* scale, float voxel_size, float* centerness, float* bbox_pred,
                                          float* cls, float* max_cls, float* points, void* stream) {
  if (n <= 0) return n == 0 ? 0 : CNRMA_EINVAL;
  if (R < 6 || n_cls <= 0 || ldy < 1 + R + n_cls) return CNRMA_EINVAL;
  hipLaunchKernelGGL(head_post_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), y, ldy,
                     coords, n, R, n_cls, scale, voxel_size, centerness, bbox_pred, cls, max_cls, points);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_fcaf3d_max_score_f32(const float* cls, const float* centerness, int64_t n, int n_cls,
                                          float* max_score, void* stream) {
  if (n <= 0) return n == 0 ? 0 : CNRMA_EINVAL;
  hipLaunchKernelGGL(max_score_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), cls,
                     centerness, n, n_cls, max_score);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_fcaf3d_select_decode_f32(const int64_t* ids, int64_t k, const float* cls, const float* centerness,
                                              const float* reg, const float* points_xyz, int n_cls, int R,
                                              int yaw_mode, float* scores, float* boxes, void* stream) {
  if (k <= 0) return k == 0 ? 0 : CNRMA_EINVAL;
  if ((yaw_mode == 0 && R != 6) || ((yaw_mode == 1 || yaw_mode == 2) && R != 8) || (yaw_mode == 3 && R != 7) ||
      yaw_mode < 0 || yaw_mode > 3)
    return CNRMA_EINVAL;
  hipLaunchKernelGGL(select_decode_kernel, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, as_stream(stream), ids, k,
                     cls, centerness, reg, points_xyz, n_cls, R, yaw_mode, yaw_mode == 0 ? 6 : 7, scores, boxes);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_select_rows_f32(const float* points, int64_t M, int C, const int32_t* sel_index, float addx,
                                     float addy, float addz, float* coords, float* feats, void* stream) {
  if (M <= 0 || C <= 0) return CNRMA_EINVAL;
  int64_t blocks = ceil_div(M * (3 + C), 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(select_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), points, M, C,
                     sel_index, addx, addy, addz, coords, feats);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_fcaf3d_decode_f32(const float* points_xyz, const float* reg, int R, int64_t n, int yaw_mode,
                                       float* boxes, void* stream) {
  if (n <= 0) return n == 0 ? 0 : CNRMA_EINVAL;
  if ((yaw_mode == 0 && R != 6) || ((yaw_mode == 1 || yaw_mode == 2) && R != 8) || (yaw_mode == 3 && R != 7) ||
      yaw_mode < 0 || yaw_mode > 3)
    return CNRMA_EINVAL;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), points_xyz, reg,
                     R, n, yaw_mode, boxes);
  CNRMA_LAUNCH_CHECK();
  return 0;
}

extern "C" int cnrma_fcaf3d_scores_f32(const float* cls, const float* centerness, int64_t n, int n_cls, float* scores,
                                       float* max_score, void* stream) {
  if (n <= 0) return n == 0 ? 0 : CNRMA_EINVAL;
  if (n_cls <= 0) return CNRMA_EINVAL;
  hipLaunchKernelGGL(scores_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), cls, centerness,
                     n, n_cls, scores, max_score);
  CNRMA_LAUNCH_CHECK();
  return 0;
}
