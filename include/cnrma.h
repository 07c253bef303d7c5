/*
 * cnrma.h -- C-ABI of libcnrma_hip.so: the MI355X (gfx950) hot path of CN-RMA.
 *
 * The reference (SerCharles/CN-RMA) is pure Python over third-party CUDA packages; it has no native ABI of
 * its own.  This header is therefore the boundary a maintainer binds with ctypes: every
 * entry point names the reference Python it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; buffers are caller-owned (torch
 *     allocates them); the library never allocates, frees or synchronises unless stated;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, in order;
 *   - return value: 0 = ok, negative = -(hipError_t) of the failed launch or CNRMA_EINVAL for bad arguments;
 *   - fp32 everywhere the reference is fp32; integer outputs are int32;
 *   - feature maps are consumed channels-last ("NHWC": [V][H][W][C]); cnrma_nchw_to_nhwc_f32 converts the
 *     reference's NCHW layout in one pass;
 *   - re-entrant: no global mutable state.
 */
#ifndef CNRMA_H
#define CNRMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNRMA_EINVAL (-22)
#define CNRMA_ABI_VERSION 6   /* 6: the cnrma_debug_* entry points (and the experimental kernels behind them) left the product library: they exist in libcnrma_hip_exp.so (-DCNRMA_EXPERIMENTS) only; no signature changed; 5: + cnrma_sparse_conv_prepare_weights_bf16_t, cnrma_sparse_conv_wgrad_go_bf16, cnrma_sparse_conv_go_bf16 (+ its weight images), cnrma_bn_train_forward_f32 / _backward_f32; 4: gather-once convolution family, records-based point selection (SampleWs layout), *_ref_f32 hand-off */

int cnrma_abi_version(void);

/* A kernel launch that sets n_bytes (a multiple of 4, 4-byte aligned) to `byte` -- unlike a hipMemsetAsync node it re-executes
 * reliably when a captured graph is replayed.  The static trace clears ONE arena per scene with it (all hash tables and
 * neighbour tables of the scene: 0xFF = empty slot / no neighbour) and passes precleared = 1 to the builders below, which
 * then skip their own clearing launch (27 launches per scene). */
int cnrma_fill_bytes_u8(void* dst, int byte, size_t n_bytes, void* stream);

/* out[0] = number of i < n with values[i] < lo[i] or values[i] > hi[i] (n <= 4096).  The static trace registers every capacity /
 * branch assumption of its size plan as such a range on a device word (the live row counts the reference reads back with
 * nonzero() / ME's coordinate manager: ray_marching.py:781, :328-330); this folds them into the scene's status word in ONE launch
 * (it was two compares, an or, a sum and a cast: five). */
int cnrma_range_violations_i32(const int32_t* values, const int32_t* lo, const int32_t* hi, int n, int32_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * layout helper: feat_nchw[V][C][H][W] -> feat_nhwc[V][H][W][C]
 * (the reference keeps NCHW: projects/mvsdetection/models/ray_marching.py:64 and :799 gather [b,:,py,px])
 * ---------------------------------------------------------------------------------------------------------- */
int cnrma_nchw_to_nhwc_f32(const float* feat_nchw, float* feat_nhwc, int V, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a1-a3  dense unprojection + accumulate + mean
 * replaces: coordinates()            projects/mvsdetection/datasets/tsdf.py:14-29
 *           backproject()            projects/mvsdetection/models/ray_marching.py:21-69
 *           aggregate_2d_features()  ray_marching.py:220-244   (sum over views, in view order)
 *           clear_3d_features()      ray_marching.py:247-257   (divide by count, zero unseen voxels)
 * proj[V][3][4]: rows 0-1 already divided by backbone2d_stride (ray_marching.py:238-239).
 * volume[C][X][Y][Z] (mean, 0 where count == 0), count[X][Y][Z] int32 (number of views that see the voxel).
 * workspace: NULL, or CNRMA_DENSE_WORKSPACE_BYTES of device memory zeroed once by the caller (one per call in flight).  It is
 *   only used by the lockstep schedules behind cnrma_debug_dense_tuning (monotonic arrival counters of their group
 *   barriers: never reset; a barrier that timed out or a change of the group size merely dephases later barriers -- a
 *   performance effect, never a correctness one).  The product schedule (free-running brick order) ignores it.
 * ---------------------------------------------------------------------------------------------------------- */
#define CNRMA_DENSE_WORKSPACE_BYTES 1024
int cnrma_backproject_accum_f32(const float* feat_nhwc, const float* proj, int V, int C, int H, int W,
                                int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                float* volume, int32_t* count, void* workspace, int64_t workspace_bytes, void* stream);

/* The same with the feature maps handed over BY REFERENCE: feat_nhwc_ref is a device word holding the address of the
 * channels-last maps (read with one scalar load at kernel start).  A captured launch sequence (HIP graph) bakes its pointer
 * arguments in; through the reference it reads whatever tensor the producer -- the 2D network of ray_marching.py:211-213,
 * run in torch.channels_last -- has just written: the caller stores that tensor's address into the word (an 8-byte copy
 * on the same stream) and replays.  No layout pass, no copy of V x C x H' x W' floats into a static buffer.  C % 4 == 0. */
int cnrma_backproject_accum_ref_f32(const float* const* feat_nhwc_ref, const float* proj, int V, int C, int H, int W,
                                    int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                    float* volume, int32_t* count, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef CNRMA_EXPERIMENTS
/* libcnrma_hip_exp.so ONLY (built with -DCNRMA_EXPERIMENTS; the product library libcnrma_hip.so exports no cnrma_debug_* symbol
 * and holds no tuning state).  Debug / A-B aid (scripts/dense_ab.py, tests of the alternative voxel orders): overrides the dense
 * kernel's schedule switches {variant, slab, st, zt, tt, zi, chunk, persist, lpv, pipe, epi, lockstep, lattice, nt, own, stagger,
 * groups, ldspad} (host-side global state of that library; n = 0 restores the product configuration). */
int cnrma_debug_dense_tuning(const int* values, int n);
#endif

/* Backward of cnrma_backproject_accum_f32 w.r.t. the feature maps (training):
 * grad_feat_nhwc[v][pix_v(g)][c] += grad_volume[c][g] / count[g] for every valid (voxel g, view v) pair; the output is
 * zeroed first; float atomics (~58 voxels share a pixel), so the sum order is not fixed. */
int cnrma_backproject_backward_f32(const float* grad_volume, const int32_t* count, const float* proj, int V, int C, int H,
                                   int W, int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                                   float* grad_feat_nhwc, void* stream);

/* single view, debug/parity: px,py int32 [G] (rounded pixel, INT32_MIN when not finite), valid uint8 [G] */
int cnrma_backproject_index_f32(const float* proj_view, int H, int W, int X, int Y, int Z, float voxel_size,
                                float ox, float oy, float oz, int32_t* px, int32_t* py, uint8_t* valid,
                                void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a4  ray parameters      replaces get_ray_parameter()  ray_marching.py:71-111
 * proj_inv[V][4][4] = torch.inverse([P;0 0 0 1]) computed by the host (kept on the host LAPACK so that it is
 * bit-identical to the reference's call at ray_marching.py:100).  o[V][3], d[V][3][H*W].
 * ---------------------------------------------------------------------------------------------------------- */
int cnrma_ray_params_f32(const float* proj_inv, int V, int H, int W, float* o, float* d, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a5/a7  NeuS ray-marching aggregation, two phases (variable-length output)
 * replaces: ray_projection_neus()                 ray_marching.py:687-807
 *           aggregate_2d_features_ray_marching()  ray_marching.py:260-307 (concat of views, w/mean(w) scaling)
 * All V views in one launch; one lane per ray, steps marched in order (fp64 running product like the CPU
 * cumprod at :760).  t_one = sqrt(X^2+Y^2+Z^2)*voxel_size/N as computed in double by the host, rounded to fp32
 * (:710-714).  Sample order everywhere = (view, row, col, step).
 *
 * phase 1  count[V*H*W] int32 kept samples per ray; wsum[V*H*W] fp64 sum of kept weights per ray.
 * then     cnrma_exclusive_scan_i32(count) -> row_offset[V*H*W+1]  (row_offset[last] = M).
 * phase 2  writes rows.  Destination is described by three strided views so that the same kernel serves
 *            (i) the reference's raw rows [M][4+C]      (xyz=rows, w=rows+3, feat=rows+4, all stride 4+C) and
 *            (ii) the fused production layout: coords[Ms][3] (+offset), feats[Ms][C] * (w * inv_mean) with the
 *                 subsample mask of sample_points() applied (fcaf3d_transforms.py:283-296, ray_marching.py:364-405).
 *          sel_index: NULL = every row m goes to output row m; else int32[M], output row of source row m or -1
 *          (an exclusive prefix sum of the keep-mask, -1 where dropped).
 *          w_div: NULL = store features unscaled; else a DEVICE scalar (mean weight, see cnrma_rma_mean_weight):
 *                 features are multiplied by (w / w_div[0]) (:303-304).
 *          out_sample: NULL or int32 [M][2] = (ray index, step) of every row (alignment aid for parity tests).
 *          add[3]: added to xyz (the scene offset of switch_pointcloud, :364); pass zeros for raw rows.
 * ---------------------------------------------------------------------------------------------------------- */
int cnrma_rma_neus_count_f32(const float* proj_inv, const float* tsdf, int V, int H, int W, int X, int Y, int Z,
                             float voxel_size, float ox, float oy, float oz, int n_steps, float t_one, float thr,
                             int32_t* count, double* wsum, void* stream);

int cnrma_rma_neus_emit_f32(const float* proj_inv, const float* tsdf, const float* feat_nhwc, int V, int C, int H,
                            int W, int X, int Y, int Z, float voxel_size, float ox, float oy, float oz, int n_steps,
                            float t_one, float thr, const int32_t* row_offset, const int32_t* sel_index,
                            const float* w_div, float addx, float addy, float addz,
                            float* out_xyz, int xyz_stride, float* out_w, int w_stride, float* out_feat,
                            int feat_stride, int32_t* out_sample, void* stream);

/* Single-march variant of the NeuS pair (production path).  cnrma_rma_neus_march_f32 = phase 1 plus a per-ray record
 * of the kept samples: kept[ray][cap] x {int32 weight bits, int32 step} (8 bytes each), cap >= floor(1/thr) + 2 (a
 * ray's weights sum to <= 1, so it keeps at most 1/thr samples; overflow: int32[4] -- [0] counts violations and must read
 * 0, [1..3] are scratch words of the launch).
 * cnrma_rma_neus_emit_rows_f32 = phase 2 from those records: a per-ray pass moves the records of the selected samples to
 * their output position (records: scratch of 16 bytes x n_out), then one 8-lane group per OUTPUT row (n_out of them)
 * writes the row -- nothing is re-marched.
 * Same destination description and arithmetic (place = o + d * (n * t_one)) as cnrma_rma_neus_emit_f32. */
/* sig_table (may be NULL): table[voxel] = sigmoid(-tsdf[voxel]) with the reference's CPU arithmetic, built once per scene by
 * cnrma_rma_sigmoid_table_f32 -- the march then reads it instead of evaluating the sigmoid at every step (same function of
 * the same value: bit-identical weights).  With sig_table == NULL the TSDF is read and the sigmoid evaluated per step. */
int cnrma_rma_sigmoid_table_f32(const float* tsdf, int64_t n, float* table, void* stream);
/* The march's per-scene tables in one call: the sigmoid table and (skip_table != NULL) the free-space skip table -- one byte per
 * 4 x 4 x 4 block of voxels = a radius R in {0, 4, .., 16} such that every voxel within Chebyshev distance R of any voxel of the
 * block holds the block's table value bit for bit.  Samples whose successor has the same table value are no-ops of the march
 * (ray_marching.py:759-767: alpha = 0, w = 0 < thr, transmittance x 1), so a ray standing in such a block jumps over the steps
 * that are certain to stay inside the radius without evaluating them: same records, same sums, fewer evaluated steps (rays spend
 * most of their steps in free space).  skip_table: cnrma_rma_skip_table_bytes(X, Y, Z) bytes (radii + scratch of the build). */
size_t cnrma_rma_skip_table_bytes(int X, int Y, int Z);
int cnrma_rma_march_tables_f32(const float* tsdf, int X, int Y, int Z, float* table, void* skip_table, void* stream);
#ifdef CNRMA_EXPERIMENTS
/* libcnrma_hip_exp.so only.  Parity aid: the march's division by the voxel size (reciprocal + two quotient refinements) next to
 * the IEEE division */
int cnrma_debug_div_by_voxel_size_f32(const float* a, int64_t n, float voxel_size, float* q_fast, float* q_ref, void* stream);
#endif
int cnrma_rma_neus_march_f32(const float* proj_inv, const float* tsdf, const float* sig_table, int V, int H, int W, int X,
                             int Y, int Z, float voxel_size, float ox, float oy, float oz, int n_steps, float t_one,
                             float thr, int32_t* count, double* wsum, void* kept, int cap, int32_t* overflow,
                             const void* skip_table, void* stream);       /* skip_table: may be NULL; needs sig_table */

/* The layout pass (cnrma_nchw_to_nhwc_f32) and the march (cnrma_rma_neus_march_f32) of one scene in ONE launch: the two
 * are independent (the march does not read the feature maps), one is a pure HBM stream and the other VALU-bound on
 * cache-resident data, and as separate launches they do not overlap.  Same outputs as the two calls, bit for bit; falls
 * back to two launches when the 16-byte layout kernel does not apply (H*W % 4, C % 4, alignment) or without sig_table. */
int cnrma_nchw_to_nhwc_march_f32(const float* feat_nchw, float* feat_nhwc, int C, const float* proj_inv, const float* tsdf,
                                 const float* sig_table, int V, int H, int W, int X, int Y, int Z, float voxel_size, float ox,
                                 float oy, float oz, int n_steps, float t_one, float thr, int32_t* count, double* wsum,
                                 void* kept, int cap, int32_t* overflow, const void* skip_table, void* stream);

/* Backward of cnrma_rma_neus_emit_rows_f32 w.r.t. the feature maps (training; the weights carry no
 * gradient: the reference computes them under torch.no_grad(), ray_marching.py:705).
 * grad_feat_nhwc[ray][c] = sum over the selected rows j of the ray of grad_out_feat[j][c] * (w_j / w_div[0]); rows are
 * walked in step order from the kept-sample records of the forward march (no atomics; rays without rows get zeros). */
int cnrma_rma_neus_rows_backward_f32(const float* grad_out_feat, int grad_stride, int V, int C, int H, int W,
                                     const int32_t* row_offset, const void* kept, int cap, const int32_t* sel_index,
                                     const float* w_div, float* grad_feat_nhwc, void* stream);
/* n_out = capacity of the output (rows of `records`, grid size); n_out_dev (may be NULL) = device word with the live number of
 * output rows (the n_sel of cnrma_mask_to_index); sel_cap = entries of sel_index (rows >= sel_cap are dropped).
 * kept == NULL and row_offset == NULL: `records` already holds the rows to emit (cnrma_rma_select_records). */
int cnrma_rma_neus_emit_rows_f32(const float* proj_inv, const float* feat_nhwc, int V, int C, int H, int W,
                                 int n_steps, float t_one, const int32_t* row_offset, int64_t n_out,
                                 const int32_t* n_out_dev, const void* kept, int cap, const int32_t* sel_index,
                                 int64_t sel_cap, void* records, const float* w_div, float addx, float addy, float addz,
                                 float* out_xyz, int xyz_stride, float* out_w, int w_stride, float* out_feat,
                                 int feat_stride, int32_t* out_sample, void* stream);
/* ... with the feature maps by reference (see cnrma_backproject_accum_ref_f32) */
int cnrma_rma_neus_emit_rows_ref_f32(const float* proj_inv, const float* const* feat_nhwc_ref, int V, int C, int H, int W,
                                 int n_steps, float t_one, const int32_t* row_offset, int64_t n_out,
                                 const int32_t* n_out_dev, const void* kept, int cap, const int32_t* sel_index,
                                 int64_t sel_cap, void* records, const float* w_div, float addx, float addy, float addz,
                                 float* out_xyz, int xyz_stride, float* out_w, int w_stride, float* out_feat,
                                 int feat_stride, int32_t* out_sample, void* stream);

/* Device-side replacement of sample_points()'s np.random.choice(M, n_keep, replace=False)
 * (fcaf3d_transforms.py:283-296): mask[0..M) gets exactly min(M, n_keep) ones, a uniformly random subset that is a
 * deterministic function of `seed` (smallest n_keep 32-bit hash keys, ties by index).  M = min(m_dev[0], m_cap) is read
 * on the device (no host sync); mask has m_cap entries, those behind M are set to 0.  seed_dev (may be NULL): device word
 * mixed into the seed, so that replays of a captured launch sequence draw fresh subsets.  Same distribution as the
 * reference, different random stream. */
size_t cnrma_sample_workspace_bytes(void);
int cnrma_sample_mask(const int32_t* m_dev, int64_t m_cap, int n_keep, uint32_t seed, const uint32_t* seed_dev,
                      uint8_t* mask, void* workspace, void* stream);
/* sample_points (ray_marching.py:339-358) on the march's per-ray sample records, without the m_cap-sized mask and index:
 * the same random subset as cnrma_sample_mask(m_dev, m_cap, n_keep, seed, seed_dev) -- the keep predicate is a function of
 * the row number -- counted per ray, placed by a scan over the R rays, and written as records[j] = {ray, step, weight bits, 0}
 * in row order (what cnrma_mask_to_index + the record scatter of cnrma_rma_neus_emit_rows_f32 produce); n_sel[0] = rows
 * written (<= min(n_keep, rec_cap)).  row_offset [R + 1] = exclusive scan of the march's counts, kept / cap = its records.
 * Scratch: ray_counts [R], ray_offsets [R + 1], scan_ws cnrma_scan_workspace_bytes(R), sample_ws cnrma_sample_workspace_bytes(). */
int cnrma_rma_select_records(const int32_t* row_offset, int64_t R, const void* kept, int cap, const int32_t* m_dev,
                             int64_t m_cap, int n_keep, uint32_t seed, const uint32_t* seed_dev, void* sample_ws,
                             int32_t* ray_counts, int32_t* ray_offsets, void* scan_ws, int64_t rec_cap, void* records,
                             int32_t* n_sel, void* stream);
/* keep-mask of the k largest scores (ties -> smaller index): the row set of torch.topk(scores, k) as used by the
 * pts_threshold pruning (fcaf3d_head.py:131-137) and nms_pre (:252-256), by 3-pass radix select instead of a sort.
 * n = min(n_dev[0], n_cap) (device); mask has n_cap entries, those behind n are set to 0;
 * workspace: cnrma_sample_workspace_bytes(). */
int cnrma_topk_mask_f32(const float* scores, const int32_t* n_dev, int64_t n_cap, int k, uint8_t* mask,
                        void* workspace, void* stream);
/* torch.topk(scores, k)[1] for nms_pre (fcaf3d_head.py:252-256), k <= 1024: the same radix select, the kept rows collected
 * in a list and sorted by one workgroup -- out_idx[0..min(n, k)) = rows in descending score order (ties -> smaller row),
 * the slots behind them hold row 0.  workspace: cnrma_sample_workspace_bytes(). */
int cnrma_topk_indices_f32(const float* scores, const int32_t* n_dev, int64_t n_cap, int k, int64_t* out_idx,
                           void* workspace, void* stream);

/* a6  depth variant   replaces ray_projection_depth()  ray_marching.py:809-956
 * Every ray emits exactly NUM = max(1, 2*select_grids) candidate slots; count[ray] = number of slots with
 * weight > 0.  Same two-phase protocol and destination description as the NeuS pair. */
int cnrma_rma_depth_count_f32(const float* proj_inv, const float* tsdf, int V, int H, int W, int X, int Y, int Z,
                              float voxel_size, float ox, float oy, float oz, int n_steps, float t_one,
                              int select_grids, int32_t* count, double* wsum, void* stream);

int cnrma_rma_depth_emit_f32(const float* proj_inv, const float* tsdf, const float* feat_nhwc, int V, int C, int H,
                             int W, int X, int Y, int Z, float voxel_size, float ox, float oy, float oz,
                             int n_steps, float t_one, int select_grids, const int32_t* row_offset,
                             const int32_t* sel_index, const float* w_div, float addx, float addy, float addz,
                             float* out_xyz, int xyz_stride, float* out_w, int w_stride, float* out_feat,
                             int feat_stride, int64_t sel_cap, int64_t out_cap, void* stream);
/* sel_cap / out_cap (0 = unbounded): capacities of sel_index and of the output buffers when the row count stays on the
 * device (static trace): rows beyond them are dropped instead of written. */

/* Reference quirk: a view whose rays keep exactly ONE sample in total is dropped (ray_marching.py:781-782: squeeze()
 * makes the index 0-dim, len() raises, the bare except skips the view): zeroes that view's counts and weight sums. */
int cnrma_rma_drop_single_sample_views(int32_t* count, double* wsum, int V, int64_t rays_per_view, void* stream);

/* mean_w[0] = (float)(wsum_total[0] / m_total[0])   -- torch.mean(weights), ray_marching.py:303 */
int cnrma_rma_mean_weight(const double* wsum_total, const int32_t* m_total, float* mean_w, void* stream);

/* utilities used between the phases (device-side; no host sync) */
size_t cnrma_scan_workspace_bytes(int64_t n);
/* out[0..n] (n+1 entries): out[i] = sum(in[0..i-1]) */
int cnrma_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* workspace, void* stream);
/* sum of n doubles -> out[0] */
int cnrma_sum_f64(const double* in, double* out, int64_t n, void* workspace, void* stream);
/* mask uint8[n] -> sel_index int32[n] (exclusive rank where mask, -1 elsewhere); total written to n_sel[0] */
int cnrma_mask_to_index(const uint8_t* mask, int32_t* sel_index, int32_t* n_sel, int64_t n, void* workspace,
                        void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a8  switch_pointcloud (test path) on an existing point matrix
 * replaces ray_marching.py:360-405: coord = xyz + offset, keep rows where mask (order preserved).
 * points[M][3+C] -> coords[Ms][3], feats[Ms][C]; sel_index as above (NULL = keep all).
 * ---------------------------------------------------------------------------------------------------------- */
int cnrma_select_rows_f32(const float* points, int64_t M, int C, const int32_t* sel_index, float addx, float addy,
                          float addz, float* coords, float* feats, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a9  voxelisation = ME.utils.batch_sparse_collate + ME.SparseTensor(quantization_mode=RANDOM_SUBSAMPLE)
 * replaces ray_marching.py:328-330 (MinkowskiEngine v0.5.4; semantics SURVEY.md Appendix A).
 * q = floor(coord / voxel_size) (true fp32 division) -> int32, batch id prepended; duplicate voxels collapse
 * to the row with the SMALLEST source index (the CPU behaviour of ME).  row_order: 0 = output rows in order of
 * first occurrence; 1 = output rows sorted by the Morton code of (x,y,z) (ME leaves the row order
 * implementation-defined; spatial order makes the convolution gathers cache-friendly and is inherited by every
 * strided level).
 * hash_keys uint64[hash_cap], hash_vals int32[hash_cap]: open-addressing table, hash_cap a power of two >= 2*M;
 * on return it maps voxel key -> output row (reusable as the coordinate map of the level).
 * A coordinate that does not fit the coordinate key (|coord / voxel_size| >= 32767, NaN, batch id >= 65536) makes the call
 * report n_out[0] = -1 (the host raises; the static plan's status word flags it) instead of aliasing another voxel.
 * M = capacity of the input, m_dev (may be NULL) = device word with its live row count.  out_cap (0 = M): rows of the
 * output buffers; unique voxels beyond it are dropped and left unmapped (n_out still counts them: n_out > out_cap tells).
 * out_coords int32[Mu][4] (b,x,y,z), out_feats[Mu][C], out_src int32[Mu], n_out[0] = Mu (device).
 * workspace: cnrma_voxelize_workspace_bytes(M).
 * ---------------------------------------------------------------------------------------------------------- */
size_t cnrma_voxelize_workspace_bytes(int64_t M);
int cnrma_voxelize_f32(const float* coords, const float* feats, int64_t M, const int32_t* m_dev, int C,
                       float voxel_size, int batch_id, int row_order, uint64_t* hash_keys, int32_t* hash_vals,
                       int64_t hash_cap, int32_t* out_coords, float* out_feats, int32_t* out_src, int64_t out_cap,
                       int32_t* n_out, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a10-a11  sparse operators (replace the MinkowskiEngine v0.5.4 surface used by
 *          projects/mvsdetection/models/fcaf3d_backbone.py:26-31,63-70 and fcaf3d_head.py:64-98,107-139,275-298)
 * Coordinates: int32 [N][4] = (batch, x, y, z) at an explicit tensor stride; features fp32 [N][C] row-major.
 * Row counts that are produced on the device are written to int32 device words (n_out) and all kernels that
 * consume such a tensor take BOTH a capacity (grid sizing) and a device pointer to the live row count, so a whole
 * forward can be enqueued without host synchronisation.  Pass n_dev = NULL to use the capacity as the count.
 * ---------------------------------------------------------------------------------------------------------- */

/* coordinate map: build a hash table key(coords[i]) -> i for N unique coordinates */
int cnrma_sparse_build_map(const int32_t* coords, int64_t n_cap, const int32_t* n_dev, uint64_t* hash_keys,
                           int32_t* hash_vals, int64_t hash_cap, int precleared /* hash_keys already hold 0xFF bytes */,
                           void* stream);

/* strided output coordinate set: unique(floor(p / new_stride) * new_stride) in first-occurrence order
 * (MinkowskiConvolution / MinkowskiMaxPooling with stride 2: fcaf3d_backbone.py:26-31, BasicBlock stride 2).
 * Builds the output map (hash) as well.  out_cap (0 = n_cap): rows of out_coords; sites beyond it are dropped and left
 * unmapped (table value -1), n_out still counts them.  workspace: cnrma_voxelize_workspace_bytes(n_cap). */
int cnrma_sparse_stride_coords(const int32_t* in_coords, int64_t n_cap, const int32_t* n_dev, int new_stride,
                               uint64_t* hash_keys, int32_t* hash_vals, int64_t hash_cap, int32_t* out_coords,
                               int64_t out_cap, int32_t* n_out, void* workspace, void* stream);
/* The same set, in the same order, for input rows SORTED by the voxeliser's Morton key (cnrma_voxelize_f32 row_order 1 and every
 * set strided from such a set): the parents' keys are non-decreasing along the rows, so the first row of every parent is found
 * by an adjacent comparison -- no hash table, no atomics.  The coordinate map of the result is not built (cnrma_sparse_build_map
 * when a kernel map needs it).  workspace: cnrma_voxelize_workspace_bytes(n_cap). */
int cnrma_sparse_stride_coords_sorted(const int32_t* in_coords, int64_t n_cap, const int32_t* n_dev, int new_stride,
                                      int32_t* out_coords, int64_t out_cap, int32_t* n_out, void* workspace, void* stream);

/* neighbour table ("kernel map", output-stationary): nbr[No][K] = input row at out_coord + offset[k], or -1.
 * offsets int32 [K][3] in coordinate units (already multiplied by the tensor stride), k with x fastest. */
int cnrma_sparse_kernel_map(const int32_t* out_coords, int64_t no_cap, const int32_t* no_dev,
                            const uint64_t* in_hash_keys, const int32_t* in_hash_vals, int64_t hash_cap,
                            const int32_t* offsets, int K, int32_t* nbr, void* stream);

/* Faster builders of the SAME table for the two structured cases (results identical to cnrma_sparse_kernel_map):
 * _symmetric: stride-1 odd kernel on one coordinate set -- nbr[o][k] = i <=> nbr[i][K-1-k] = o, half the probes;
 * _strided:   stride-2 maps (conv k3 / k1, pooling k2) driven from the input side: an input can only feed the
 *             outputs on the coarse lattice around it (3.4 probes per input for k3, 1 for k2/k1, instead of K per
 *             output); probes go to the OUTPUT set's map.  in_stride = tensor stride of the input. */
int cnrma_sparse_kernel_map_symmetric(const int32_t* coords, int64_t n_cap, const int32_t* n_dev,
                                      const uint64_t* hash_keys, const int32_t* hash_vals, int64_t hash_cap,
                                      const int32_t* offsets, int K, int32_t* nbr, int precleared /* nbr holds 0xFF bytes */,
                                      void* stream);
int cnrma_sparse_kernel_map_strided(const int32_t* in_coords, int64_t n_cap, const int32_t* n_dev, int in_stride,
                                    int kernel_size, const uint64_t* out_hash_keys, const int32_t* out_hash_vals,
                                    int64_t hash_cap, int32_t* nbr, int64_t no_cap, int precleared, void* stream);
/* _children: the 3x3x3 stride-1 table of the set cnrma_sparse_convtr_gen_* generates (fcaf3d_head.py:61-70 up blocks: all 8
 *             children of every parent, child m of parent p at row 8 p + m) from the PARENTS' 3x3x3 table parent_nbr
 *             [np_cap][27]: nbr[8 p + m][k] = 8 * parent_nbr[p][k'] + m' with (k', m') the parent offset / child rank of
 *             child m displaced by offset k -- no hash table, no probes.  nbr [8 * np_cap][27]. */
int cnrma_sparse_kernel_map_children(const int32_t* parent_nbr, int64_t np_cap, const int32_t* np_dev, int32_t* nbr,
                                     void* stream);

/* fused sparse convolution, output-stationary gather-GEMM on fp32 MFMA:
 *   out[o] = act( (sum_k in[nbr[o][k]] @ W[k]) * scale + shift + residual[o] )
 * W[K][Cin][Cout] (ME "kernel" layout), scale/shift per output channel (folded BatchNorm / bias; NULL = 1 / 0),
 * residual [No][Cout] or NULL, act: 0 none, 1 ReLU, 2 ELU(alpha=1).  nbr == NULL means K == 1 identity map.
 * workspace (optional, may be NULL): fp32 scratch for the split over kernel offsets that short layers use to fill
 * the chip (partial slabs reduced in a fixed order: run-to-run deterministic, no atomics);
 * cnrma_sparse_conv_workspace_bytes gives the size that allows the widest split. */
size_t cnrma_sparse_conv_workspace_bytes(int64_t no_cap, int Cout, int K);
int cnrma_sparse_conv_f32(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* weight, int Cout,
                          const float* scale, const float* shift, const float* residual, int act,
                          float* out_feats, int64_t no_cap, const int32_t* no_dev, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Which kernel variant a convolution of these sizes launches (pure host function; the choice depends on the output
 * CAPACITY no_cap, never on live row counts, so a captured launch sequence replays the same kernels): out6 = {tile rows,
 * tile columns, splits over the kernel offsets, offsets per split, load stages in flight (1 | 2), tile shape id}.
 * mode: 0 = fp32 MFMA kernel (cnrma_sparse_conv_f32), 1 = f16x3, 2 = bf16, 3 = bf16x6.  Tests use it to prove which
 * variants a parity case covered (fcaf3d_backbone.py:59-107 / fcaf3d_head.py:61-139 run through all of them). */
int cnrma_sparse_conv_plan(int64_t no_cap, int Cin, int Cout, int K, int mode, int slices, size_t workspace_bytes, int* out6);
#ifdef CNRMA_EXPERIMENTS
/* libcnrma_hip_exp.so ONLY (-DCNRMA_EXPERIMENTS): that library additionally carries the measured-and-rejected kernel forms (the
 * warp-specialised stage kernel, the first and third form of the gather-once convolution, ablation / s_memtime-stamped
 * instantiations); the product library has none of them and no tuning state.
 * Debug / A-B aid (scripts/conv_sweep.py, variant-forcing tests): overrides {tile shape id, splits, load stages in flight,
 * ablation mask} of every later convolution launch (-1 = the launcher's choice; n = 0 restores the product configuration).
 * A non-zero ablation mask routes f16x3 launches to a DIAGNOSTIC kernel that leaves stage components out (timing
 * experiments: its results are meaningless; stage kernel bits: 1 MFMAs, 2 A loads, 4 B loads, 8 LDS stores, 16 barriers;
 * gather-once kernel: 1 MFMAs + fragment reads, 2 union-row loads, 4 weight loads, 8 LDS stores, 16 epilogue stores,
 * 32 local-index load; tile-union builder: 32 / 64 insertion / numbering).  Host-side global state; product code never calls it and nothing reads the
 * environment. */
int cnrma_debug_conv_tuning(const int* values, int n);
#endif

/* fp32-grade convolution on the bf16 matrix cores ("bf16x6": each fp32 operand split exactly into 3 bf16 pieces, the 6
 * significant partial products accumulated in fp32; relative error ~2^-23 per product, i.e. that of an fp32 fma chain;
 * 2.67x fewer matrix-pipe cycles than the fp32 MFMA path).  Needs Cin % 32 == 0.
 * cnrma_sparse_conv_prepare_weights: weight fp32 [K][Cin][Cout] -> weight_split bf16 [3][K][Cin/32][Cout_p][32], Cout_p = Cout
 * rounded up to 128 with zero rows (cnrma_sparse_conv_weight_bytes bytes), done once per layer.  Same arguments /
 * epilogue as cnrma_sparse_conv_f32 otherwise. */
size_t cnrma_sparse_conv_weight_bytes(int K, int Cin, int Cout);
int cnrma_sparse_conv_prepare_weights(const float* weight, int K, int Cin, int Cout, void* weight_split, void* stream);
/* Pre-split feature companions: a feature matrix [N][C] (C % 8 == 0) can carry its bf16 split
 * [N+1][C/8][3 planes][8] (48 bytes per 8 channels; row N is all zeros and stands in for missing neighbours).
 * cnrma_sparse_split_features builds one; the bf16x6 convolutions read it (in_split, may be NULL: then the fp32
 * features are split while they are staged) and write the companion of their OUTPUT in the epilogue (out_split, may be
 * NULL; [no_cap+1] rows), so chains of convolutions never re-split and the gathers carry no VALU work. */
int cnrma_sparse_split_features(const float* feats, int64_t n_cap, const int32_t* n_dev, int C, void* out_split,
                                void* stream);
int cnrma_sparse_conv_bf16x6(const float* in_feats, const void* in_split, int64_t in_zero_row, int Cin,
                             const int32_t* nbr, int K, const void* weight_split, int Cout, const float* scale,
                             const float* shift, const float* residual, int act, float* out_feats, void* out_split,
                             int64_t no_cap, const int32_t* no_dev, void* workspace, size_t workspace_bytes,
                             void* stream);
/* generative transpose (k2 s2) on the same path: weight_split from prepare_weights(K = 8); out_split has 8*n_cap+1 rows */
int cnrma_sparse_convtr_gen_bf16x6(const int32_t* in_coords, const float* in_feats, const void* in_split, int64_t n_cap,
                                   const int32_t* n_dev, int Cin, int half_stride, const void* weight_split, int Cout,
                                   const float* scale, const float* shift, int act, int32_t* out_coords,
                                   float* out_feats, void* out_split, void* stream);

/* Pair-list variant of cnrma_sparse_conv_f16x3 for layers whose kernel map is nearly empty (the stem: a stride-2
 * convolution on a point sample has 1.3-1.5 neighbours per output row, so every row tile of the output-stationary kernel
 * carries all 27 offsets and 95 % of its matrix work multiplies zero rows).  The valid (output, offset) entries of `nbr`
 * are regrouped into a list sorted by offset (runs padded to 128 rows), the MFMA kernel runs over that list with the
 * weight slice chosen per run, and every output row adds its products in ascending offset order (deterministic) before
 * the fused epilogue.  Same operands, same result contract as cnrma_sparse_conv_f16x3 (replaces the same
 * ME.MinkowskiConvolution call: fcaf3d_backbone.py:79-80 of the reference).  pair_cap: capacity of the list, a multiple
 * of 128 and >= (number of valid entries) + 128 * K -- for a stride-2 convolution min(27 * N_out, 8 * N_in) + 128 * K is a
 * provable bound (an input row is a neighbour of at most 2 outputs per axis).  workspace:
 * cnrma_sparse_conv_pairs_workspace_bytes(no_cap, K, Cout, pair_cap) bytes. */
size_t cnrma_sparse_conv_pairs_workspace_bytes(int64_t no_cap, int K, int Cout, int64_t pair_cap);
int cnrma_sparse_conv_pairs_f16x3(const float* in_feats, const float* in_amax, int Cin, const int32_t* nbr, int K,
                                  const void* weight_split, int Cout, const float* scale, const float* shift,
                                  const float* residual, int act, float* out_feats, float* out_amax, int64_t no_cap,
                                  const int32_t* no_dev, int64_t pair_cap, void* workspace, size_t workspace_bytes,
                                  void* stream);
/* The same regrouping in exact fp32 (`v_mfma_f32_32x32x2_f32` over the pair runs; weight = ME layout fp32 [K][Cin][Cout]): the
 * stem of the 256-channel configuration at the reference's own arithmetic (fcaf3d_backbone.py:79-80). */
int cnrma_sparse_conv_pairs_f32(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* weight, int Cout,
                                const float* scale, const float* shift, const float* residual, int act, float* out_feats,
                                int64_t no_cap, const int32_t* no_dev, int64_t pair_cap, void* workspace, size_t workspace_bytes,
                                void* stream);

/* 22-bit convolution on the fp16 matrix cores ("f16x3"): each operand is scaled by a power of two taken from an upper
 * bound of its tensor's magnitude (so that the largest element sits at 2^13..2^14) and split into two fp16 pieces
 * a 2^s = h + m (round to nearest: |a 2^s - h - m| <= 2^-22 |a 2^s|); the products hh + hm + mh are accumulated in fp32 and
 * the scales are removed in the epilogue (exact).  Relative error <= 3 x 2^-22 per product -- below the rounding noise
 * of an fp32 accumulation of the same length -- at half the matrix-pipe cycles and 2/3 of the LDS traffic of bf16x6.
 * Needs Cin % 32 == 0.  A magnitude bound ("amax") is cnrma_amax_bytes() bytes of device memory: 64 words, one per
 * 64-byte line, whose maximum is the bound (blocks publish into the slot their index hashes to: same-address atomics
 * would serialise).  in_amax: bound >= max|in_feats| (cnrma_absmax_f32, or the out_amax of the producing convolution);
 * out_amax (may be NULL): zeroed by the caller, receives max|out_feats|.
 * cnrma_sparse_conv_prepare_weights_f16: weight fp32 [K][Cin][Cout] -> fp16 [2][K][Cin/32][Cout_p][32] of weight * 2^s + a trailer
 * holding max|weight| (cnrma_sparse_conv_f16_weight_bytes bytes in all), done once per layer. */
size_t cnrma_amax_bytes(void);
int cnrma_absmax_f32(const float* in, int64_t n_cap, const int32_t* n_dev, int C, float* out_amax, void* stream);
/* The feature half of cnrma_rma_neus_emit_rows_f32 (ray_marching.py:298-307: point_features * weights / mean(weights)) for point
 * records {ray, step, weight bits, 0} in ANY order: out_feat[j][:] = feat[ray_j][:] * (w_j / w_div[0]), same arithmetic.  The
 * static trace carries the 16-byte records of cnrma_rma_select_records through the voxeliser (as 4-float rows) and emits the
 * features of the surviving representatives straight into the sparse tensor's rows, with their magnitude bound (out_amax:
 * cnrma_amax_bytes() ZEROED bytes, or NULL): no [M, C] intermediate, no row gather, no absmax pass.  Exactly one of feat_nhwc /
 * feat_nhwc_ref (see cnrma_backproject_accum_ref_f32) is given. */
int cnrma_rma_emit_features_f32(const float* feat_nhwc, const float* const* feat_nhwc_ref, int C, const void* records,
                                int64_t n_cap, const int32_t* n_dev, const float* w_div, float* out_feat, int feat_stride,
                                float* out_amax, void* stream);
size_t cnrma_sparse_conv_f16_weight_bytes(int K, int Cin, int Cout);
int cnrma_sparse_conv_prepare_weights_f16(const float* weight, int K, int Cin, int Cout, void* weight_split,
                                          void* stream);
int cnrma_sparse_conv_f16x3(const float* in_feats, const float* in_amax, int Cin, const int32_t* nbr, int K,
                            const void* weight_split, int Cout, const float* scale, const float* shift,
                            const float* residual, int act, float* out_feats, float* out_amax, int64_t no_cap,
                            const int32_t* no_dev, void* workspace, size_t workspace_bytes, void* stream);
/* "Gather-once" form of cnrma_sparse_conv_f16x3 for 3x3x3 stride-1 convolutions (fcaf3d_backbone.py:59-87 BasicBlock
 * convolutions, fcaf3d_head.py:61-83 out / up blocks): same operands, same epilogue, same result up to fp32 rounding order.
 * cnrma_sparse_tile_union_build lists, per 64-row output tile of a neighbour table nbr[no_cap][27], the DISTINCT input rows
 * the tile reads (in groups of kernel offsets whose union fits the LDS image) and the local index of every (row, offset)
 * in that list -- once per (coordinate set, kernel), into cnrma_sparse_tile_union_bytes(no_cap) bytes, reusable by every
 * convolution on that pair.  The convolution then stages a tile's union rows once per 32-channel slice and runs the
 * offsets from LDS without barriers; weights come in MFMA-fragment order (cnrma_sparse_conv_prepare_weights_f16_frag,
 * cnrma_sparse_conv_f16_weight_bytes bytes).  Cin % 32 == 0, Cout >= 64.  workspace: as cnrma_sparse_conv_f32 (slabs of the
 * split over channel slices that short layers use).  tile_counters (may be NULL): ceil(no_cap / 64) * ceil(Cout / 64) zeroed
 * 32-bit words owned by the caller's stream; with them the last block of a tile adds the slabs up itself (same order, same
 * sums as the reduce launch it replaces) and leaves the words zero again.  Measured 2.5-3.5x slower than the reduce launch
 * (device-scope fences per block: an L2 write-back / invalidate each); the plugin passes NULL. */
size_t cnrma_sparse_tile_union_bytes(int64_t no_cap);
int cnrma_sparse_tile_union_build(const int32_t* nbr, int64_t no_cap, const int32_t* no_dev, int K, void* tile_union, void* stream);
int cnrma_sparse_conv_prepare_weights_f16_frag(const float* weight, int K, int Cin, int Cout, void* weight_frag, void* stream);
int cnrma_sparse_conv_go_f16x3(const float* in_feats, const float* in_amax, int Cin, const void* tile_union,
                               const void* weight_frag, int Cout, const float* scale, const float* shift,
                               const float* residual, int act, float* out_feats, float* out_amax, int64_t no_cap,
                               const int32_t* no_dev, void* workspace, size_t workspace_bytes, void* tile_counters,
                               void* stream);
/* What cnrma_sparse_conv_go_f16x3 launches for an output CAPACITY of no_cap rows (a pure host function of the sizes and the
 * workspace: captured launch sequences replay the same kernels; the tests prove variant coverage with it): out8 = {form (1 = the
 * round-5 kernel: metadata requested up front, cached row numbers, scalar offset loop; 0 = the round-4 kernel, kept for the
 * bit-identity test), tile columns (64: the block's four waves are 2 column tiles x 2 offset halves; 128: 4 column tiles),
 * splits over the 32-channel slices (> 1: partial slabs in the workspace + conv_reduce_kernel), slices per split, work order of
 * the one-dimensional grid (0 plain, 1 (column tile, split) groups -> XCDs, 2 row tiles -> XCDs), residual added inside the
 * kernel (0: by the reduce launch), blocks, weight offsets in flight per wave}.  No reference counterpart (ME picks its kernels
 * inside MinkowskiConvolution, fcaf3d_backbone.py:26-31). */
int cnrma_sparse_conv_go_plan(int64_t no_cap, int Cin, int Cout, size_t workspace_bytes, int has_residual, int* out8);
/* Exact-fp32 gather-once convolution (`v_mfma_f32_32x32x2_f32`): the reference's own arithmetic for the 3x3x3 stride-1
 * MinkowskiConvolutions (fcaf3d_backbone.py:26-31, :59-87; fcaf3d_head.py:61-83) on the tile unions of
 * cnrma_sparse_tile_union_build -- same operands / epilogue as cnrma_sparse_conv_f32, an fp32 fma chain per output in the order
 * (32-channel slice, offset group, offset, channel pair).  Weights in MFMA-fragment order, prepared once per layer
 * (cnrma_sparse_conv_prepare_weights_f32_frag into cnrma_sparse_conv_f32_frag_weight_bytes bytes: [K][Cin/32][Cout_p/32][4][64][4]
 * floats).  Cin % 32 == 0, Cout >= 64.  workspace: slabs of the split over channel slices (short layers), as
 * cnrma_sparse_conv_go_f16x3; cnrma_sparse_conv_go_plan describes the launch (the form / in-flight fields do not apply). */
size_t cnrma_sparse_conv_f32_frag_weight_bytes(int K, int Cin, int Cout);
int cnrma_sparse_conv_prepare_weights_f32_frag(const float* weight, int K, int Cin, int Cout, void* weight_frag, void* stream);
int cnrma_sparse_conv_go_f32(const float* in_feats, int Cin, const void* tile_union, const void* weight_frag, int Cout,
                             const float* scale, const float* shift, const float* residual, int act, float* out_feats,
                             int64_t no_cap, const int32_t* no_dev, void* workspace, size_t workspace_bytes, void* stream);
int cnrma_sparse_convtr_gen_f16x3(const int32_t* in_coords, const float* in_feats, const float* in_amax, int64_t n_cap,
                                  const int32_t* n_dev, int Cin, int half_stride, const void* weight_split, int Cout,
                                  const float* scale, const float* shift, int act, int32_t* out_coords,
                                  float* out_feats, float* out_amax, void* stream);

/* bf16 convolution (one bf16 piece per operand, round to nearest, fp32 accumulation: `v_mfma_f32_32x32x16_bf16`) -- the
 * precision of the reference's autocast training configuration (BASELINE configs[4]); NOT within the 1e-4 of the
 * inference path, which stays on f16x3 / f32.  Needs Cin % 32 == 0; weights prepared once:
 * fp32 [K][Cin][Cout] -> bf16 [K][Cin/32][Cout_p][32].  Same arguments / epilogue as cnrma_sparse_conv_f32 otherwise. */
size_t cnrma_sparse_conv_bf16_weight_bytes(int K, int Cin, int Cout);
int cnrma_sparse_conv_prepare_weights_bf16(const float* weight, int K, int Cin, int Cout, void* weight_bf16, void* stream);
/* the image of the data gradient's weights straight from W: Wd[k] = W[flip ? K - 1 - k : k]^T ([Cout] -> [Cin]; needs
 * Cout % 32 == 0; cnrma_sparse_conv_bf16_weight_bytes(K, Cout, Cin) bytes).  flip = 1: for a convolution whose output
 * coordinates ARE its input coordinates (odd kernel, stride 1) the transposed neighbour table is the table with its
 * offsets mirrored, nbr_t[i][k] = nbr[i][K - 1 - k] -- the data gradient runs on the forward table with these weights */
int cnrma_sparse_conv_prepare_weights_bf16_t(const float* weight, int K, int Cin, int Cout, int flip, void* weight_bf16,
                                             void* stream);
int cnrma_sparse_conv_bf16(const float* in_feats, int Cin, const int32_t* nbr, int K, const void* weight_bf16, int Cout,
                           const float* scale, const float* shift, const float* residual, int act, float* out_feats,
                           int64_t no_cap, const int32_t* no_dev, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the sparse convolution (training): what MinkowskiEngine's autograd runs under the reference's
 * train step (ray_marching.py:409-451 forward_train -> fcaf3d_backbone.py:59-107 / fcaf3d_head.py:61-139 convolutions ->
 * loss.backward() in mmcv's optimizer hook) for every ME.MinkowskiConvolution.
 * cnrma_sparse_kernel_map_transpose: nbr_t[n_in][K] with nbr_t[i][k] = o where nbr[o][k] == i (else -1); the data
 *   gradient is then the forward convolution of grad_out over nbr_t with the per-offset transposed weights.
 * cnrma_sparse_conv_wgrad_f32: slabs[chunk][K][Cin][Cout] = sum over the chunk's output rows o of
 *   in_feats[nbr[o][k]]^T (x) grad_out[o] (fp32 MFMA); the weight gradient is the sum of the
 *   cnrma_sparse_conv_wgrad_chunks(no_cap, rows_per_chunk) slabs.  nbr == NULL: identity map (K == 1).
 *   One block of 8 waves per (chunk, offset, 64 x 64 tile), the waves' partial tiles added in a fixed order: every
 *   element of every slab is written (no clearing needed), the result does not depend on scheduling; rows_per_chunk even,
 *   >= 256 keeps all waves busy. */
int cnrma_sparse_kernel_map_transpose(const int32_t* nbr, int64_t no_cap, const int32_t* no_dev, int K, int64_t n_in,
                                      int32_t* nbr_t, void* stream);
int cnrma_sparse_conv_wgrad_chunks(int64_t no_cap, int rows_per_chunk);
int cnrma_sparse_conv_wgrad_f32(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* grad_out, int Cout,
                                int64_t no_cap, const int32_t* no_dev, int rows_per_chunk, float* slabs, void* stream);
/* the same reduction with both operands rounded to bf16 (fp32 accumulation, v_mfma_f32_32x32x16_bf16): the weight
 * gradient of a torch.autocast(bfloat16) training step, as AMP computes it */
int cnrma_sparse_conv_wgrad_bf16(const float* in_feats, int Cin, const int32_t* nbr, int K, const float* grad_out, int Cout,
                                 int64_t no_cap, const int32_t* no_dev, int rows_per_chunk, float* slabs, void* stream);

/* (training, the same ME.MinkowskiConvolution backward; reference configuration: fp16 AMP of configs[4] -> bf16 autocast here)
 * the bf16 weight gradient of a 27-offset convolution on the gather-once structure of its forward
 * (cnrma_sparse_tile_union_build over the same neighbour table): slabs[part][27][Cin][Cout], part = a range of
 * ceil(tiles / parts) 64-row tiles; the weight gradient is the sum of the `parts` slabs, every element of which is written.
 * A block stages a tile's distinct input rows and its grad_out rows once in LDS (bf16, transposed) and runs 7 offsets over
 * them, one per wave; the accumulators live across the part's tiles.  Cin % 4 == Cout % 4 == 0.  Operands rounded to bf16,
 * fp32 accumulation: same products as cnrma_sparse_conv_wgrad_bf16, summed in another (fixed) order. */
int cnrma_sparse_conv_wgrad_go_bf16(const float* in_feats, int Cin, const void* tile_union, const float* grad_out, int Cout,
                                    int64_t no_cap, const int32_t* no_dev, int parts, float* slabs, void* stream);

/* (training: forward and data gradient of the BasicBlock / head 3x3x3 stride-1 ME.MinkowskiConvolution of
 * fcaf3d_backbone.py:59-87 and fcaf3d_head.py:61-83 under autocast)
 * the bf16 convolution (cnrma_sparse_conv_bf16's arithmetic: one bf16 piece per operand, fp32 accumulation) on the
 * gather-once structure of a 27-offset table (cnrma_sparse_tile_union_build): forward and data gradient of the autocast
 * training step.  No epilogue (training composes BatchNorm / activations in torch).  Weight image in MFMA-fragment order:
 *   transpose == 0: W [K][Cin][Cout] as it is (cnrma_sparse_conv_bf16_frag_weight_bytes(K, Cin, Cout) bytes);
 *   transpose == 1: the data gradient's weights W'[k] = W[flip ? K - 1 - k : k]^T (..._weight_bytes(K, Cout, Cin) bytes);
 * the image's input width % 32 == 0, its output width >= 64.  workspace (optional): no_cap x Cout x 4 x (Cin / 32) bytes let
 * short layers split over channel slices. */
size_t cnrma_sparse_conv_bf16_frag_weight_bytes(int K, int Cin, int Cout);
int cnrma_sparse_conv_prepare_weights_bf16_frag(const float* weight, int K, int Cin, int Cout, int transpose, int flip,
                                                void* weight_frag, void* stream);
/* both images of one weight tensor in one launch: the forward's (transpose 0) and the data gradient's (transpose 1, `flip`);
 * Cin % 32 == Cout % 32 == 0 */
int cnrma_sparse_conv_prepare_weights_bf16_frag_pair(const float* weight, int K, int Cin, int Cout, int flip,
                                                     void* frag_forward, void* frag_transposed, void* stream);
int cnrma_sparse_conv_go_bf16(const float* in_feats, int Cin, const void* tile_union, const void* weight_frag, int Cout,
                              float* out_feats, int64_t no_cap, const int32_t* no_dev, void* workspace, size_t workspace_bytes,
                              void* stream);

/* generative transposed convolution k=2 s=2 (fcaf3d_head.py:72-78): 8 children per parent, no overlap.
 * out_coords[8*i+k] = in_coords[i] + {0, half}^3 (k: x fastest); out_feats[8*i+k] = act((in[i] @ W[k])*scale+shift) */
int cnrma_sparse_convtr_gen_f32(const int32_t* in_coords, const float* in_feats, int64_t n_cap, const int32_t* n_dev,
                                int Cin, int half_stride, const float* weight, int Cout, const float* scale,
                                const float* shift, int act, int32_t* out_coords, float* out_feats, void* stream);

/* max pooling over the neighbour table (MinkowskiMaxPooling k=2 s=2, fcaf3d_backbone.py:31) */
int cnrma_sparse_maxpool_f32(const float* in_feats, int C, const int32_t* nbr, int K, float* out_feats,
                             int64_t no_cap, const int32_t* no_dev, void* stream);

/* MinkowskiInstanceNorm of ONE scene's row segment: per-channel mean / biased variance over the rows
 * [row0, row0 + n) -- row0 = *row0_dev (NULL: 0), n = min(n_cap, *n_dev) --, eps = 1e-8, affine (weight, bias),
 * optional ReLU (fcaf3d_backbone.py:29-30).  A multi-scene tensor (scene-major rows) takes one call per scene with the
 * scene's device-side offset / count.  stats_ws: cnrma_instnorm_workspace_bytes(C); on return stats_ws[0..C) = mean,
 * [C..2C) = biased variance (fp64). */
size_t cnrma_instnorm_workspace_bytes(int C);
int cnrma_sparse_instnorm_f32(const float* in_feats, int64_t n_cap, const int32_t* n_dev, const int32_t* row0_dev, int C,
                              const float* weight, const float* bias, float eps, int relu, float* out_feats,
                              double* stats_ws, void* stream);
/* The stem's InstanceNorm - ReLU - MaxPool (fcaf3d_backbone.py:29-31) without the normalised intermediate: out_feats == NULL in
 * cnrma_sparse_instnorm_f32 leaves only the statistics in stats_ws; this entry normalises the candidates of every pooling
 * window on the fly (same operations, same order: bit-identical to the two launches) and publishes max|out| (out_amax:
 * cnrma_amax_bytes() zeroed bytes, or NULL).  C % 4 == 0; nbr [no_cap][K] = the pooling table. */
int cnrma_sparse_instnorm_maxpool_f32(const float* in_feats, int C, const double* stats, const float* weight,
                                      const float* bias, float eps, int relu, const int32_t* nbr, int K, float* out_feats,
                                      int64_t no_cap, const int32_t* no_dev, float* out_amax, void* stream);

/* MinkowskiBatchNorm in training (nn.BatchNorm1d over the rows, fcaf3d_backbone.py / fcaf3d_head.py blocks): the forward is
 * cnrma_sparse_instnorm_f32 with the layer's eps (it leaves {mean, biased variance} in stats_ws); this is the backward:
 * grad_weight = sum dy * xhat, grad_bias = sum dy, grad_in = weight / sigma * (dy - mean(dy) - xhat * mean(dy * xhat)).
 * fp64 column sums in a fixed order (deterministic).  ws: cnrma_instnorm_workspace_bytes(C). */
/* BatchNorm1d in training mode fused with what follows it in a residual block (ME.MinkowskiBatchNorm + MinkowskiReLU of
 * ME's BasicBlock / Bottleneck, reached from fcaf3d_backbone.py:59-107 under the train step):
 *   out = act( (x - mean) / sqrt(var + eps) * weight + bias [+ residual] ),  C <= 256, C % 4 == 0, n >= 2;
 *   act (`relu` / `act`): 0 none, 1 ReLU, 2 ELU (alpha 1: the head's blocks, fcaf3d_head.py:61-83);
 * running_mean / running_var (unbiased estimate) / num_batches_tracked updated in place when given (momentum).
 * stats_ws: cnrma_instnorm_workspace_bytes(C); on return {mean[C], biased variance[C]} in fp64 = the backward's `stats`.
 * backward: y (the forward's output) != NULL applies the activation's derivative to grad_out (ReLU: y > 0; ELU: y + 1 where
 * y <= 0); grad_residual != NULL receives that
 * gradient.  fp64 column sums in a fixed order (deterministic). */
int cnrma_bn_train_forward_f32(const float* x, int64_t n, int C, const float* weight, const float* bias, float eps,
                               const float* residual, int relu, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float* out, double* stats_ws, void* stream);
int cnrma_bn_train_backward_f32(const float* grad_out, const float* x, const float* y, int act, int64_t n, int C, const double* stats,
                                const float* weight, float eps, float* grad_in, float* grad_residual, float* grad_weight,
                                float* grad_bias, double* ws, void* stream);
int cnrma_bn_backward_f32(const float* grad_out, const float* x, int64_t n, int C, const double* stats, const float* weight,
                          float eps, float* grad_in, float* grad_weight, float* grad_bias, double* ws, void* stream);

/* union-add of two sparse tensors at the same tensor stride (`inputs[i] + x`, fcaf3d_head.py:114):
 * output rows = all rows of A (in order) followed by the rows of B that are not in A.
 * a_hash maps A's coords -> A rows.  out_coords [Na+Nb cap][4], out_feats likewise; n_out device word.
 * On return the A hash additionally maps B-only coords -> their new rows (it becomes the union's map).
 * out_cap (>= na_cap): rows of the output buffers; B-only rows beyond it are dropped (n_out still counts them). */
size_t cnrma_union_workspace_bytes(int64_t nb);
int cnrma_sparse_union_add_f32(const int32_t* a_coords, const float* a_feats, int64_t na_cap, const int32_t* na_dev,
                               const int32_t* b_coords, const float* b_feats, int64_t nb_cap, const int32_t* nb_dev,
                               int C, uint64_t* a_hash_keys, int32_t* a_hash_vals, int64_t hash_cap,
                               int32_t* out_coords, float* out_feats, int64_t out_cap, int32_t* n_out, void* workspace,
                               void* stream);

/* features_at_coordinates (linear interpolation on the coarser lattice, fcaf3d_head.py:129):
 * out[i] = sum over the 8 corners c of floor(q/s)*s + {0,s}^3 of score[c] * prod_d (1 - |q_d - c_d| / s),
 * missing corners contribute 0.  q = query coords (int32 [N][4]), score [Ns][1]. */
int cnrma_sparse_interp_f32(const int32_t* q_coords, int64_t n_cap, const int32_t* n_dev, const float* score,
                            const uint64_t* s_hash_keys, const int32_t* s_hash_vals, int64_t hash_cap,
                            int score_stride, float* out, void* stream);

/* MinkowskiPruning by keep-mask (fcaf3d_head.py:138): rows kept in order. sel_index from cnrma_mask_to_index. */
int cnrma_sparse_prune_f32(const int32_t* in_coords, const float* in_feats, int64_t n_cap, const int32_t* n_dev,
                           int C, const int32_t* sel_index, int32_t* out_coords, float* out_feats, void* stream);

/* row-wise max over channels (scores.features.max(dim=1), fcaf3d_head.py:280) */
int cnrma_rowmax_f32(const float* in, int64_t n_cap, const int32_t* n_dev, int C, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * a12  box decoding   replaces FCAF3DHead._bbox_pred_to_bbox  fcaf3d_head.py:300-349 and the score product of
 *      _get_bboxes_single :249  (scores = sigmoid(cls) * sigmoid(centerness)).
 * points int32 coords [n][4] are scaled by voxel_size (:296); reg[n][R] holds exp()'d distances in 0..5 and raw
 * angle terms in 6.. ; yaw_mode: 0 = 6-DoF (R=6), 1 = 'fcaf3d' (R=8 -> 7), 2 = 'sin-cos' (R=8 -> 7), 3 = 'naive' (R=7).
 * ---------------------------------------------------------------------------------------------------------- */
int cnrma_fcaf3d_decode_f32(const float* points_xyz, const float* reg, int R, int64_t n, int yaw_mode,
                            float* boxes, void* stream);
/* fused variants used by the production path:
 * head_post: tail of FCAF3DHead.forward_single (fcaf3d_head.py:276-298) over the fused head GEMM output
 *   y[n][ldy] = [centerness | reg(R) | cls(n_cls) | pad] -> centerness[n], bbox_pred[n][R] (exp(scale*reg[:6]), raw
 *   angles), cls[n][n_cls], max_cls[n], points[n][3] = coords[:,1:] * voxel_size; scale = DEVICE scalar (Scale layer).
 * max_score: max_c sigmoid(cls)*sigmoid(centerness), the top-k key of :250-253.
 * select_decode: for the rows ids[0..k) (int64, NULL = identity): scores (:249) and decoded boxes (:300-349). */
int cnrma_fcaf3d_head_post_f32(const float* y, int ldy, const int32_t* coords, int64_t n, int R, int n_cls,
                               const float* scale, float voxel_size, float* centerness, float* bbox_pred, float* cls,
                               float* max_cls, float* points, void* stream);
int cnrma_fcaf3d_max_score_f32(const float* cls, const float* centerness, int64_t n, int n_cls, float* max_score,
                               void* stream);
int cnrma_fcaf3d_select_decode_f32(const int64_t* ids, int64_t k, const float* cls, const float* centerness,
                                   const float* reg, const float* points_xyz, int n_cls, int R, int yaw_mode,
                                   float* scores, float* boxes, void* stream);
int cnrma_fcaf3d_scores_f32(const float* cls, const float* centerness, int64_t n, int n_cls, float* scores,
                            float* max_score, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * "next" row 1: offline 3D NMS   replaces mmdet3d's pcdet_nms_gpu / pcdet_nms_normal_gpu as called by
 * post_process/nms_bbox.py:17-58 (third-party OpenPCDet iou3d_nms semantics: bird's-eye-view IoU, greedy in score order).
 * boxes_sorted [n][7] = (x, y, z, dx, dy, dz, heading), already in descending score order; rotated = 0 ignores heading.
 * mask [n][ceil(n/64)] uint64: bit j of row i set when j > i and IoU(i, j) > iou_thr (the host does the greedy scan).
 * cnrma_box_iou_f32: pairwise IoU matrix [na][nb]; mode3d = 1 multiplies the BEV overlap by the height overlap
 * (the IoU used by the mAP evaluation of post_process/evaluate_bbox.py through mmdet3d's indoor_eval).
 * ---------------------------------------------------------------------------------------------------------- */
int cnrma_nms_mask_f32(const float* boxes_sorted, int n, float iou_thr, int rotated, uint64_t* mask, void* stream);
int cnrma_box_iou_f32(const float* a, int na, const float* b, int nb, int rotated, int mode3d, float* iou, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNRMA_H */
