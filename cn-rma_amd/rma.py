"""Host mirror of the aggregation half of the reference's ray_marching.py.

Every function here drives the HIP kernels of csrc/ through the C-ABI (include/cnrma.h); torch only owns the
device buffers and the stream.  Names and argument meaning follow the reference:

    backproject / aggregate_2d_features / clear_3d_features   -> backproject_accum()
    get_ray_parameter                                         -> projection_inverse() + ray_params()
    ray_projection_neus / ray_projection_depth                -> rma_view_rows()  (reference layout, per view)
    aggregate_2d_features_ray_marching + switch_pointcloud    -> aggregate_points()  (fused production path)

Reference: projects/mvsdetection/models/ray_marching.py (line numbers in each docstring).
"""
import math

import numpy as np
import torch

from . import _lib
from . import plan as P
from ._lib import call, ptr, stream


SIGMOID_TABLE = True     # production march: table-driven kernel (False: the per-step sigmoid kernel, kept for A/B runs)
MARCH_SKIP = "auto"      # free-space skipping in the table-driven march (cnrma_rma_march_tables_f32): "auto" = from MARCH_SKIP_MIN_RAYS rays
                         # on (north-star shape, 12.3 M rays: 2.71 -> 1.26 ms with the table builds; ScanNet shape, 0.77 M rays on a
                         # stride-4 grid of maps: 0.30 -> 0.39 ms, the two extra launches cost more than the skipped steps save:
                         # profiles/r05_march_ab_*.log); True / False force it (A/B, parity tests)
MARCH_SKIP_MIN_RAYS = 4_000_000


def _f32(t):
    return t.contiguous().to(torch.float32)


def step_length(dims, voxel_size, n_steps):
    """t_one as the Python double of ray_marching.py:710-711."""
    X, Y, Z = dims
    return math.sqrt(X ** 2 + Y ** 2 + Z ** 2) * voxel_size / n_steps


def scale_projection(projection, stride):
    """rows 0-1 divided by backbone2d_stride (ray_marching.py:238-239 / :275-276). projection [...,3,4]."""
    p = projection.clone()
    p[..., :2, :] = p[..., :2, :] / stride
    return p


def to_nhwc(features, out=None):
    """features [V,C,H,W] (reference layout) -> channels-last [V,H,W,C] on the device (one HIP pass).
    A tensor whose MEMORY is already channels-last (torch.channels_last: what a 2D network run in that memory format
    writes) is returned as a view -- no pass at all (25 GB of traffic at the north-star shape)."""
    _lib.require_gpu()
    if (out is None and features.dim() == 4 and features.dtype == torch.float32 and features.is_cuda
            and features.permute(0, 2, 3, 1).is_contiguous()):
        return features.permute(0, 2, 3, 1)
    features = _f32(features)
    V, C, H, W = features.shape
    if out is None:
        out = torch.empty((V, H, W, C), dtype=torch.float32, device=features.device)
    assert out.shape == (V, H, W, C) and out.is_contiguous() and out.dtype == torch.float32
    call("cnrma_nchw_to_nhwc_f32", ptr(features), ptr(out), V, C, H, W, stream())
    return out


def is_channels_last(features):
    """[V,C,H,W] fp32 on the GPU whose MEMORY is channels-last (what a 2D network run in torch.channels_last writes): the
    hot path reads such maps in place"""
    return (torch.is_tensor(features) and features.dim() == 4 and features.dtype == torch.float32 and features.is_cuda
            and features.permute(0, 2, 3, 1).is_contiguous())


def backproject_accum(features_nhwc, projections, dims, voxel_size, origin, stride, proj_scaled=None, feat_ref=None, shape=None):
    """Dense unprojection of all views + mean (ray_marching.py:21-69, :220-257) in one kernel.

    features_nhwc [V,H,W,C] device fp32; projections [V,3,4] full-resolution (scaled here by `stride`), or
    proj_scaled [V,3,4] = scale_projection(projections, stride) already on the device (no host work at all).
    Returns volume [C,X,Y,Z] (mean over the views that see the voxel, 0 elsewhere) and count [X,Y,Z] int32;
    the reference's `valid` is `count > 0`.
    """
    _lib.require_gpu()
    X, Y, Z = dims
    if feat_ref is not None:
        # by reference: feat_ref = device int64 [1] holding the address of the [V,H,W,C] maps (shape given explicitly)
        V, H, W, C = shape
        dev = feat_ref.device
    else:
        V, H, W, C = features_nhwc.shape
        dev = features_nhwc.device
    proj = proj_scaled if proj_scaled is not None else _f32(scale_projection(projections.to(torch.float32), stride)).to(dev)
    volume = torch.empty((C, X, Y, Z), dtype=torch.float32, device=dev)
    count = torch.empty((X, Y, Z), dtype=torch.int32, device=dev)
    st = stream()
    ws = _dense_workspace(dev, st)
    if feat_ref is not None:
        call("cnrma_backproject_accum_ref_f32", ptr(feat_ref), ptr(proj), V, C, H, W, X, Y, Z, float(voxel_size),
             float(origin[0]), float(origin[1]), float(origin[2]), ptr(volume), ptr(count), ptr(ws), ws.numel() * 4, st)
    else:
        call("cnrma_backproject_accum_f32", ptr(features_nhwc), ptr(proj), V, C, H, W, X, Y, Z, float(voxel_size),
             float(origin[0]), float(origin[1]), float(origin[2]), ptr(volume), ptr(count), ptr(ws), ws.numel() * 4, st)
    return volume, count


_DENSE_WS = {}
_DENSE_KEYS = ("variant", "slab", "st", "zt", "tt", "zi", "chunk", "persist", "lpv", "pipe", "epi", "lockstep", "lattice", "nt", "own", "stagger", "groups", "ldspad")


def _dense_workspace(dev, st):
    """arrival counters of the dense kernel's lockstep schedules (debug / A-B only; the product schedule ignores them):
    one zeroed block per (device, stream).  The counters are monotonic -- never reset -- so calls on one stream simply
    keep counting"""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), st)
    ws = _DENSE_WS.get(key)
    if ws is None:
        if len(_DENSE_WS) >= 64:                  # streams come and go; their handles are all this table knows of them
            _DENSE_WS.pop(next(iter(_DENSE_WS)))
        ws = torch.zeros(256, dtype=torch.int32, device=dev)
        _DENSE_WS[key] = ws
    return ws


def dense_tuning(**kw):
    """debug / A-B aid (scripts/dense_ab.py, traversal-order tests): override schedule switches of the dense kernel;
    no arguments = the product configuration.  Never called by product code."""
    import ctypes
    if not kw:
        # back to the product library (libcnrma_hip.so has no tuning state, nothing to reset there)
        if _lib.experiments_active():
            call("cnrma_debug_dense_tuning", None, 0)
            _lib.experiments(False, "dense")
        return
    _lib.experiments(True, "dense")       # the alternative schedules exist in libcnrma_hip_exp.so only
    base = dict(variant=1, slab=1, st=16, zt=32, tt=8, zi=32, chunk=-1, persist=0, lpv=0, pipe=1, epi=0, lockstep=0, lattice=0, nt=0, own=0, stagger=0, groups=8, ldspad=0)
    base.update(DENSE_DEFAULTS)
    unknown = set(kw) - set(base)
    assert not unknown, unknown
    base.update(kw)
    arr = (ctypes.c_int * len(_DENSE_KEYS))(*[int(base[k]) for k in _DENSE_KEYS])
    call("cnrma_debug_dense_tuning", arr, len(_DENSE_KEYS))


DENSE_DEFAULTS = {}        # mirrors the non-trivial defaults of csrc/dense.hip's DenseTune (kept in step by a test)


class BackprojectAccum(torch.autograd.Function):
    """differentiable backproject_accum(): features NCHW [V,C,H,W] -> (volume [C,X,Y,Z], count [X,Y,Z]); the gradient of the
    volume flows to the feature maps (training of the Atlas 3D network together with the 2D backbone)."""

    @staticmethod
    def forward(ctx, features_nchw, projections, dims, voxel_size, origin, stride):
        nhwc = to_nhwc(features_nchw.detach())
        volume, count = backproject_accum(nhwc, projections, dims, voxel_size, origin, stride)
        ctx.args = (projections, tuple(dims), float(voxel_size), tuple(float(o) for o in origin), stride, nhwc.shape)
        ctx.count = count
        ctx.mark_non_differentiable(count)
        return volume, count

    @staticmethod
    def backward(ctx, grad_volume, _grad_count):
        projections, dims, vs, origin, stride, (V, H, W, C) = ctx.args
        dev = grad_volume.device
        proj = _f32(scale_projection(projections.to(torch.float32), stride)).to(dev)
        g = torch.empty((V, H, W, C), dtype=torch.float32, device=dev)
        call("cnrma_backproject_backward_f32", ptr(grad_volume.contiguous().float()), ptr(ctx.count), ptr(proj), V, C, H, W,
             dims[0], dims[1], dims[2], vs, origin[0], origin[1], origin[2], ptr(g), stream())
        return g.permute(0, 3, 1, 2).contiguous(), None, None, None, None, None


def backproject_index(projection_scaled, hw, dims, voxel_size, origin, device):
    """Debug/parity: rounded pixel (px, py) and validity of every voxel for ONE view (ray_marching.py:51-58)."""
    _lib.require_gpu()
    H, W = hw
    X, Y, Z = dims
    G = X * Y * Z
    proj = _f32(projection_scaled).to(device)
    px = torch.empty(G, dtype=torch.int32, device=device)
    py = torch.empty(G, dtype=torch.int32, device=device)
    valid = torch.empty(G, dtype=torch.uint8, device=device)
    call("cnrma_backproject_index_f32", ptr(proj), H, W, X, Y, Z, float(voxel_size), float(origin[0]),
         float(origin[1]), float(origin[2]), ptr(px), ptr(py), ptr(valid), stream())
    return px, py, valid


def projection_inverse(projections, stride):
    """[V,3,4] full-res -> [V,4,4] inverse of [P/stride; 0 0 0 1], computed with torch.inverse on the HOST in fp32,
    one matrix at a time, exactly as ray_marching.py:96-102 does, so that it is bit-identical to the reference's
    CPU run.  (V 4x4 LAPACK calls: microseconds; this is the only host arithmetic on the path.)"""
    p = scale_projection(projections.detach().to("cpu", torch.float32), stride)
    last = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]]).expand(p.shape[0], 1, 4)
    # one batched call: on CPU torch loops LAPACK getrf/getri over the matrices, i.e. bit-identical to the reference's
    # per-matrix loop (checked against the golden proj_inv in tests/test_oracle_cpu.py)
    return torch.inverse(torch.cat((p, last), dim=1))


def ray_params(proj_inv, H, W):
    """get_ray_parameter (ray_marching.py:71-111): o [V,3], d [V,3,H*W] on the device of proj_inv."""
    _lib.require_gpu()
    V = proj_inv.shape[0]
    o = torch.empty((V, 3), dtype=torch.float32, device=proj_inv.device)
    d = torch.empty((V, 3, H * W), dtype=torch.float32, device=proj_inv.device)
    call("cnrma_ray_params_f32", ptr(_f32(proj_inv)), V, H, W, ptr(o), ptr(d), stream())
    return o, d


class _March:
    """Argument pack shared by the count / emit calls of one scene."""

    def __init__(self, features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, mode, select_grids, shape=None,
                 device=None, feat_ref=None):
        """features_nhwc [V,H,W,C], or None with `shape` = (V,H,W,C) and `device`: the maps then come by reference
        (feat_ref: device int64 [1] holding their address) or are bound later (bind_features)"""
        self.feat = features_nhwc
        self.feat_ref = feat_ref
        self.V, self.H, self.W, self.C = features_nhwc.shape if features_nhwc is not None else shape
        self.dev = features_nhwc.device if features_nhwc is not None else torch.device(device)
        self.pinv = _f32(proj_inv).to(self.dev)
        self.tsdf = _f32(tsdf).to(self.dev)
        self.X, self.Y, self.Z = dims
        assert self.tsdf.numel() == self.X * self.Y * self.Z
        self.vs = float(voxel_size)
        self.org = [float(x) for x in origin]
        self.N = int(n_steps)
        self.t_one = float(step_length(dims, voxel_size, n_steps))
        self.thr = float(thr) if thr is not None else 0.0
        self.mode = mode
        self.k = int(select_grids or 0)
        self.R = self.V * self.H * self.W

    def count(self):
        cnt = torch.empty(self.R, dtype=torch.int32, device=self.dev)
        wsum = torch.empty(self.R, dtype=torch.float64, device=self.dev)
        head = (ptr(self.pinv), ptr(self.tsdf), self.V, self.H, self.W, self.X, self.Y, self.Z, self.vs, *self.org,
                self.N, self.t_one)
        if self.mode == "neus":
            call("cnrma_rma_neus_count_f32", *head, self.thr, ptr(cnt), ptr(wsum), stream())
        else:
            call("cnrma_rma_depth_count_f32", *head, self.k, ptr(cnt), ptr(wsum), stream())
        return cnt, wsum

    def kept_cap(self):
        """slots per ray for the kept-sample records: a ray's weights sum to <= 1, so it keeps <= 1/thr samples"""
        if self.mode != "neus" or not (self.thr > 1.0 / 62.0):
            return 0
        return int(1.0 / self.thr) + 2

    def march(self, layout_from=None, into=None):
        """phase 1 + kept-sample records (NeuS with thr > 1/62 only).  layout_from: the scene's NCHW feature maps -- the
        channels-last layout pass into self.feat then runs in the SAME launch as the march (cnrma_nchw_to_nhwc_march_f32:
        the HBM-bound copy hides the VALU-bound march).  into: (cnt, wsum, kept, overflow, table) buffers to (re)use."""
        cap = self.kept_cap()
        if into is not None:
            cnt, wsum, kept, overflow, tab = into
        else:
            cnt = torch.empty(self.R, dtype=torch.int32, device=self.dev)
            wsum = torch.empty(self.R, dtype=torch.float64, device=self.dev)
            kept = torch.empty((self.R, cap, 2), dtype=torch.int32, device=self.dev)
            overflow = torch.empty(4, dtype=torch.int32, device=self.dev)      # [0] violations; [1..2] scratch of the fused launch
            tab = torch.empty_like(self.tsdf) if SIGMOID_TABLE else None
        overflow_all, overflow = overflow, overflow[:1]
        self._overflow = overflow
        skip = None
        if tab is not None:      # sigmoid(-tsdf) once per voxel instead of once per marched step (bit-identical) + the free-space
            if (self.R >= MARCH_SKIP_MIN_RAYS) if MARCH_SKIP == "auto" else bool(MARCH_SKIP):   # skip table (radius per 4^3 block)
                if getattr(self, "_skip", None) is None:
                    self._skip = torch.empty(_lib.load().cnrma_rma_skip_table_bytes(self.X, self.Y, self.Z), dtype=torch.uint8,
                                             device=self.dev)
                skip = self._skip
            call("cnrma_rma_march_tables_f32", ptr(self.tsdf), self.X, self.Y, self.Z, ptr(tab), ptr(skip), stream())
        tail = (ptr(self.pinv), ptr(self.tsdf), ptr(tab), self.V, self.H, self.W, self.X, self.Y, self.Z, self.vs, *self.org,
                self.N, self.t_one, self.thr, ptr(cnt), ptr(wsum), ptr(kept), cap, ptr(overflow_all), ptr(skip), stream())
        if layout_from is not None:
            src = _f32(layout_from)
            assert tuple(src.shape) == (self.V, self.C, self.H, self.W) and self.feat is not None and self.feat.is_contiguous()
            call("cnrma_nchw_to_nhwc_march_f32", ptr(src), ptr(self.feat), self.C, *tail)
        else:
            call("cnrma_rma_neus_march_f32", *tail)
        return cnt, wsum, kept, overflow

    def march_buffers(self):
        """persistent outputs of march() for a static slot: (cnt, wsum, kept, overflow, table)"""
        cap = self.kept_cap()
        return (torch.empty(self.R, dtype=torch.int32, device=self.dev), torch.empty(self.R, dtype=torch.float64, device=self.dev),
                torch.empty((self.R, cap, 2), dtype=torch.int32, device=self.dev),
                torch.empty(4, dtype=torch.int32, device=self.dev), torch.empty_like(self.tsdf) if SIGMOID_TABLE else None)

    def emit_rows(self, row_offset, n_out, kept, sel_index, w_div, add, out_xyz, xyz_stride, out_w, w_stride, out_feat,
                  feat_stride, out_sample=None, n_out_dev=None):
        """n_out = rows of the output buffers; n_out_dev = device word with the live row count (None: all n_out)"""
        rec = torch.empty((int(n_out), 4), dtype=torch.int32, device=self.dev)
        by_ref = self.feat_ref is not None
        call("cnrma_rma_neus_emit_rows_ref_f32" if by_ref else "cnrma_rma_neus_emit_rows_f32", ptr(self.pinv),
             ptr(self.feat_ref) if by_ref else ptr(self.feat), self.V, self.C, self.H, self.W, self.N,
             self.t_one, ptr(row_offset), int(n_out), ptr(n_out_dev), ptr(kept), kept.shape[1], ptr(sel_index),
             sel_index.numel() if sel_index is not None else 0, ptr(rec), ptr(w_div), float(add[0]),
             float(add[1]), float(add[2]), out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride, ptr(out_sample),
             stream())

    def emit_records(self, rec, n_out, n_out_dev, w_div, add, out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride,
                     out_sample=None):
        """emit_rows for records that are already selected and placed (select_records)"""
        by_ref = self.feat_ref is not None
        call("cnrma_rma_neus_emit_rows_ref_f32" if by_ref else "cnrma_rma_neus_emit_rows_f32", ptr(self.pinv),
             ptr(self.feat_ref) if by_ref else ptr(self.feat), self.V, self.C, self.H, self.W, self.N,
             self.t_one, None, int(n_out), ptr(n_out_dev), None, 0, None, 0, ptr(rec), ptr(w_div), float(add[0]),
             float(add[1]), float(add[2]), out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride, ptr(out_sample),
             stream())

    def emit(self, row_offset, sel_index, w_div, add, out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride,
             out_sample=None, sel_cap=0, out_cap=0):
        head = (ptr(self.pinv), ptr(self.tsdf), ptr(self.feat), self.V, self.C, self.H, self.W, self.X, self.Y, self.Z,
                self.vs, *self.org, self.N, self.t_one)
        tail = (ptr(row_offset), ptr(sel_index), ptr(w_div), float(add[0]), float(add[1]), float(add[2]),
                out_xyz, xyz_stride, out_w, w_stride, out_feat, feat_stride)
        if self.mode == "neus":
            call("cnrma_rma_neus_emit_f32", *head, self.thr, *tail, ptr(out_sample), stream())
        else:
            call("cnrma_rma_depth_emit_f32", *head, self.k, *tail, int(sel_cap), int(out_cap), stream())


def exclusive_scan(count):
    """int32 [n] -> int32 [n+1] exclusive prefix sums (last entry = total), on the device."""
    n = count.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=count.device)
    ws = torch.empty(_lib.load().cnrma_scan_workspace_bytes(n), dtype=torch.uint8, device=count.device)
    call("cnrma_exclusive_scan_i32", ptr(count), ptr(out), n, ptr(ws), stream())
    return out


def mask_to_index(mask_u8):
    """uint8 [n] keep-mask -> (sel_index int32 [n], n_sel int32 [1]) on the device."""
    n = mask_u8.numel()
    sel = torch.empty(n, dtype=torch.int32, device=mask_u8.device)
    n_sel = torch.empty(1, dtype=torch.int32, device=mask_u8.device)
    ws = torch.empty(_lib.load().cnrma_scan_workspace_bytes(n), dtype=torch.uint8, device=mask_u8.device)
    call("cnrma_mask_to_index", ptr(mask_u8), ptr(sel), ptr(n_sel), n, ptr(ws), stream())
    return sel, n_sel


_SAMPLE_CALLS = [0]


def sample_mask_device(m_dev, M, n_keep, seed=None, seed_dev=None):
    """uint8 [M] keep-mask with exactly min(m, n_keep) ones among the first m = min(m_dev[0], M) rows (zeros behind):
    a uniformly random subset drawn ON THE DEVICE (device-side stand-in for sample_points' np.random.choice;
    deterministic in `seed`, advancing per call when seed is None; seed_dev: device word mixed into the seed)."""
    if seed is None:
        _SAMPLE_CALLS[0] += 1
        seed = (0x9E3779B9 * _SAMPLE_CALLS[0] + int(torch.initial_seed())) & 0xFFFFFFFF
    mask = torch.empty(M, dtype=torch.uint8, device=m_dev.device)
    ws = torch.empty(_lib.load().cnrma_sample_workspace_bytes(), dtype=torch.uint8, device=m_dev.device)
    call("cnrma_sample_mask", m_dev.data_ptr(), M, int(n_keep), int(seed), ptr(seed_dev), ptr(mask), ptr(ws), stream())
    return mask


def select_records(row_offset, kept, m_dev, M, n_keep, rec_cap, seed=None, seed_dev=None):
    """sample_mask_device + mask_to_index + the record scatter in one go, per ray (cnrma_rma_select_records): the same random
    subset, records int32 [rec_cap, 4] = {ray, step, weight bits, 0} in row order and their number n_sel int32 [1]"""
    if seed is None:
        _SAMPLE_CALLS[0] += 1
        seed = (0x9E3779B9 * _SAMPLE_CALLS[0] + int(torch.initial_seed())) & 0xFFFFFFFF
    dev = m_dev.device
    R = row_offset.numel() - 1
    lib = _lib.load()
    rec = torch.empty((int(rec_cap), 4), dtype=torch.int32, device=dev)
    n_sel = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.cnrma_sample_workspace_bytes(), dtype=torch.uint8, device=dev)
    cnt = torch.empty(R, dtype=torch.int32, device=dev)
    off = torch.empty(R + 1, dtype=torch.int32, device=dev)
    sws = torch.empty(lib.cnrma_scan_workspace_bytes(R), dtype=torch.uint8, device=dev)
    call("cnrma_rma_select_records", ptr(row_offset), R, ptr(kept), kept.shape[1], m_dev.data_ptr(), int(M), int(n_keep),
         int(seed), ptr(seed_dev), ptr(ws), ptr(cnt), ptr(off), ptr(sws), int(rec_cap), ptr(rec), ptr(n_sel), stream())
    return rec, n_sel


def _drop_single_sample_views(cnt, wsum, V):
    """Reference quirk: a view that keeps exactly ONE sample is dropped entirely -- torch.squeeze() makes the
    index 0-dim, len() raises and the bare except skips the view (ray_marching.py:781-782, :282-287)."""
    call("cnrma_rma_drop_single_sample_views", ptr(cnt), ptr(wsum), int(V), cnt.numel() // int(V), stream())


def rma_view_rows(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps=300, thr=0.05, mode="neus",
                  select_grids=0, with_samples=False, single_march=None):
    """Raw rows in the reference's own layout [M, 3+1+C] = [x,y,z,w,feat] for ALL given views, view-major
    (ray_projection_neus :687-807 / ray_projection_depth :809-956), plus per-view row counts [V].
    With `with_samples` also returns int32 [M,2] = (ray index over all views, step) per row."""
    _lib.require_gpu()
    m = _March(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, mode, select_grids)
    if single_march is None:
        single_march = m.kept_cap() > 0
    if single_march:
        cnt, wsum, kept, overflow = m.march()
    else:
        cnt, wsum = m.count()
    off = exclusive_scan(cnt)
    M = int(off[-1].item())
    per_view = cnt.view(m.V, -1).sum(dim=1)
    Wd = 4 + m.C
    rows = torch.empty((M, Wd), dtype=torch.float32, device=m.dev)
    samples = torch.empty((M, 2), dtype=torch.int32, device=m.dev) if with_samples else None
    if M > 0:
        base = rows.data_ptr()
        if single_march:
            assert int(overflow.item()) == 0
            m.emit_rows(off, M, kept, None, None, (0.0, 0.0, 0.0), base, Wd, base + 12, Wd, base + 16, Wd, samples)
        else:
            m.emit(off, None, None, (0.0, 0.0, 0.0), base, Wd, base + 12, Wd, base + 16, Wd, samples)
    return (rows, per_view, samples) if with_samples else (rows, per_view)


def aggregate_points(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps=300, thr=0.05, mode="neus",
                     select_grids=0, offset=(0.0, 0.0, 0.0), max_points=None, sampler="numpy", mask=None,
                     reference_quirks=True, seed=None):
    """Fused aggregate_2d_features_ray_marching (:260-307) + switch_pointcloud test path (:339-407).

    Returns (coords [Ms,3], feats [Ms,C], info).  feats = feature * (w / mean(w)) with the mean over ALL M rows of
    the scene (:303); coords = place + offset (:364); when max_points is set and M > max_points only the rows of
    the keep-mask are written (sample_points, fcaf3d_transforms.py:283-296), order preserved.

    sampler: "numpy"  -- the mask comes from numpy's global RNG exactly like the reference (host, bit-parity);
             "device" -- a uniformly random mask with exactly max_points ones drawn on the GPU (same
                         distribution, different RNG stream; no host RNG on the critical path).
    mask:    optional explicit keep-mask (numpy bool / torch uint8 of length M); overrides `sampler`.
    """
    st = aggregate_begin(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, mode, select_grids,
                         reference_quirks)
    # the one read-back the reference also has (nonzero(), :781): total row count (+ the record-overflow guard)
    return aggregate_finish(st, _lib.read_ints(st["readback"]), offset, max_points, sampler, mask, seed)


def aggregate_begin(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps=300, thr=0.05, mode="neus",
                    select_grids=0, reference_quirks=True):
    """first half of aggregate_points: march, row offsets, mean weight -- everything up to the read-back of the row
    count.  st["readback"] (int32 [1 or 2] on the device) must be read by the caller (several scenes can share one
    device->host read: torch.cat of their read-backs) and handed to aggregate_finish()."""
    _lib.require_gpu()
    m = _March(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, mode, select_grids)
    single_march = m.kept_cap() > 0
    kept = overflow = None
    if single_march:
        cnt, wsum, kept, overflow = m.march()
    else:
        cnt, wsum = m.count()
    if reference_quirks:
        _drop_single_sample_views(cnt, wsum, m.V)
    off = exclusive_scan(cnt)
    ws = torch.empty(_lib.load().cnrma_scan_workspace_bytes(m.R), dtype=torch.uint8, device=m.dev)
    wtot = torch.empty(1, dtype=torch.float64, device=m.dev)
    mean_w = torch.empty(1, dtype=torch.float32, device=m.dev)
    call("cnrma_sum_f64", ptr(wsum), ptr(wtot), m.R, ptr(ws), stream())
    m_total = off[m.R:]
    call("cnrma_rma_mean_weight", ptr(wtot), m_total.data_ptr(), ptr(mean_w), stream())
    readback = torch.cat((m_total, overflow)) if single_march else m_total
    return dict(m=m, single_march=single_march, cnt=cnt, off=off, kept=kept, mean_w=mean_w, m_total=m_total,
                readback=readback)


def aggregate_finish(st, readback, offset=(0.0, 0.0, 0.0), max_points=None, sampler="numpy", mask=None, seed=None):
    """second half of aggregate_points; readback = st["readback"] as host ints; seed: of the device sampler (None =
    a fresh one per call)"""
    m, single_march, off, kept, mean_w, m_total, cnt = (st[k] for k in ("m", "single_march", "off", "kept", "mean_w",
                                                                         "m_total", "cnt"))
    M = int(readback[0])
    if single_march and readback[1]:
        raise _lib.CnrmaError("kept-sample record overflow: a ray kept more than 1/thr samples")
    if M == 0:
        raise TypeError("no valid points in any view (ray_marching.py:300)")
    if P.current() is not None:
        P.current().record(M)
    sel = None
    Ms = M
    if mask is not None or (max_points is not None and M > max_points):
        Ms = None
        if mask is None:
            Ms = int(max_points)
            if sampler == "numpy":
                mask = np.zeros(M, dtype=bool)
                mask[np.random.choice(M, max_points, replace=False)] = True
            elif sampler == "device":
                mask = sample_mask_device(m_total, M, max_points, seed=seed)
            else:
                raise ValueError(f"unknown sampler {sampler!r}")
        if isinstance(mask, np.ndarray):
            if Ms is None:
                Ms = int(mask.sum())
            mask = torch.from_numpy(mask.astype(np.uint8))
        mask = mask.to(device=m.dev, dtype=torch.uint8).contiguous()
        assert mask.numel() == M, "mask length must equal the number of aggregated rows"
        sel, n_sel = mask_to_index(mask)
        if Ms is None:
            Ms = int(n_sel.item())
    coords = torch.empty((Ms, 3), dtype=torch.float32, device=m.dev)
    feats = torch.empty((Ms, m.C), dtype=torch.float32, device=m.dev)
    if single_march:
        m.emit_rows(off, Ms, kept, sel, mean_w, offset, coords.data_ptr(), 3, None, 0, feats.data_ptr(), m.C)
    else:
        m.emit(off, sel, mean_w, offset, coords.data_ptr(), 3, None, 0, feats.data_ptr(), m.C)
    info = dict(M=M, M_selected=Ms, mean_w=mean_w, row_offset=off, count=cnt, kept=kept, sel=sel, march=m)
    return coords, feats, info


def aggregate_points_static(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps=300, thr=0.05,
                            offset=(0.0, 0.0, 0.0), max_points=None, seed=0, seed_dev=None, reference_quirks=True,
                            marched=None, mode="neus", select_grids=0, feat_ref=None, shape=None, defer_feats=False):
    """aggregate_points() without a device->host read (the static trace of plan.Plan; NeuS single-march only): the row
    count M stays on the device, the selection always goes through the device sampler (it keeps every row when
    M <= max_points) and the outputs are capacity-sized.  Returns (coords [cap,3], feats [cap,C], n_dev int32 [1], info);
    rows >= n_dev[0] are undefined.  defer_feats (NeuS): only the places are emitted, feats is None and info["rec"] holds the
    point records -- the caller carries them through the voxeliser and emits the features of the surviving rows in place
    (emit_point_features), or all of them on demand (point_features)."""
    plan = P.current()
    assert plan is not None and plan.static
    depth = mode == "depth"
    kept = overflow = None
    if marched is not None:          # (march object, its outputs): the march already ran (with the layout pass, outside the trace)
        m, (cnt, wsum, kept, overflow) = marched[0], marched[1][:4]
    elif depth:                      # ray_projection_depth (:809-956): a fixed number of rows per ray with a sign change
        m = _March(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, 0.0, "depth", select_grids)
        cnt, wsum = m.count()
    else:
        m = _March(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, "neus", 0, shape=shape,
                   device=proj_inv.device, feat_ref=feat_ref)
        if m.kept_cap() <= 0:
            raise _lib.CnrmaError("the static trace needs the single-march NeuS path (thr > 1/62)")
        cnt, wsum, kept, overflow = m.march()
    if reference_quirks:
        _drop_single_sample_views(cnt, wsum, m.V)
    off = exclusive_scan(cnt)
    ws = torch.empty(_lib.load().cnrma_scan_workspace_bytes(m.R), dtype=torch.uint8, device=m.dev)
    wtot = torch.empty(1, dtype=torch.float64, device=m.dev)
    mean_w = torch.empty(1, dtype=torch.float32, device=m.dev)
    call("cnrma_sum_f64", ptr(wsum), ptr(wtot), m.R, ptr(ws), stream())
    m_total = off[m.R:]
    call("cnrma_rma_mean_weight", ptr(wtot), m_total.data_ptr(), ptr(mean_w), stream())
    rows_per_ray = max(1, 2 * m.k) if depth else kept.shape[1]
    M_cap = plan.next_cap(bound=m.R * rows_per_ray, n_dev=m_total, lo=1)      # lo = 1: M == 0 is the reference's TypeError
    if overflow is not None:
        plan.watch(overflow, 0, 0)
    n_keep = int(max_points) if max_points is not None else M_cap
    cap = min(M_cap, n_keep)
    defer_feats = bool(defer_feats) and not depth
    coords = torch.empty((cap, 3), dtype=torch.float32, device=m.dev)
    feats = None if defer_feats else torch.empty((cap, m.C), dtype=torch.float32, device=m.dev)
    sel = rec = None
    if depth:
        mask = sample_mask_device(m_total, M_cap, n_keep, seed=seed, seed_dev=seed_dev)
        sel, n_sel = mask_to_index(mask)
        m.emit(off, sel, mean_w, offset, coords.data_ptr(), 3, None, 0, feats.data_ptr(), m.C, sel_cap=M_cap, out_cap=cap)
    else:
        # the subset is drawn per ray from the sample records: no M-sized mask, no M-sized index (72 M rows at the north star)
        rec, n_sel = select_records(off, kept, m_total, M_cap, n_keep, cap, seed=seed, seed_dev=seed_dev)
        m.emit_records(rec, cap, n_sel, mean_w, offset, coords.data_ptr(), 3, None, 0,
                       feats.data_ptr() if feats is not None else None, m.C)
    info = dict(M=m_total, M_selected=n_sel, mean_w=mean_w, row_offset=off, count=cnt, kept=kept, sel=sel, march=m, rec=rec)
    return coords, feats, n_sel, info


def emit_point_features(info, rec, n_rows, n_dev, out=None, amax=None):
    """feats [n_rows, C] of the point records `rec` (int32 [n_rows, 4], any order): feat[ray] * w / mean(w), the arithmetic of
    the row emission; amax: zeroed magnitude-bound slots that receive max|feats| (cnrma_rma_emit_features_f32)"""
    m = info["march"]
    if out is None:
        out = torch.empty((int(n_rows), m.C), dtype=torch.float32, device=m.dev)
    by_ref = m.feat_ref is not None
    call("cnrma_rma_emit_features_f32", None if by_ref else ptr(m.feat), ptr(m.feat_ref) if by_ref else None, m.C, ptr(rec),
         int(n_rows), ptr(n_dev), ptr(info["mean_w"]), ptr(out), out.shape[1], ptr(amax), stream())
    return out


def point_features(info):
    """the [cap, C] feature rows of a deferred aggregation, in point order (what feats would have been)"""
    rec = info["rec"]
    return emit_point_features(info, rec, rec.shape[0], info["M_selected"])


def aggregate_points_backward(info, grad_feats):
    """gradient of aggregate_points()'s `feats` output w.r.t. the NHWC feature maps [V,H,W,C] (the weights and places
    carry no gradient, ray_marching.py:705): needs the forward's `info` (single-march path)."""
    m, kept = info["march"], info["kept"]
    if kept is None:
        raise NotImplementedError("the backward needs the kept-sample records of the single-march forward (NeuS, thr > 1/62)")
    g = grad_feats.contiguous().float()
    out = torch.empty((m.V, m.H, m.W, m.C), dtype=torch.float32, device=m.dev)
    call("cnrma_rma_neus_rows_backward_f32", ptr(g), g.shape[1], m.V, m.C, m.H, m.W, ptr(info["row_offset"]), ptr(kept),
         kept.shape[1], ptr(info["sel"]), ptr(info["mean_w"]), ptr(out), stream())
    return out


class AggregatePoints(torch.autograd.Function):
    """differentiable aggregate_points(): (features NCHW [V,C,H,W], ...) -> (coords [Ms,3], feats [Ms,C]); the gradient
    flows from `feats` to the feature maps (and on into the 2D backbone), like the reference's indexing
    features[b, :, v, u] (ray_marching.py:793-797)."""

    @staticmethod
    def forward(ctx, features_nchw, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, offset, max_points, sampler, mask):
        nhwc = to_nhwc(features_nchw.detach())
        coords, feats, info = aggregate_points(nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, "neus", 0,
                                               offset, max_points, sampler, mask)
        ctx.info = info
        ctx.mark_non_differentiable(coords)
        return coords, feats

    @staticmethod
    def backward(ctx, _grad_coords, grad_feats):
        g = aggregate_points_backward(ctx.info, grad_feats)                 # [V,H,W,C]
        return (g.permute(0, 3, 1, 2).contiguous(),) + (None,) * 11


def aggregate_rows(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps=300, thr=0.05, mode="neus",
                   select_grids=0, reference_quirks=True):
    """points_detection of the reference after :298-307: [M, 3+C] = [xyz, feat * w/mean(w)] (no offset, no mask)."""
    coords, feats, info = aggregate_points(features_nhwc, proj_inv, tsdf, dims, voxel_size, origin, n_steps, thr, mode,
                                           select_grids, reference_quirks=reference_quirks)
    return torch.cat((coords, feats), dim=1), info


def select_rows(points, offset, mask=None):
    """switch_pointcloud test path on an existing [M,3+C] matrix (ray_marching.py:360-405)."""
    _lib.require_gpu()
    points = _f32(points)
    M, Wd = points.shape
    C = Wd - 3
    sel = None
    Ms = M
    if mask is not None:
        if isinstance(mask, np.ndarray):
            mask = torch.from_numpy(mask.astype(np.uint8))
        mask = mask.to(device=points.device, dtype=torch.uint8).contiguous()
        sel, _ = mask_to_index(mask)
        Ms = int(mask.sum().item())
    coords = torch.empty((Ms, 3), dtype=torch.float32, device=points.device)
    feats = torch.empty((Ms, C), dtype=torch.float32, device=points.device)
    call("cnrma_select_rows_f32", ptr(points), M, C, ptr(sel), float(offset[0]), float(offset[1]), float(offset[2]),
         ptr(coords), ptr(feats), stream())
    return coords, feats
