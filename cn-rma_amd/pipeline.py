"""The hot path end to end for one scene (SURVEY.md 8d metric):

    features [V,C,H,W] + projections [V,3,4] + TSDF [X,Y,Z]
      -> dense unprojection-accumulate (a1-a3)  -> volume, count          (input of the Atlas 3D CNN, out of scope)
      -> ray-marching aggregation (a4-a7) fused with the point selection (a8)
      -> voxelisation (a9) -> FCAF3D backbone (a10) -> neck + head (a11) -> decode (a12) -> raw boxes + scores

Mirrors the order of RayMarching.forward_test (projects/mvsdetection/models/ray_marching.py:456-521) with the 2D
backbone and the Atlas reconstruction network replaced by their outputs (features, TSDF) as inputs.
"""
import torch

from . import _lib, rma
from . import plan as P
from . import sparse as S


DEFER_POINT_FEATURES = True     # static trace: point features are emitted once, in voxel order, into the sparse tensor (trace_net)
DENSE_BRANCH = False     # StaticScene: the dense unprojection as a parallel branch of the scene graph (fork / join inside the
                         # capture).  Measured (bench.py --dense-branch 1): NS 40.8 vs 43.0 scenes/s, S 163 vs 202 -- a graph with a
                         # second stream in it replays slower than the linear chain, and the chip-filling kernel gains nothing


class SceneConfig:
    def __init__(self, dims, voxel_size=0.04, origin=(0.0, 0.0, 0.0), stride=4, n_steps=300, thr=0.05,
                 max_points=500000, voxel_size_fcaf3d=0.01, ray_marching_type="neus", depth_points=None,
                 sampler="device", sample_seed=None):
        self.dims = tuple(dims)
        self.voxel_size = voxel_size
        self.origin = tuple(origin)
        self.stride = stride
        self.n_steps = n_steps
        self.thr = thr
        self.max_points = max_points
        self.voxel_size_fcaf3d = voxel_size_fcaf3d
        self.ray_marching_type = ray_marching_type
        self.depth_points = depth_points
        self.sampler = sampler
        self.sample_seed = sample_seed        # device sampler: fixed seed (None = a fresh subset per call)


def graph_node_count(graph):
    """number of nodes of a captured torch.cuda.CUDAGraph (hipGraphGetNodes on its raw handle; the graph must have been
    created with keep_graph=True); None when the runtime does not give it"""
    import ctypes
    import os
    try:
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        n = ctypes.c_size_t(0)
        hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        hip.hipGraphGetNodes.restype = ctypes.c_int
        if hip.hipGraphGetNodes(ctypes.c_void_p(graph.raw_cuda_graph()), None, ctypes.byref(n)) != 0:
            return None
        return int(n.value)
    except Exception:       # noqa: BLE001 -- a diagnostic: never fatal
        return None


def weight_tensors(*modules):
    """every parameter and buffer of the modules, as a list (built once per captured context)"""
    import itertools
    return [t for m in modules for t in itertools.chain(m.parameters(), m.buffers())]


def weights_tag(tensors):
    """cheap fingerprint of a model's weights: (count, sum of in-place version counters, first storage address).  Optimiser
    steps, load_state_dict() and every other in-place write bump a version; a moved / re-created model changes the
    address.  (Writes through `.data` bump nothing: call the owner's reset after those.)  A captured graph replays
    prepared weight images, so its owner compares this tag before each replay."""
    v = 0
    for t in tensors:
        v += t._version
    return (len(tensors), v, tensors[0].data_ptr() if tensors else 0)


def _offset_list(offset):
    if offset is None:
        return (0.0, 0.0, 0.0)
    return tuple(float(x) for x in torch.as_tensor(offset, dtype=torch.float32).detach().reshape(3).cpu().tolist())


class StageTimer:
    """optional per-stage HIP-event timing on the current stream"""

    def __init__(self, enabled):
        self.enabled = enabled
        self.marks = []

    def mark(self, name):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def result(self):
        if not self.enabled:
            return {}
        torch.cuda.synchronize()
        out = {}
        for (_, a), (n, b) in zip(self.marks[:-1], self.marks[1:]):
            out[n] = out.get(n, 0.0) + a.elapsed_time(b)
        return out


@torch.no_grad()
def forward_scenes(cfg, backbone, head, scenes):
    """Several scenes through ONE sparse network pass.  scenes: list of dicts(features [V,C,H,W], projection [V,3,4],
    tsdf [X,Y,Z], offset=(0,0,0)).  The geometric half (dense unprojection, ray marching, selection, voxelisation)
    runs per scene; the scenes' voxels are then collated into one multi-scene sparse tensor (batch id = list index,
    ray_marching.py:328-330 builds exactly that for B samples) for the backbone + neck/head, whose ~300 small launches
    and split-K tails are thereby shared; decode is per scene.  Returns one dict per scene like forward_scene()."""
    parts, infos, states = [], [], []
    for sc_ in scenes:                                   # phase 1: everything up to the row-count read-back, all scenes
        feats = rma.to_nhwc(sc_["features"])
        volume, count = rma.backproject_accum(feats, sc_["projection"], cfg.dims, cfg.voxel_size, cfg.origin, cfg.stride)
        proj_inv = rma.projection_inverse(sc_["projection"], cfg.stride).to(feats.device, non_blocking=True)
        states.append(rma.aggregate_begin(feats, proj_inv, sc_["tsdf"], cfg.dims, cfg.voxel_size, cfg.origin, cfg.n_steps,
                                          cfg.thr, cfg.ray_marching_type, cfg.depth_points))
        infos.append(dict(volume=volume, count=count))
    reads = _lib.read_ints(torch.cat([st["readback"] for st in states]))       # ONE device->host read for all scenes
    w = len(reads) // len(scenes)
    for b, (sc_, st) in enumerate(zip(scenes, states)):    # phase 2: selection + emission
        coords, pfeats, info = rma.aggregate_finish(st, reads[b * w:(b + 1) * w], sc_.get("offset", (0.0, 0.0, 0.0)),
                                                    cfg.max_points, cfg.sampler, sc_.get("mask"))
        parts.append((coords, pfeats))
        infos[b].update(info)
    x = S.sparse_collate(parts, cfg.voxel_size_fcaf3d)
    levels = backbone(x)
    cen, box, cls, pts, scn = map(list, head(levels, fused=True))
    dets = head.get_bboxes_fused(cen, box, cls, pts, scn, len(scenes))
    outs = []
    for b, (bboxes, scores) in enumerate(dets):
        outs.append(dict(bboxes=bboxes, scores=scores, M=infos[b]["M"], M_selected=infos[b]["M_selected"],
                         M_unique=x.cs.batch_counts()[b], volume=infos[b]["volume"], count=infos[b]["count"],
                         level_rows=[len(l) for l in levels], head_rows=[len(c[0]) for c in cen]))
    return outs


@torch.no_grad()
def forward_scene(cfg, backbone, head, features_nchw, projections, tsdf, offset=(0.0, 0.0, 0.0), dense=True,
                  timing=False, mask=None, proj_inv=None):
    """One scene forward.  features_nchw [V,C,H,W] device fp32 (the 2D backbone's layout), projections [V,3,4]
    (full-resolution pixel units; a host copy avoids a D2H), tsdf [X,Y,Z] device.  Returns a dict with the decoded
    boxes/scores and the intermediate sizes needed to recompute the algorithmic bytes."""
    tm = StageTimer(timing)
    tm.mark("start")
    feats = rma.to_nhwc(features_nchw)
    tm.mark("nhwc")
    out = {}
    if dense:
        volume, count = rma.backproject_accum(feats, projections, cfg.dims, cfg.voxel_size, cfg.origin, cfg.stride)
        out["volume"], out["count"] = volume, count
        tm.mark("dense")
    if proj_inv is None:
        proj_inv = rma.projection_inverse(projections, cfg.stride)
    proj_inv = proj_inv.to(feats.device, non_blocking=True)
    coords, pfeats, info = rma.aggregate_points(
        feats, proj_inv, tsdf, cfg.dims, cfg.voxel_size, cfg.origin, cfg.n_steps, cfg.thr, cfg.ray_marching_type,
        cfg.depth_points, offset=offset, max_points=cfg.max_points, sampler=cfg.sampler, mask=mask, seed=cfg.sample_seed)
    tm.mark("rma")
    x, _ = S.voxelize(coords, pfeats, cfg.voxel_size_fcaf3d)
    if P.current() is not None:
        P.current().mark("net")              # what follows scales with the number of scenes in a pass (Plan.scaled)
    tm.mark("voxelize")
    levels = backbone(x)
    tm.mark("backbone")
    cen, box, cls, pts = map(list, head(levels))
    tm.mark("head")
    bboxes, scores = head._get_bboxes_single([c[0] for c in cen], [b[0] for b in box], [c[0] for c in cls],
                                             [p[0] for p in pts])
    tm.mark("decode")
    out.update(bboxes=bboxes, scores=scores, M=info["M"], M_selected=info["M_selected"], M_unique=len(x),
               level_rows=[len(l) for l in levels], head_rows=[len(c[0]) for c in cen], stage_ms=tm.result())
    return out


def trace_net(plan, backbone, head, coords, feats, n_dev, voxel_size, device, extra_counts=(), late=None):
    """the sparse half inside a static trace (plan.static): voxelise -> MinkResNet34 -> neck / head -> decode of the point
    rows [0, n_dev) of the capacity-sized (coords, feats).  Returns the padded detections, the per-level head outputs and
    one small tensor of live counts -- no device->host read.
    late (feats is None): the aggregation's info with the point RECORDS instead of features -- the voxeliser carries the
    16-byte records (as 4-float rows) through its representative selection and sort, and the features of the surviving rows
    are emitted straight into the sparse tensor, with their magnitude bound: no [M, C] intermediate, no row gather, no
    absmax pass (0.27 ms per scene at the north-star shape)."""
    if feats is None:
        rec_f = late["rec"].view(torch.float32)                        # bit patterns travel through float4 copies untouched
        xr, _ = S.voxelize(coords, rec_f, voxel_size, n_dev=n_dev)
        amax = S._amax_slot(device)
        F = rma.emit_point_features(late, xr.F.view(torch.int32), xr.cs.n, xr.cs.n_dev, amax=amax)
        x = S.SparseTensor(F, xr.cs, None, amax)
    else:
        x, _ = S.voxelize(coords, feats, voxel_size, n_dev=n_dev)
    levels = backbone(x)
    cen, box, cls, pts, css = map(list, head(levels, fused=True))
    bboxes, scores, valid, sizes = head.get_bboxes_static(cen, box, cls, pts, css)
    status = plan.status(device)
    # every live count of the pass in ONE small tensor (read together with the detections, if at all)
    counts = torch.cat([c.view(1) for c in extra_counts] + [x.cs.n_dev.view(1)] + [l.cs.n_dev.view(1) for l in levels] +
                       [c[0].n_dev.view(1) for c in css]).to(torch.int32)
    return dict(bboxes=bboxes, scores=scores, valid=valid, sizes=sizes, status=status, counts=counts, n_levels=len(levels),
                n_extra=len(extra_counts), levels=levels,
                head=dict(centerness=[c[0] for c in cen], bbox_pred=[b[0] for b in box], cls_score=[c[0] for c in cls],
                          points=[p[0] for p in pts], n_dev=[c[0].n_dev for c in css]))


class StaticNet:
    """The sparse half alone (point rows -> raw detections) as a replayable HIP graph: what StaticScene runs after the
    aggregation, without the geometric half in front.  Used to pin the graph path of the network to the oracle on
    arbitrary point sets (tests) and by callers that bring their own points."""

    def __init__(self, backbone, head, voxel_size, device, margin=1.2, stream=None):
        self.backbone, self.head, self.voxel_size = backbone, head, voxel_size
        self.device = torch.device(device)
        self.margin = margin
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.plan = self.graph = self.out = None

    def build(self, coords, feats, capture=True, cap=None):
        """calibrate eagerly on (coords [M,3], feats [M,C]), then trace statically at point capacity `cap` (default M)
        and capture.  Returns the eager result dict(bboxes, scores)."""
        _lib.require_gpu()
        self._enter(coords, feats)
        with torch.cuda.stream(self.stream), torch.no_grad():
            plan = P.Plan(self.margin)
            with P.using(plan):
                x, _ = S.voxelize(coords, feats, self.voxel_size)
                levels = self.backbone(x)
                cen, box, cls, pts = map(list, self.head(levels))
                bboxes, scores = self.head._get_bboxes_single([c[0] for c in cen], [b[0] for b in box], [c[0] for c in cls],
                                                              [p[0] for p in pts])
            self.plan = plan
            M, C = feats.shape
            self.cap = int(cap or M)
            self.coords = torch.zeros((self.cap, 3), dtype=torch.float32, device=self.device)
            self.feats = torch.zeros((self.cap, C), dtype=torch.float32, device=self.device)
            self.n_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._load(coords, feats)
            self.out = self._trace()
            self.stream.synchronize()
            if capture:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.out = self._trace()
                self.stream.synchronize()
        return dict(bboxes=bboxes, scores=scores, level_rows=[len(l) for l in levels])

    def _load(self, coords, feats):
        n = coords.shape[0]
        assert n <= self.cap
        self.coords[:n].copy_(coords, non_blocking=True)
        self.feats[:n].copy_(feats, non_blocking=True)
        self.n_dev.fill_(n)

    def _trace(self):
        self.plan.begin_static()
        with P.using(self.plan), torch.no_grad():
            out = trace_net(self.plan, self.backbone, self.head, self.coords, self.feats, self.n_dev, self.voxel_size,
                            self.device)
        self.plan.end_static()
        return out

    def _enter(self, *inputs):
        """order self.stream behind the caller's stream (which produced the inputs) and keep their memory from being recycled
        while self.stream still reads it"""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
            for t in inputs:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.stream)

    def run(self, coords, feats):
        self._enter(coords, feats)
        with torch.cuda.stream(self.stream):
            self._load(coords, feats)
            if self.graph is not None:
                self.graph.replay()
            else:
                self.out = self._trace()
            self.done = torch.cuda.Event()
            self.done.record()
        self.out["done"] = self.done
        return self.out


class StaticScene:
    """One scene forward as a replayable HIP graph: the whole launch sequence of forward_scene() (dense unprojection,
    march, selection, voxelisation, MinkResNet34, neck/head, decode -- ~350 launches) captured once and replayed per
    scene with a single host call; no device->host read, no allocation, no Python in between.

    build() runs the scene eagerly once under a recording plan.Plan (every data-dependent row count is read back, as
    the reference does at nonzero() / in ME's coordinate manager), derives capacities (recorded size x `margin`) and
    re-runs the sequence in static mode -- once plainly (this creates the trace's device constants and is compared with
    the eager result by the tests), once under stream capture.  run() then costs: one layout kernel (the 2D backbone's
    NCHW maps -> the static channels-last buffer), three small input copies, one graph launch.  The outputs are static
    buffers, valid until the next run() of this object; `status` (device int32) counts violated capacity / branch
    assumptions -- non-zero means the scene outgrew the plan and must be re-run eagerly (forward_scene)."""

    def __init__(self, cfg, backbone, head, device, margin=1.2, dense=True, stream=None, by_reference=True):
        self.cfg, self.backbone, self.head = cfg, backbone, head
        self.device = torch.device(device)
        self.margin, self.dense = margin, dense
        # feature hand-off of channels-last maps (run()'s `by_reference` overrides it per scene): True = the trace reads the
        # caller's tensor in place for the WHOLE replay; False = copied into the slot's own buffer first
        self.by_reference = bool(by_reference)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.graph = None
        self.plan = None
        self.out = None
        self._copied = None
        self.check_weights = True          # compare the weights' fingerprint before every replay (~20 us of host time)
        self._weights = weight_tensors(backbone, head)
        self._tag = None

    # ---- inputs --------------------------------------------------------------------------------------------------
    def _alloc_inputs(self, features_nchw, tsdf):
        V, C, H, W = features_nchw.shape
        dev = self.device
        self.shape_nhwc = (V, H, W, C)
        # The kernels of the trace read the feature maps BY REFERENCE (feat_ref: a device word with their address, written per
        # scene): maps that are channels-last in memory -- what the 2D network hands over when it runs in
        # torch.channels_last -- are read where they lie; NCHW maps go through the layout pass into `nhwc`, a buffer that is
        # only allocated when the first such scene arrives (12.6 GB at the north-star shape).
        self.nhwc = None
        self.feat_ref = torch.zeros(1, dtype=torch.int64, device=dev)
        self._pin_ref = torch.zeros(1, dtype=torch.int64, pin_memory=True)
        self._held = None
        self.proj_scaled = torch.empty((V, 3, 4), dtype=torch.float32, device=dev)
        self.proj_inv = torch.empty((V, 4, 4), dtype=torch.float32, device=dev)
        self.tsdf = torch.empty(tuple(self.cfg.dims), dtype=torch.float32, device=dev)
        self.seed_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.offset_dev = torch.zeros(3, dtype=torch.float32, device=dev)       # switch_pointcloud's per-scene offset
        self._pin_off = torch.zeros(3, dtype=torch.float32, pin_memory=True)
        cfg = self.cfg
        # the march runs with the layout pass, in _load (one launch), into buffers the graph reads
        self.march = self.march_out = None
        if cfg.ray_marching_type == "neus":
            self.march = rma._March(None, self.proj_inv, self.tsdf, cfg.dims, cfg.voxel_size, cfg.origin, cfg.n_steps,
                                    cfg.thr, "neus", 0, shape=self.shape_nhwc, device=dev, feat_ref=self.feat_ref)
            if self.march.kept_cap() <= 0:
                raise _lib.CnrmaError("the static trace needs the single-march NeuS path (thr > 1/62)")
            self.march_out = self.march.march_buffers()
        else:
            self._nhwc_buffer()                  # depth mode: its emission kernel reads the static channels-last copy
        self._pin_proj = torch.empty((V, 3, 4), dtype=torch.float32, pin_memory=True)
        self._pin_inv = torch.empty((V, 4, 4), dtype=torch.float32, pin_memory=True)

    def _nhwc_buffer(self):
        """the slot's own channels-last feature buffer [V,H,W,C] (layout-pass target for NCHW inputs; producers may also write
        their output straight into it and call run(None, ...))"""
        if self.nhwc is None:
            self.nhwc = torch.empty(self.shape_nhwc, dtype=torch.float32, device=self.device)
            if self.march is not None:
                self.march.feat = self.nhwc
        return self.nhwc

    def _load(self, features_nchw, projections, tsdf, proj_inv=None, offset=None, by_reference=None):
        """stage one scene's inputs into the static buffers (on self.stream, which must be current).  by_reference: see
        run().  proj_inv: the
        [V,4,4] inverse of [P/stride; 0 0 0 1] when the caller pins it (parity tests: LAPACK's inverse is not bit-stable
        across host CPUs); computed here on the host like ray_marching.py:96-102 otherwise."""
        if self._copied is not None:
            self._copied.synchronize()           # the previous scene's copies out of the pinned buffers have executed
        p = projections.detach().to("cpu", torch.float32)
        self._pin_proj.copy_(rma.scale_projection(p, self.cfg.stride))
        self._pin_inv.copy_(rma.projection_inverse(p, self.cfg.stride) if proj_inv is None
                            else proj_inv.detach().to("cpu", torch.float32))
        self.proj_scaled.copy_(self._pin_proj, non_blocking=True)
        self.proj_inv.copy_(self._pin_inv, non_blocking=True)
        self.tsdf.copy_(tsdf.reshape(self.tsdf.shape), non_blocking=True)
        if offset is None:
            self._pin_off.zero_()
        else:
            self._pin_off.copy_(torch.as_tensor(offset, dtype=torch.float32).detach().reshape(3).cpu())
        self.offset_dev.copy_(self._pin_off, non_blocking=True)
        # where the trace reads the feature maps from: features_nchw None = the caller already wrote _nhwc_buffer(); maps that
        # are channels-last in memory are read in place (NeuS mode); everything else goes through the layout pass (NCHW ->
        # the slot's buffer, in ONE launch with the march)
        layout_from = None
        by_reference = self.by_reference if by_reference is None else bool(by_reference)
        self._held = None
        if features_nchw is None:
            addr = self._nhwc_buffer().data_ptr()
        elif rma.is_channels_last(features_nchw) and self.march is not None and by_reference:
            assert tuple(features_nchw.shape) == (self.shape_nhwc[0], self.shape_nhwc[3], self.shape_nhwc[1], self.shape_nhwc[2])
            addr = features_nchw.data_ptr()
            self._held = features_nchw           # alive until this slot takes its next scene
        elif rma.is_channels_last(features_nchw):
            self._nhwc_buffer().copy_(features_nchw.permute(0, 2, 3, 1), non_blocking=True)
            addr = self.nhwc.data_ptr()
        else:
            addr = self._nhwc_buffer().data_ptr()
            layout_from = features_nchw
        self._pin_ref[0] = addr
        self.feat_ref.copy_(self._pin_ref, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()                    # the pinned staging buffers are free again once this point has executed
        if self.march is not None:
            self.march.march(layout_from=layout_from, into=self.march_out)
        elif layout_from is not None:            # depth mode: the layout pass alone (its two small kernels run inside the trace)
            rma.to_nhwc(layout_from, out=self.nhwc)
        self._consumed = torch.cuda.Event()      # copies and the layout pass are behind this point: unless the maps are read
        self._consumed.record()                  # by reference (self._held), the caller's feature tensor is free again

    def _enter(self, *inputs):
        """order self.stream behind the caller's stream (the 2D backbone / Atlas head that produced the inputs ran
        there) and keep the inputs' memory from being recycled while self.stream still reads it"""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
            for t in inputs:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(self.stream)

    # ---- the launch sequence ---------------------------------------------------------------------------------------
    def _trace(self):
        cfg, plan = self.cfg, self.plan
        plan.begin_static()
        out = {}
        with P.using(plan), torch.no_grad():
            side = None
            if self.dense:
                # the dense volume is an output of the scene (the Atlas 3D network's input), nothing downstream in the hot path
                # reads it: DENSE_BRANCH runs it as a parallel branch of the graph (forked stream, joined at the end) -- off
                # by default, see the note at the switch
                main = torch.cuda.current_stream(self.device)
                if DENSE_BRANCH:
                    if getattr(self, "_side", None) is None:
                        self._side = torch.cuda.Stream(device=self.device)
                    side = self._side
                    side.wait_stream(main)
                with torch.cuda.stream(side if side is not None else main):
                    out["volume"], out["count"] = rma.backproject_accum(None, None, cfg.dims, cfg.voxel_size, cfg.origin,
                                                                        cfg.stride, proj_scaled=self.proj_scaled,
                                                                        feat_ref=self.feat_ref, shape=self.shape_nhwc)
            fixed = cfg.sample_seed is not None          # a fixed seed: every replay draws the same subset (as forward_scene)
            coords, feats, n_sel, info = rma.aggregate_points_static(
                self.nhwc, self.proj_inv, self.tsdf, cfg.dims, cfg.voxel_size, cfg.origin, cfg.n_steps, cfg.thr,
                max_points=cfg.max_points, seed=cfg.sample_seed if fixed else 0x5EED,
                seed_dev=None if fixed else self.seed_dev,
                marched=(self.march, self.march_out) if self.march is not None else None,
                mode=cfg.ray_marching_type, select_grids=cfg.depth_points or 0, feat_ref=self.feat_ref, shape=self.shape_nhwc,
                defer_feats=DEFER_POINT_FEATURES)
            moved = coords + self.offset_dev     # ray_marching.py:364 (one fp32 add per coordinate, as the reference)
            out.update(trace_net(plan, self.backbone, self.head, moved, feats, n_sel, cfg.voxel_size_fcaf3d, self.device,
                                 extra_counts=[info["M"], n_sel], late=info))
            if not fixed:
                self.seed_dev.add_(1)            # the next replay draws a fresh point subset
            if side is not None:
                torch.cuda.current_stream(self.device).wait_stream(side)          # join
        plan.end_static()
        # the aggregated points: places and row count are static outputs; with deferred features the [cap, C] rows are
        # produced on demand from the slot's records (point_features(): one launch outside the graph, on the caller's stream)
        out.update(points=(coords, feats, n_sel), points_info=info if feats is None else None)
        return out

    @staticmethod
    def point_features(out):
        """features [cap, C] of the aggregated points of a static output (rows >= out["points"][2] undefined): the stored rows,
        or -- deferred emission -- emitted now from the slot's records (valid until the slot's next run)"""
        coords, feats, n_sel = out["points"]
        if feats is None:
            if out.get("done") is not None:
                torch.cuda.current_stream(coords.device).wait_event(out["done"])
            feats = rma.point_features(out["points_info"])
        return feats

    def calibrate(self, features_nchw, projections, tsdf, proj_inv=None, offset=None):
        """one eager forward under a recording plan; several calls (several scenes of the configuration) are merged:
        capacities then cover the largest of them.  Returns the eager result."""
        _lib.require_gpu()
        self._enter(features_nchw, tsdf)
        with torch.cuda.stream(self.stream):
            plan = P.Plan(self.margin)
            with P.using(plan):
                eager = forward_scene(self.cfg, self.backbone, self.head, features_nchw, projections, tsdf, dense=self.dense,
                                      proj_inv=proj_inv, offset=_offset_list(offset))
        self.plan = plan if self.plan is None else self.plan.merge(plan)
        return eager

    def build(self, features_nchw, projections, tsdf, capture=True, plan=None, proj_inv=None):
        """calibrate on this scene (eager) unless calibrate() has been called or a finished `plan` of the same
        configuration is handed in (its sizes/flags are copied), then trace statically and capture.
        Returns the eager result of the calibration done here (None otherwise)."""
        _lib.require_gpu()
        eager = None
        if plan is not None:
            self.plan = P.Plan(self.margin)
            self.plan.sizes, self.plan.flags, self.plan.marks = list(plan.sizes), list(plan.flags), dict(plan.marks)
        elif self.plan is None:
            eager = self.calibrate(features_nchw, projections, tsdf, proj_inv)
        self._enter(features_nchw, tsdf)
        with torch.cuda.stream(self.stream):
            self._alloc_inputs(features_nchw, tsdf)
            self._load(features_nchw, projections, tsdf, proj_inv)
            self.out = self._trace()                         # plain static run: creates the trace's constants
            self.stream.synchronize()
            if capture:
                try:                                             # keep_graph only serves the node count, a diagnostic
                    self.graph, kept = torch.cuda.CUDAGraph(keep_graph=True), True
                except TypeError:                                # a torch build without it: capture instantiates by itself
                    self.graph, kept = torch.cuda.CUDAGraph(), False
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.out = self._trace()
                self.n_nodes = graph_node_count(self.graph) if kept else None   # launches per scene (bench: graph_nodes_per_scene)
                # tables of the CAPTURED trace that did not fit the 0xFF arena its sizing run measured (each clears itself: one
                # launch more; 0 unless the two runs took different table sequences -- ADVICE round 5)
                self.arena_fallbacks = self.plan.arena_fallbacks
                if kept:
                    self.graph.instantiate()
                self.stream.synchronize()
        self._tag = weights_tag(self._weights)
        return eager

    def run(self, features_nchw, projections, tsdf, proj_inv=None, offset=None, by_reference=None):
        """enqueue one scene on self.stream (ordered behind the caller's current stream, which produced the inputs);
        returns the static output dict (device tensors, valid until the next run; consumers on another stream wait on
        `self.done` first -- detections() does).

        Feature hand-off (`by_reference`, None = the slot's default set at construction, True unless told otherwise):
        * maps that are channels-last in memory and by_reference=True: the captured graph reads the CALLER'S tensor in
          place during the whole replay (~16 ms at the north-star shape, with up to `slots` scenes in flight).  CONTRACT: the
          caller must not write that memory -- no in-place op, no reuse of a preallocated / double-buffered / graph-static
          output buffer of the 2D network -- until `out["done"]` has completed.  The slot only keeps the memory from being
          freed or recycled (it holds the tensor and record_stream()s it); an in-place overwrite corrupts the dense volume
          and the point features silently (the plan status stays 0).
        * by_reference=False (or NCHW maps, always): the maps are copied / converted into the slot's own buffer (12.6 GB
          at the north-star shape, allocated on first use) before the trace starts; the caller may overwrite them as soon
          as `out["inputs_consumed"]` (an event on the slot's stream) has completed."""
        if self.graph is not None and self.check_weights and weights_tag(self._weights) != self._tag:
            raise _lib.CnrmaError("the model's weights changed since this scene graph was captured (optimiser step, "
                                  "load_state_dict, .to()): the graph replays prepared weight images -- rebuild() it")
        self._enter(features_nchw, tsdf)
        with torch.cuda.stream(self.stream):
            self._load(features_nchw, projections, tsdf, proj_inv, offset, by_reference)
            consumed = self._consumed
            if self.graph is not None:
                self.graph.replay()
            else:
                self.out = self._trace()
            self.done = torch.cuda.Event()
            self.done.record()
        self.out["done"] = self.done
        # when the caller may write the feature maps again: by reference -> only after the whole scene
        self.out["inputs_consumed"] = self.done if self._held is not None else consumed
        return self.out

    def rebuild(self, features_nchw, projections, tsdf):
        """fold the sizes of the scenes that outgrew the plan (detect()'s eager fall-backs) into the plan and trace /
        capture again at the larger capacities"""
        if getattr(self, "outgrown", None) is not None:
            self.plan.static = False
            self.plan = self.plan.merge(self.outgrown) if len(self.plan.sizes) == len(self.outgrown.sizes) else self.outgrown
            self.outgrown, self.n_outgrown = None, 0
        fresh = P.Plan(self.margin)
        fresh.sizes, fresh.flags, fresh.marks = list(self.plan.sizes), list(self.plan.flags), dict(self.plan.marks)
        self.plan = None
        return self.build(features_nchw, projections, tsdf, capture=True, plan=fresh)

    @staticmethod
    def detections(out):
        """static outputs -> (bboxes [K,*], scores [K,n_cls], info) like forward_scene; ONE device->host read.  Raises
        when the scene violated an assumption of the plan (the caller re-runs it with forward_scene)."""
        L = len(out["sizes"])
        if out.get("done") is not None:
            torch.cuda.current_stream(out["bboxes"].device).wait_event(out["done"])
        host = _lib.read_ints(torch.cat((out["status"].view(-1), out["valid"].view(-1), out["counts"].view(-1))))
        if host[0] != 0:
            raise _lib.CnrmaError(f"{host[0]} capacity / branch assumption(s) of the static plan violated: re-run eagerly")
        valid, counts = host[1:1 + L], host[1 + L:]
        rows, r0 = [], 0
        for k, v in zip(out["sizes"], valid):
            rows.append(torch.arange(r0, r0 + v, device=out["bboxes"].device))
            r0 += k
        rows = torch.cat(rows)
        nl, ne = out["n_levels"], out.get("n_extra", 2)
        info = dict(M_unique=counts[ne], level_rows=counts[ne + 1:ne + 1 + nl], head_rows=counts[ne + 1 + nl:ne + 1 + 2 * nl])
        if ne == 2:
            info.update(M=counts[0], M_selected=counts[1])
        return out["bboxes"].index_select(0, rows), out["scores"].index_select(0, rows), info

    def detect(self, features_nchw, projections, tsdf, rebuild_after=4, offset=None, by_reference=None):
        """run() + detections() with the fallback a server wants: a scene that outgrows the size plan (status != 0) is
        re-run through the eager path (sizes read back from the device) and its sizes are kept; after `rebuild_after`
        such scenes the plan is enlarged and the graph captured again (rebuild()), so a deployment whose scenes grew
        stops paying a replay plus an eager pass per scene.  Returns (bboxes, scores, info); info["static"] tells which
        path produced them."""
        out = self.run(features_nchw, projections, tsdf, offset=offset, by_reference=by_reference)
        try:
            with torch.cuda.stream(self.stream):          # the read-back waits on the stream the graph runs on
                b, s, info = self.detections(out)
            info["static"] = True
            return b, s, info
        except _lib.CnrmaError:
            with torch.cuda.stream(self.stream):
                grown = P.Plan(self.margin)
                with P.using(grown):
                    e = forward_scene(self.cfg, self.backbone, self.head, features_nchw, projections, tsdf, dense=self.dense,
                                      offset=_offset_list(offset))
            prev = getattr(self, "outgrown", None)
            same = prev is not None and len(prev.sizes) == len(grown.sizes) and len(prev.flags) == len(grown.flags)
            self.outgrown = prev.merge(grown) if same else grown
            self.n_outgrown = getattr(self, "n_outgrown", 0) + 1
            info = {k: e[k] for k in ("M", "M_selected", "M_unique", "level_rows", "head_rows")}
            info["static"] = False
            if rebuild_after and self.n_outgrown >= rebuild_after:
                self.rebuild(features_nchw, projections, tsdf)
                info["rebuilt"] = True
            return e["bboxes"], e["scores"], info


class StaticBatch:
    """Several scenes per static pass: the geometric half (layout, dense unprojection, march, selection) per scene, then
    ONE collated multi-scene sparse tensor (batch id = scene, what ME.utils.batch_sparse_collate builds for B samples,
    ray_marching.py:328-330) through the backbone / neck / head, per-scene instance norm / pruning / decode with
    device-side row counts -- captured as ONE HIP graph.  The ~300 latency-bound launches of the network (35 of its 51
    convolutions run on < 12 k rows per scene) are thereby shared by the scenes of a pass.

    The size plan is derived from a calibrated SINGLE-scene plan (Plan.scaled): per-scene capacities for the geometric
    half, n_scenes x the single-scene capacities for the network."""

    def __init__(self, cfg, backbone, head, device, n_scenes, margin=1.2, dense=True, stream=None):
        self.cfg, self.backbone, self.head = cfg, backbone, head
        self.device = torch.device(device)
        self.B, self.margin, self.dense = int(n_scenes), margin, dense
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        # input holders: one StaticScene per scene of the pass, used for its static input buffers and its loader only
        self.holders = [StaticScene(cfg, backbone, head, device, margin, dense, stream=self.stream) for _ in range(self.B)]
        self.graph = self.plan = self.out = None

    def build(self, scenes, plan, capture=True):
        """scenes: B tuples (features_nchw, projections, tsdf[, offset]); plan: a calibrated single-scene Plan (merged over
        the scenes of the configuration, StaticScene.calibrate)."""
        _lib.require_gpu()
        assert len(scenes) == self.B
        self.plan = plan.scaled(self.B)
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:                              # inputs written on the caller's stream just before build()
            self.stream.wait_stream(cur)
            for sc_ in scenes:
                for t_ in (sc_[0], sc_[2]):
                    if torch.is_tensor(t_) and t_.is_cuda:
                        t_.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            for h, sc_ in zip(self.holders, scenes):
                h._alloc_inputs(sc_[0], sc_[2])
            self._load(scenes)
            self.out = self._trace()
            self.stream.synchronize()
            if capture:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.out = self._trace()
                self.stream.synchronize()

    def _load(self, scenes):
        for h, sc_ in zip(self.holders, scenes):
            h._load(sc_[0], sc_[1], sc_[2], None, sc_[3] if len(sc_) > 3 else None)

    def _trace(self):
        cfg, plan, B = self.cfg, self.plan, self.B
        plan.begin_static()
        out = {}
        with P.using(plan), torch.no_grad():
            vols, pts, Ms = [], [], []
            fixed = cfg.sample_seed is not None
            for b, h in enumerate(self.holders):                  # the geometric half, scene by scene
                if self.dense:
                    vols.append(rma.backproject_accum(None, None, cfg.dims, cfg.voxel_size, cfg.origin, cfg.stride,
                                                      proj_scaled=h.proj_scaled, feat_ref=h.feat_ref, shape=h.shape_nhwc))
                coords, feats, n_sel, info = rma.aggregate_points_static(
                    h.nhwc, h.proj_inv, h.tsdf, cfg.dims, cfg.voxel_size, cfg.origin, cfg.n_steps, cfg.thr,
                    max_points=cfg.max_points, seed=(cfg.sample_seed if fixed else 0x5EED) + 7919 * b * (0 if fixed else 1),
                    seed_dev=None if fixed else h.seed_dev, marched=(h.march, h.march_out) if h.march is not None else None,
                    mode=cfg.ray_marching_type, select_grids=cfg.depth_points or 0, feat_ref=h.feat_ref, shape=h.shape_nhwc)
                pts.append((coords + h.offset_dev, feats, n_sel))
                Ms += [info["M"].view(1), n_sel.view(1)]
                if not fixed:
                    h.seed_dev.add_(1)
            x = S.sparse_collate_static(pts, cfg.voxel_size_fcaf3d)
            levels = self.backbone(x)
            cen, box, cls, pt, css = map(list, self.head(levels, fused=True))
            bboxes, scores, valid, sizes = self.head.get_bboxes_static_multi(cen, box, cls, pt, css, B)
            status = plan.status(self.device)
            per_level = [l.cs.counts_dev()[0] for l in levels] + [c[0].counts_dev()[0] for c in css]
            counts = torch.cat(Ms + [x.cs.counts_dev()[0]] + per_level).to(torch.int32)
        plan.end_static()
        if self.dense:
            out.update(volume=[v[0] for v in vols], count=[v[1] for v in vols])
        out.update(bboxes=bboxes, scores=scores, valid=valid, sizes=sizes, status=status, counts=counts, n_levels=len(levels))
        return out

    def run(self, scenes):
        """enqueue one pass of B scenes on self.stream; returns the static output dict (valid until the next run)"""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
            for sc_ in scenes:
                for t_ in (sc_[0], sc_[2]):
                    if torch.is_tensor(t_) and t_.is_cuda:
                        t_.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self._load(scenes)
            if self.graph is not None:
                self.graph.replay()
            else:
                self.out = self._trace()
            self.done = torch.cuda.Event()
            self.done.record()
        self.out["done"] = self.done
        return self.out

    def detections(self, out):
        """static outputs -> [(bboxes, scores, info)] per scene; ONE device->host read for the whole pass"""
        B, L = self.B, len(out["sizes"])
        if out.get("done") is not None:
            torch.cuda.current_stream(self.device).wait_event(out["done"])
        host = _lib.read_ints(torch.cat((out["status"].view(-1), out["valid"].view(-1), out["counts"].view(-1))))
        if host[0] != 0:
            raise _lib.CnrmaError(f"{host[0]} capacity / branch assumption(s) of the static plan violated: re-run eagerly")
        valid = host[1:1 + B * L]
        counts = host[1 + B * L:]
        nl = out["n_levels"]
        res = []
        for b in range(B):
            rows, r0 = [], 0
            for k, v in zip(out["sizes"], valid[b * L:(b + 1) * L]):
                rows.append(torch.arange(r0, r0 + v, device=self.device))
                r0 += k
            rows = torch.cat(rows)
            lv = counts[2 * B + B:]                               # after (M, M_sel) x B and the B unique counts
            info = dict(M=counts[2 * b], M_selected=counts[2 * b + 1], M_unique=counts[2 * B + b],
                        level_rows=[lv[l * B + b] for l in range(nl)], head_rows=[lv[(nl + l) * B + b] for l in range(nl)])
            res.append((out["bboxes"][b].index_select(0, rows), out["scores"][b].index_select(0, rows), info))
        return res


def gather_padded_detections(det, valid, det_all=None, valid_all=None):
    """ONE fixed-size exchange of a step's detections: det [S, K, W] = the padded raw boxes + scores of the
    S scenes a rank processed in the step, valid [S, L] = live rows per level.  Returns det_all [world, S, K, W] and
    valid_all [world, S, L].  No counts are read back: the padded blocks are small (<= 4000 x 25 floats per scene) and the
    row counts travel beside them.  Over RCCL (backend "nccl") this is two all_gather_into_tensor calls on device
    memory; other backends (gloo in the CPU / single-device tests) go through host lists."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return det.unsqueeze(0), valid.unsqueeze(0)
    W = dist.get_world_size()
    if det_all is None:
        det_all = torch.empty((W,) + tuple(det.shape), dtype=det.dtype, device=det.device)
        valid_all = torch.empty((W,) + tuple(valid.shape), dtype=valid.dtype, device=valid.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(det_all, det)
        dist.all_gather_into_tensor(valid_all, valid)
    else:
        d, v = det.cpu(), valid.cpu()
        ds, vs = [torch.empty_like(d) for _ in range(W)], [torch.empty_like(v) for _ in range(W)]
        dist.all_gather(ds, d)
        dist.all_gather(vs, v)
        det_all.copy_(torch.stack(ds))
        valid_all.copy_(torch.stack(vs))
    return det_all, valid_all


def gather_detections(bboxes, scores):
    """variable-length all-gather of one scene's detections per rank over RCCL: first the row counts,
    then one padded [K_max, box+cls] block per rank.  Returns a list (per rank) of (bboxes, scores)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [(bboxes, scores)]
    W = dist.get_world_size()
    k = torch.tensor([bboxes.shape[0]], dtype=torch.int64, device=bboxes.device)
    ks = [torch.zeros_like(k) for _ in range(W)]
    dist.all_gather(ks, k)
    kmax = int(max(int(x.item()) for x in ks))
    nb, nc = bboxes.shape[1], scores.shape[1]
    block = torch.zeros((kmax, nb + nc), dtype=torch.float32, device=bboxes.device)
    block[:bboxes.shape[0], :nb] = bboxes
    block[:bboxes.shape[0], nb:] = scores
    blocks = [torch.empty_like(block) for _ in range(W)]
    dist.all_gather(blocks, block)
    return [(b[:int(n.item()), :nb], b[:int(n.item()), nb:]) for b, n in zip(blocks, ks)]
