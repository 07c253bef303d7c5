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
from . import sparse as S


class SceneConfig:
    def __init__(self, dims, voxel_size=0.04, origin=(0.0, 0.0, 0.0), stride=4, n_steps=300, thr=0.05,
                 max_points=500000, voxel_size_fcaf3d=0.01, ray_marching_type="neus", depth_points=None,
                 sampler="device"):
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
        cfg.depth_points, offset=offset, max_points=cfg.max_points, sampler=cfg.sampler, mask=mask)
    tm.mark("rma")
    x, _ = S.voxelize(coords, pfeats, cfg.voxel_size_fcaf3d)
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
