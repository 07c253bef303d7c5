"""FCAF3D neck + head on the HIP sparse engine.

Registered name, constructor signature, forward outputs, pruning rule, box decoding and the raw-box dump follow
the reference's projects/mvsdetection/models/fcaf3d_head.py (FCAF3DHead :24-349, compute_centerness :395-402,
FCAF3DAssigner :405-484).  Parameter names match (up_block_i.0.kernel, out_block_i.1.bn.*, centerness_conv.kernel,
reg_conv.kernel, cls_conv.kernel/bias, scales.i.scale).  Training losses (:142-214) need mmdet's loss registry and
are a "next" row of SURVEY.md 8(f); the assigner is provided because it is pure tensor math."""
import os

import numpy as np
import torch
from torch import nn

from cnrma_amd import nn as snn
from cnrma_amd import plan as P
from cnrma_amd import sparse as S

from ..registry import BBOX_ASSIGNERS, HEADS, HAVE_MMDET, build_assigner


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar factor (key: `scale`)."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


class _TestCfg(dict):
    """test_cfg with attribute access (mmcv.Config style): missing keys read as None; dunder look-ups fall through, so that
    copy.deepcopy / pickle of the head work"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self.get(name)


def _without_type(cfg):
    return {k: v for k, v in dict(cfg).items() if k not in ("type", "use_sigmoid")}


class _SigmoidFocalLoss(nn.Module):
    """mmdet FocalLoss(use_sigmoid=True): sum over rows and classes of alpha_t (1 - p_t)^gamma BCE / avg_factor;
    label < 0 (or >= n_classes) = background row (all-zero target)."""

    def __init__(self, gamma=2.0, alpha=0.25, loss_weight=1.0, **_):
        super().__init__()
        self.gamma, self.alpha, self.loss_weight = gamma, alpha, loss_weight

    def forward(self, pred, labels, avg_factor=None):
        t = torch.zeros_like(pred)
        pos = (labels >= 0) & (labels < pred.shape[1])
        t[pos.nonzero().view(-1), labels[pos].long()] = 1.0
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        w = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = nn.functional.binary_cross_entropy_with_logits(pred, t, reduction="none") * w
        return self.loss_weight * loss.sum() / (avg_factor if avg_factor is not None else max(loss.numel(), 1))


class _SigmoidBCELoss(nn.Module):
    """mmdet CrossEntropyLoss(use_sigmoid=True): BCE with logits, summed / avg_factor"""

    def __init__(self, loss_weight=1.0, **_):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target, avg_factor=None):
        loss = nn.functional.binary_cross_entropy_with_logits(pred, target.float(), reduction="none")
        return self.loss_weight * loss.sum() / (avg_factor if avg_factor is not None else max(loss.numel(), 1))


class _IoU3DLoss(nn.Module):
    """IoU3DLoss of the FCAF3D code base: (1 - IoU3D) of (cx, cy, cz, dx, dy, dz[, yaw]) boxes, weighted, / avg_factor.
    with_yaw=False: axis-aligned boxes (ScanNet); with_yaw=True: rotated about z (ARKit) through core/rotated_iou.py."""

    def __init__(self, loss_weight=1.0, with_yaw=False, **_):
        super().__init__()
        self.loss_weight, self.with_yaw = loss_weight, with_yaw

    def forward(self, pred, target, weight=None, avg_factor=None):
        if self.with_yaw and pred.shape[1] >= 7:
            from ..core.rotated_iou import rotated_iou_3d
            iou = rotated_iou_3d(pred[:, :7], target[:, :7].to(pred.dtype))
        else:
            p_lo, p_hi = pred[:, :3] - pred[:, 3:6] / 2, pred[:, :3] + pred[:, 3:6] / 2
            t_lo, t_hi = target[:, :3] - target[:, 3:6] / 2, target[:, :3] + target[:, 3:6] / 2
            # products written out: prod()'s backward reads a scalar back (a host synchronisation inside loss.backward())
            ov = (torch.min(p_hi, t_hi) - torch.max(p_lo, t_lo)).clamp(min=0)
            inter = ov[:, 0] * ov[:, 1] * ov[:, 2]
            union = pred[:, 3] * pred[:, 4] * pred[:, 5] + target[:, 3] * target[:, 4] * target[:, 5] - inter
            iou = inter / union.clamp(min=1e-8)
        loss = 1 - iou
        if weight is not None:
            loss = loss * weight
        return self.loss_weight * loss.sum() / (avg_factor if avg_factor is not None else max(loss.numel(), 1))


from ..core.boxes import GTBoxes  # noqa: E402,F401  (kept importable from here: tests and the detector use it)


@HEADS.register_module()
class FCAF3DHead(nn.Module):
    def __init__(self, n_classes, in_channels, out_channels, n_reg_outs, voxel_size, pts_threshold, assigner,
                 yaw_parametrization="fcaf3d",
                 loss_centerness=dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type="IoU3DLoss", loss_weight=1.0),
                 loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 train_cfg=None, test_cfg=None):
        super().__init__()
        self.fp16_enabled = False
        self.voxel_size = voxel_size
        self.yaw_parametrization = yaw_parametrization
        self.assigner = build_assigner(assigner) if assigner is not None else None
        if HAVE_MMDET:
            from mmdet.models.builder import build_loss
            self.loss_centerness, self.loss_bbox, self.loss_cls = map(build_loss, (loss_centerness, loss_bbox, loss_cls))
        else:   # the three losses the configs name, restated in torch (mmdet / mmdet3d are third-party: unpinned)
            self.loss_centerness = _SigmoidBCELoss(**_without_type(loss_centerness))
            self.loss_bbox = _IoU3DLoss(**_without_type(loss_bbox))
            self.loss_cls = _SigmoidFocalLoss(**_without_type(loss_cls))
        self.train_cfg = train_cfg
        self.test_cfg = _TestCfg(test_cfg) if isinstance(test_cfg, dict) else test_cfg
        self.pts_threshold = pts_threshold
        self.n_classes, self.n_reg_outs = n_classes, n_reg_outs
        self._init_layers(in_channels, out_channels, n_reg_outs, n_classes)
        self._fused_head = None

    # ---- layers (reference :61-98) -------------------------------------------------------------------------
    @staticmethod
    def _make_block(cin, cout):
        return snn.FusedSequential(snn.MinkowskiConvolution(cin, cout, kernel_size=3, dimension=3),
                                   snn.MinkowskiBatchNorm(cout), snn.MinkowskiELU())

    @staticmethod
    def _make_up_block(cin, cout):
        return snn.FusedSequential(
            snn.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=2, stride=2, dimension=3),
            snn.MinkowskiBatchNorm(cout), snn.MinkowskiELU(),
            snn.MinkowskiConvolution(cout, cout, kernel_size=3, dimension=3),
            snn.MinkowskiBatchNorm(cout), snn.MinkowskiELU())

    def _init_layers(self, in_channels, out_channels, n_reg_outs, n_classes):
        self.pruning = snn.MinkowskiPruning()
        for i in range(len(in_channels)):
            if i > 0:
                setattr(self, f"up_block_{i}", self._make_up_block(in_channels[i], in_channels[i - 1]))
            setattr(self, f"out_block_{i}", self._make_block(in_channels[i], out_channels))
        self.centerness_conv = snn.MinkowskiConvolution(out_channels, 1, kernel_size=1, dimension=3)
        self.reg_conv = snn.MinkowskiConvolution(out_channels, n_reg_outs, kernel_size=1, dimension=3)
        self.cls_conv = snn.MinkowskiConvolution(out_channels, n_classes, kernel_size=1, bias=True, dimension=3)
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(len(in_channels))])

    def init_weights(self):
        nn.init.normal_(self.centerness_conv.kernel, std=.01)
        nn.init.normal_(self.reg_conv.kernel, std=.01)
        nn.init.normal_(self.cls_conv.kernel, std=.01)
        nn.init.constant_(self.cls_conv.bias, float(-np.log((1 - 0.01) / 0.01)))   # bias_init_with_prob(.01)
        self._fused_head = None

    def train(self, mode=True):
        self._fused_head = None
        return super().train(mode)

    # ---- forward (reference :107-139, :275-298) ---------------------------------------------------------------
    def forward(self, x, fused=False):
        """fused=True (eval only): several scenes stay in ONE row set per level -- every output list has a single entry
        and a fifth list holds the level's coordinate set (scene ids, cached rows per scene); decode with
        get_bboxes_fused()."""
        outs = []
        inputs = x
        x = inputs[-1]
        scores = None
        for i in range(len(inputs) - 1, -1, -1):
            if i < len(inputs) - 1:
                x = getattr(self, f"up_block_{i + 1}")(x)
                x = inputs[i] + x                       # coordinate union, features added (:114)
                x = self._prune(x, scores)
            out = getattr(self, f"out_block_{i}")(x)
            out = self.forward_single(out, self.scales[i], fused)
            scores = out[-1]
            outs.append(out[:-1] + ([out[-1].cs],) if fused else out[:-1])
        return zip(*outs[::-1])

    def _prune(self, x, scores):
        """keep the top `pts_threshold` rows per scene by the interpolated max-class score of the coarser level"""
        if self.pts_threshold < 0:
            return x
        plan = P.current()
        if plan is not None and plan.static:
            # static trace (one scene): the branch of the calibration run, its premise registered as an assumption
            B = x.cs.n_batch
            if plan.next_flag():                     # True: no calibration scene needed pruning; None (scenes differ) -> prune
                if B <= 1:
                    plan.watch(x.cs.n_dev, 0, self.pts_threshold)
                else:                                # several scenes per pass: the premise holds per scene
                    cnt, _ = x.cs.counts_dev()
                    for b in range(B):
                        plan.watch(cnt[b:b + 1], 0, self.pts_threshold)
                return x
            with torch.no_grad():
                interpolated = S.interpolate(scores, x.C, x.cs.n_dev)
                if B <= 1:
                    mask = S.topk_mask(interpolated, self.pts_threshold, n_dev=x.cs.n_dev)   # keeps every row when n <= threshold
                else:                                # per scene: the other scenes' rows ranked -inf and never kept
                    flat, scene = interpolated.view(-1), x.C[:, 0]
                    low = torch.full_like(flat, float("-inf"))
                    mask = None
                    for b in range(B):
                        mine = scene == b
                        m = S.topk_mask(torch.where(mine, flat, low), self.pts_threshold, n_dev=x.cs.n_dev) & mine.to(torch.uint8)
                        mask = m if mask is None else mask | m
            return S.prune(x, mask, n_keep=min(B * self.pts_threshold, x.cs.n))
        with torch.no_grad():
            counts = x.cs.batch_counts()
            skip = all(c <= self.pts_threshold for c in counts)
            if plan is not None:
                plan.record_flag(skip)
            if skip:
                return x                                # the top-k keeps every row: pruning is the identity
            interpolated = S.interpolate(scores, x.C)
            # radix-select keep-mask instead of torch.topk's sort (same row set; ties by index)
            kept = [min(c, self.pts_threshold) for c in counts]
            if len(counts) == 1:
                mask = S.topk_mask(interpolated, self.pts_threshold)
            else:                                       # per scene: the other scenes' rows are masked to -inf
                flat, scene = interpolated.view(-1), x.C[:, 0]
                low = torch.full_like(flat, float("-inf"))
                mask = None
                for b, nb in enumerate(counts):
                    mine = scene == b
                    m = mine.to(torch.uint8) if nb <= self.pts_threshold else \
                        S.topk_mask(torch.where(mine, flat, low), self.pts_threshold)
                    mask = m if mask is None else mask | m
        return S.prune(x, mask, n_keep=sum(kept), counts=kept)      # the mask holds exactly min(n, threshold) rows per scene

    def _head_weights(self):
        """the three 1x1 head convolutions as ONE [128, 1+R+n_cls] GEMM (+ bias row for the class logits)"""
        params = (self.centerness_conv.kernel, self.reg_conv.kernel, self.cls_conv.kernel, self.cls_conv.bias)
        tag = tuple((t.data_ptr(), t._version, t.device) for t in params)       # in-place updates / load_state_dict bump _version
        if self._fused_head is None or self._fused_head[2] != tag:
            with torch.no_grad():
                w = torch.cat((self.centerness_conv.kernel, self.reg_conv.kernel, self.cls_conv.kernel), dim=1)
                n_out = w.shape[1]
                pad = (-n_out) % 4                       # 16-byte rows: the conv kernel's branch-free staging path
                w = torch.cat((w, w.new_zeros(w.shape[0], pad)), dim=1).contiguous()
                b = torch.zeros(w.shape[1], device=w.device)
                b[1 + self.n_reg_outs:n_out] = self.cls_conv.bias.view(-1)
            self._fused_head = (w, b.contiguous(), tag)
        return self._fused_head[:2]

    def forward_single(self, x, scale, fused=False):
        if self.training or (x.cs.n_batch > 1 and not fused):
            if self.training:
                centerness = self.centerness_conv(x).F
                cls_score = self.cls_conv(x).F
                reg_final = self.reg_conv(x).F
            else:
                w, b = self._head_weights()
                y = S.conv(x, w, kernel_size=1, shift=b).F
                n_out = 1 + self.n_reg_outs + self.n_classes
                centerness, reg_final, cls_score = y[:, :1], y[:, 1:1 + self.n_reg_outs], y[:, 1 + self.n_reg_outs:n_out]
            prune_scores = S.SparseTensor(S.row_max(cls_score.contiguous()), x.cs)           # :279-282
            reg_distance = torch.exp(scale(reg_final[:, :6]))                                  # :284
            bbox_pred = torch.cat((reg_distance, reg_final[:, 6:]), dim=1)                     # :285-286
            perms = x.decomposition_permutations
            # index_select: its backward is an index_add (advanced indexing's sorts, far slower on large row sets)
            centernesses = [centerness.index_select(0, p) for p in perms]
            bbox_preds = [bbox_pred.index_select(0, p) for p in perms]
            cls_scores = [cls_score.index_select(0, p) for p in perms]
            points = [c * self.voxel_size for c in x.decomposed_coordinates]                   # :294-296
            return centernesses, bbox_preds, cls_scores, points, prune_scores
        # single scene, eval: the three 1x1 convolutions are ONE GEMM and the whole tail is one kernel
        w, b = self._head_weights()
        y = S.conv(x, w, kernel_size=1, shift=b).F
        cen, box, cls, mx, pts = S.head_post(y, x.C, self.n_reg_outs, self.n_classes, scale.scale, self.voxel_size)
        return [cen], [box], [cls], [pts], S.SparseTensor(mx, x.cs)

    # ---- decoding (reference :217-271, :300-349) -----------------------------------------------------------------
    def _bbox_pred_to_bbox(self, points, bbox_pred):
        if bbox_pred.shape[0] == 0:
            return bbox_pred
        if torch.is_grad_enabled() and bbox_pred.requires_grad:       # training: the same formulas in torch (:300-349)
            b, p = bbox_pred, points
            ctr = torch.stack((p[:, 0] + (b[:, 1] - b[:, 0]) / 2, p[:, 1] + (b[:, 3] - b[:, 2]) / 2,
                               p[:, 2] + (b[:, 5] - b[:, 4]) / 2), dim=-1)
            if b.shape[1] == 6:
                return torch.cat((ctr, torch.stack((b[:, 0] + b[:, 1], b[:, 2] + b[:, 3], b[:, 4] + b[:, 5]), -1)), dim=-1)
            if self.yaw_parametrization == "naive":
                return torch.cat((ctr, torch.stack((b[:, 0] + b[:, 1], b[:, 2] + b[:, 3], b[:, 4] + b[:, 5]), -1), b[:, 6:7]), -1)
            if self.yaw_parametrization == "sin-cos":
                norm = torch.sqrt(b[:, 6:7] ** 2 + b[:, 7:8] ** 2)
                size = torch.stack((b[:, 0] + b[:, 1], b[:, 2] + b[:, 3], b[:, 4] + b[:, 5]), -1)
                return torch.cat((ctr, size, torch.atan2(b[:, 6:7] / norm, b[:, 7:8] / norm)), -1)
            scale = b[:, 0] + b[:, 1] + b[:, 2] + b[:, 3]
            q = torch.exp(torch.sqrt(b[:, 6] ** 2 + b[:, 7] ** 2))
            return torch.cat((ctr, torch.stack((scale / (1 + q), scale / (1 + q) * q, b[:, 5] + b[:, 4],
                                                0.5 * torch.atan2(b[:, 6], b[:, 7])), -1)), dim=-1)
        return S.decode_boxes(points.float(), bbox_pred, self.yaw_parametrization)

    def _get_bboxes_single(self, centernesses, bbox_preds, cls_scores, points, scene_id=None, save_path=None):
        mlvl_bboxes, mlvl_scores = [], []
        nms_pre = self.test_cfg.nms_pre if self.test_cfg is not None else 0
        plan = P.current()
        for centerness, bbox_pred, cls_score, point in zip(centernesses, bbox_preds, cls_scores, points):
            ids = None
            if plan is not None:
                plan.record_flag(len(cls_score) > nms_pre > 0)
            if len(cls_score) > nms_pre > 0:
                max_scores = S.max_scores(cls_score, centerness)                            # :249-250 (ranking key only)
                ids = S.topk_indices(max_scores, nms_pre)                                   # :252-256
            boxes, scores = S.select_decode(ids, cls_score, centerness, bbox_pred, point, self.yaw_parametrization)
            mlvl_bboxes.append(boxes)
            mlvl_scores.append(scores)
        bboxes, scores = torch.cat(mlvl_bboxes), torch.cat(mlvl_scores)
        if save_path is not None:
            save_place = os.path.join(save_path, scene_id)
            os.makedirs(save_place, exist_ok=True)
            np.savez(os.path.join(save_place, scene_id + "_bbox_raw.npz"), bboxes=bboxes.detach().cpu().numpy(),
                     scores=scores.detach().cpu().numpy())                                  # :266-271
        return bboxes, scores

    def get_bboxes(self, centernesses, bbox_preds, cls_scores, points, scene_ids, save_path):
        out = []
        for i in range(len(centernesses[0])):
            out.append(self._get_bboxes_single([x[i] for x in centernesses], [x[i] for x in bbox_preds],
                                               [x[i] for x in cls_scores], [x[i] for x in points],
                                               scene_ids[i] if scene_ids is not None else None, save_path))
        return out

    def get_bboxes_static(self, centernesses, bbox_preds, cls_scores, points, coord_sets):
        """_get_bboxes_single inside the static trace (plan.Plan; one scene, forward(..., fused=True) outputs): no row
        count is read back.  Per level the branch of the calibration run is replayed -- more rows than nms_pre: the
        nms_pre best by max class score x centerness (dead rows ranked -inf), else all rows -- and its premise registered.
        Returns (bboxes [K,6|7], scores [K,n_cls], valid int32 [L], sizes): level l owns rows [sum(sizes[:l]), +sizes[l]) of
        which the first valid[l] are detections (the reference's row order within the level)."""
        plan = P.current()
        nms_pre = self.test_cfg.nms_pre if self.test_cfg is not None else 0
        boxes, scores, valid, sizes = [], [], [], []
        for cen, box, cls, pts, cs in zip(centernesses, bbox_preds, cls_scores, points, coord_sets):
            cen, box, cls, pts, cs = cen[0], box[0], cls[0], pts[0], cs[0]
            cap, n_dev = cs.n, cs.n_dev
            flag = plan.next_flag()
            if flag or (flag is None and cap > nms_pre > 0):
                # None: the calibration scenes disagreed -- the top-k form is valid on both sides (all live rows when
                # there are fewer than nms_pre, then in score order instead of row order)
                assert cap > nms_pre
                plan.watch(n_dev, nms_pre + 1 if flag else 0, cap)
                ids = S.topk_indices(S.max_scores(cls, cen), nms_pre, n_dev)
                k = nms_pre
                valid.append(torch.clamp(n_dev.view(1), max=nms_pre).to(torch.int32))
            else:
                plan.watch(n_dev, 0, min(cap, nms_pre) if nms_pre > 0 else cap)
                ids, k = None, cap
                valid.append(n_dev.view(1))
            bx, sc = S.select_decode(ids, cls, cen, box, pts, self.yaw_parametrization)
            boxes.append(bx)
            scores.append(sc)
            sizes.append(k)
        return torch.cat(boxes), torch.cat(scores), torch.cat(valid), sizes

    def get_bboxes_static_multi(self, centernesses, bbox_preds, cls_scores, points, coord_sets, n_scenes):
        """get_bboxes_static for several scenes in one row set per level (static trace of pipeline.StaticBatch).  Per level
        and scene: the rows of the other scenes are ranked -inf, the nms_pre best are taken with the radix select; a scene
        with no more than nms_pre rows on the level keeps them all, in row order like the reference (:247-256 only cuts
        when there are more).  Returns (bboxes [B,K,6|7], scores [B,K,n_cls], valid int32 [B,L], sizes): scene b / level l
        owns rows [sum(sizes[:l]), +sizes[l]) of block b, of which the first valid[b,l] are detections."""
        plan = P.current()
        nms_pre = self.test_cfg.nms_pre if self.test_cfg is not None else 0
        B = n_scenes
        boxes, scores, valid, sizes = [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)], []
        for cen, box, cls, pts, cs in zip(centernesses, bbox_preds, cls_scores, points, coord_sets):
            cen, box, cls, pts, cs = cen[0], box[0], cls[0], pts[0], cs[0]
            plan.next_flag()                                     # the single-scene branch of the calibration: not needed here
            cap, n_dev = cs.n, cs.n_dev
            k = min(nms_pre, cap) if nms_pre > 0 else cap
            cnt, _ = cs.counts_dev()
            ms = S.max_scores(cls, cen)
            live = torch.arange(cap, device=ms.device, dtype=torch.int32) < n_dev
            # ONE stable sort per level on the key (scene, descending score) -- scores are >= 0, so their bit patterns order
            # them; dead rows sort behind every scene --: each scene's rows become a contiguous run in rank order, found
            # through the device-side offsets; B radix selects per level cost 4x as much
            scene = torch.where(live, cs.C[:, 0], torch.full_like(cs.C[:, 0], B)).long()
            order = torch.sort((scene << 32) | (0xFFFFFFFF - ms.view(torch.int32).long()), stable=True)[1]
            off = torch.cumsum(cnt, 0) - cnt
            slot = plan.const(lambda: torch.arange(k, device=ms.device, dtype=torch.int64))
            big = plan.const(lambda: torch.full((k,), cap, device=ms.device, dtype=torch.int64))
            for b in range(B):
                ids = order.index_select(0, (off[b:b + 1].long() + slot).clamp_(max=cap - 1))   # scene b's best, in score order
                nb = torch.clamp(cnt[b:b + 1], max=k)
                asc = torch.sort(torch.where(slot < nb, ids, big))[0].clamp_(max=cap - 1)         # the same rows in row order
                ids = torch.where(cnt[b:b + 1] <= k, asc, ids)
                bx, sc = S.select_decode(ids, cls, cen, box, pts, self.yaw_parametrization)
                boxes[b].append(bx)
                scores[b].append(sc)
                valid[b].append(nb.to(torch.int32))
            sizes.append(k)
        return (torch.stack([torch.cat(x) for x in boxes]), torch.stack([torch.cat(x) for x in scores]),
                torch.stack([torch.cat(v) for v in valid]), sizes)

    def get_bboxes_fused(self, centernesses, bbox_preds, cls_scores, points, scenes, n_scenes):
        """decode of forward(..., fused=True): per level ONE row set for all scenes + the rows' scene ids.  Per scene the
        nms_pre best rows by max class score x centerness are picked with the other scenes masked out (reference
        _get_bboxes_single :247-256 per scene).  Returns [(bboxes, scores)] per scene."""
        nms_pre = self.test_cfg.nms_pre if self.test_cfg is not None else 0
        per_scene = [([], []) for _ in range(n_scenes)]
        for cen, box, cls, pts, cs in zip(centernesses, bbox_preds, cls_scores, points, scenes):
            cen, box, cls, pts, cs = cen[0], box[0], cls[0], pts[0], cs[0]
            sc = cs.C[:, 0]
            if n_scenes == 1:
                ids = None
                if len(cls) > nms_pre > 0:
                    ids = S.topk_indices(S.max_scores(cls, cen), nms_pre)
                groups = [ids]
            else:
                # ONE stable sort per level on the key (scene, descending score): every scene's rows become a contiguous
                # run in rank order, of which the first nms_pre are taken (scores are >= 0: their bit patterns order them)
                counts = cs.batch_counts()                  # usually cached by the pruning step of the same level
                bits = S.max_scores(cls, cen).view(torch.int32).long()
                order = torch.sort((sc.long() << 32) | (0xFFFFFFFF - bits), stable=True)[1]
                groups, r0 = [], 0
                for nb in counts:
                    groups.append(order[r0:r0 + (min(nb, nms_pre) if nms_pre > 0 else nb)])
                    r0 += nb
            for b, ids in enumerate(groups):
                bx, scr = S.select_decode(ids, cls, cen, box, pts, self.yaw_parametrization)
                per_scene[b][0].append(bx)
                per_scene[b][1].append(scr)
        return [(torch.cat(bx), torch.cat(scr)) for bx, scr in per_scene]

    def loss(self, centernesses, bbox_preds, cls_scores, points, gt_bboxes, gt_labels):
        """per-scene assignment + centerness / IoU / focal losses, averaged over the scenes (reference :142-167)"""
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) == len(points[0]) == len(gt_bboxes) \
            == len(gt_labels)
        per = [self._loss_single([x[i] for x in centernesses], [x[i] for x in bbox_preds], [x[i] for x in cls_scores],
                                 [x[i] for x in points], gt_bboxes[i], gt_labels[i]) for i in range(len(gt_bboxes))]
        return dict(loss_centerness=torch.mean(torch.stack([p[0] for p in per])),
                    loss_bbox=torch.mean(torch.stack([p[1] for p in per])),
                    loss_cls=torch.mean(torch.stack([p[2] for p in per])))

    def _loss_single(self, centernesses, bbox_preds, cls_scores, points, gt_bboxes, gt_labels):
        """reference :170-214 (single process: reduce_mean is the identity; under DDP mmdet's reduce_mean averages the
        normalisers over the ranks)"""
        if self.assigner is None:
            raise ValueError("FCAF3DHead.loss needs an assigner (config key `assigner`)")
        if not hasattr(gt_bboxes, "gravity_center"):
            gt_bboxes = GTBoxes(torch.as_tensor(gt_bboxes))
        with torch.no_grad():
            centerness_targets, bbox_targets, labels = self.assigner.assign(points, gt_bboxes, gt_labels)
        centerness, bbox_pred, cls_score, pts = map(torch.cat, (centernesses, bbox_preds, cls_scores, points))
        pos = torch.nonzero(labels >= 0).squeeze(1)
        n_pos = max(_reduce_mean(torch.tensor(float(len(pos)), device=centerness.device)), 1.0)
        loss_cls = self.loss_cls(cls_score, labels, avg_factor=n_pos)
        pos_ctr, pos_box = centerness[pos], bbox_pred[pos]
        pos_ctr_t = centerness_targets[pos].unsqueeze(1)
        denorm = max(_reduce_mean(pos_ctr_t.sum().detach()), 1e-6)
        if len(pos) > 0:
            loss_centerness = self.loss_centerness(pos_ctr, pos_ctr_t, avg_factor=n_pos)
            loss_bbox = self.loss_bbox(self._bbox_pred_to_bbox(pts[pos], pos_box), bbox_targets[pos],
                                       weight=pos_ctr_t.squeeze(1), avg_factor=denorm)
        else:
            loss_centerness, loss_bbox = pos_ctr.sum(), pos_box.sum()
        return loss_centerness, loss_bbox, loss_cls


def _reduce_mean(t):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = t.clone()
        dist.all_reduce(t.div_(dist.get_world_size()))
    return float(t)


def compute_centerness(bbox_targets):
    """sqrt of the product over axes of min/max face distances (reference :395-402)."""
    d = bbox_targets[..., :6]
    x, y, z = d[..., 0:2], d[..., 2:4], d[..., 4:6]
    c = x.min(dim=-1)[0] / x.max(dim=-1)[0] * y.min(dim=-1)[0] / y.max(dim=-1)[0] * z.min(dim=-1)[0] / z.max(dim=-1)[0]
    return torch.sqrt(c)


@BBOX_ASSIGNERS.register_module()
class FCAF3DAssigner(object):
    """Point-to-box assignment of FCAF3D (reference :405-484): inside-box, best-scale (first level with fewer than
    `limit` inside points, minus one), top-k centerness, smallest volume."""

    def __init__(self, limit, topk, n_scales):
        self.limit, self.topk, self.n_scales = limit, topk, n_scales

    def assign(self, points, gt_bboxes, gt_labels):
        """points: list (per level) of [n_i,3]; gt_bboxes exposes .gravity_center [m,3], .tensor [m,7], .volume [m].
        Returns (centerness_targets [n], bbox_targets [n,7], labels [n], -1 = background)."""
        big = 1e8
        level = torch.cat([p.new_tensor(i).expand(len(p)) for i, p in enumerate(points)])
        pts = torch.cat(points, dim=0)
        n, m = len(pts), len(gt_bboxes)
        vol = gt_bboxes.volume.to(pts.device).expand(n, m).contiguous()
        box = torch.cat((gt_bboxes.gravity_center, gt_bboxes.tensor[:, 3:]), dim=1).to(pts.device)     # [m,7]
        boxes = box.expand(n, m, 7)
        rel = pts[:, None, :] - box[None, :, :3]                                                       # [n,m,3]
        # mmdet3d 0.15 rotation_3d_in_axis(shift, -yaw, axis=2): row-vector times [[c,-s,0],[s,c,0],[0,0,1]]
        # (third-party convention, not pinned by anything under /root/reference)
        ang = -box[:, 6]
        c, s = torch.cos(ang)[None], torch.sin(ang)[None]
        rot = torch.stack((rel[..., 0] * c + rel[..., 1] * s, -rel[..., 0] * s + rel[..., 1] * c, rel[..., 2]), dim=-1)
        centers = boxes[..., :3] + rot
        half = boxes[..., 3:6] / 2
        lo = centers - boxes[..., :3] + half                       # distances to the three "min" faces
        hi = boxes[..., :3] + half - centers                        # ... and the three "max" faces
        targets = torch.stack((lo[..., 0], hi[..., 0], lo[..., 1], hi[..., 1], lo[..., 2], hi[..., 2], boxes[..., 6]), -1)
        inside = targets[..., :6].min(-1)[0] > 0                                                     # condition 1
        per_level = torch.stack([inside[level == i].sum(dim=0) for i in range(self.n_scales)], dim=0)  # [L,m]
        too_few = per_level < self.limit
        lower = torch.argmax(too_few.int(), dim=0) - 1
        lower = torch.where(lower < 0, torch.zeros_like(lower), lower)
        best = torch.where(torch.all(~too_few, dim=0), torch.full_like(lower, self.n_scales - 1), lower)
        at_best = best[None, :].expand(n, m) == level[:, None].expand(n, m)                           # condition 2
        cness = compute_centerness(targets)
        cness = torch.where(inside & at_best, cness, -torch.ones_like(cness))
        kth = torch.topk(cness, min(self.topk + 1, len(cness)), dim=0).values[-1]
        top = cness > kth[None]                                                                       # condition 3
        vol = torch.where(inside & at_best & top, vol, torch.full_like(vol, big))
        min_vol, arg = vol.min(dim=1)
        labels = torch.where(min_vol == big, torch.full_like(gt_labels[arg], -1), gt_labels[arg])
        rows = torch.arange(n, device=pts.device)
        picked = targets[rows, arg]
        return compute_centerness(picked), boxes[rows, arg], labels
