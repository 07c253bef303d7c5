"""Offline post-processing on the device: per-class 3D NMS over the raw boxes the head dumps,
and the indoor mAP evaluation.  Host mirror of the reference's post_process/nms_bbox.py:17-66 and
post_process/evaluate_bbox.py:18-100 (which call mmdet3d's pcdet_nms_* and indoor_eval -- third-party, semantics
restated in include/cnrma.h / oracle/post_oracle.py)."""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream


def box_iou(a, b, rotated=True, mode3d=True):
    """pairwise IoU [na, nb] of (x,y,z,dx,dy,dz[,heading]) boxes; BEV or 3D (BEV overlap x height overlap)."""
    _lib.require_gpu()
    a, b = _as7(a), _as7(b)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if a.shape[0] and b.shape[0]:
        call("cnrma_box_iou_f32", ptr(a), a.shape[0], ptr(b), b.shape[0], int(rotated), int(mode3d), ptr(out), stream())
    return out


def _as7(boxes):
    boxes = boxes.float()
    if boxes.shape[1] == 6:
        boxes = torch.cat((boxes, torch.zeros_like(boxes[:, :1])), dim=1)
    return boxes.contiguous()


def nms_single_class(boxes, scores, iou_thr, rotated):
    """indices (into `boxes`) of the kept boxes, in descending score order (pcdet_nms_gpu / pcdet_nms_normal_gpu)."""
    _lib.require_gpu()
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = _as7(boxes)[order].contiguous()
    words = (n + 63) // 64
    mask = torch.empty((n, words), dtype=torch.int64, device=boxes.device)
    call("cnrma_nms_mask_f32", ptr(b), n, float(iou_thr), int(rotated), ptr(mask), stream())
    m = mask.cpu().numpy().view(np.uint64)
    removed = np.zeros(words, dtype=np.uint64)
    keep = []
    for i in range(n):                                   # greedy scan in score order (sequential by nature)
        if not (removed[i >> 6] >> np.uint64(i & 63)) & np.uint64(1):
            keep.append(i)
            removed |= m[i]
    return order[torch.as_tensor(keep, dtype=torch.long, device=boxes.device)]


def nms(bboxes, scores, score_thr=0.01, iou_thr=0.5):
    """nms() of post_process/nms_bbox.py:17-58: per class keep scores > score_thr, BEV NMS (rotated when the boxes carry
    a yaw).  Returns (boxes [K, 6|7] gravity-centre, scores [K], labels [K])."""
    n_classes = scores.shape[1]
    yaw = bboxes.shape[1] == 7
    out_b, out_s, out_l = [], [], []
    for c in range(n_classes):
        ids = torch.nonzero(scores[:, c] > score_thr).squeeze(1)
        if ids.numel() == 0:
            continue
        keep = nms_single_class(bboxes[ids], scores[ids, c], iou_thr, rotated=yaw)
        out_b.append(bboxes[ids][keep])
        out_s.append(scores[ids, c][keep])
        out_l.append(torch.full((keep.numel(),), c, dtype=torch.long, device=bboxes.device))
    if not out_b:
        return bboxes.new_zeros((0, bboxes.shape[1])), bboxes.new_zeros((0,)), bboxes.new_zeros((0,), dtype=torch.long)
    return torch.cat(out_b), torch.cat(out_s), torch.cat(out_l)


def to_saved_layout(boxes):
    """Boxes as nms_bbox.py:60-66 saves them.  The reference builds DepthInstance3DBoxes(origin=(.5,.5,.5)) (bottom
    centre inside the object) and then adds h/2 back to z: the saved z is the gravity centre again."""
    return boxes.clone()


def average_precision(recall, precision):
    """area under the monotone precision envelope (mmdet3d indoor_eval.average_precision, 'area' mode)"""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def indoor_eval(gt_annos, dt_annos, iou_thrs=(0.25, 0.5), n_classes=None, device="cuda"):
    """mAP / mAR as mmdet3d's indoor_eval computes them.
    gt_annos: list of dict(boxes [G,6|7] gravity-centre, labels [G]); dt_annos: list of dict(boxes, scores, labels).
    A detection is a true positive when its best 3D IoU with a not-yet-matched GT box of the same class in the same
    scene reaches the threshold (detections visited in descending score order, each GT matched once)."""
    if n_classes is None:
        n_classes = 1 + max([int(np.max(g["labels"])) for g in gt_annos if len(g["labels"])] +
                            [int(np.max(d["labels"])) for d in dt_annos if len(d["labels"])] + [0])
    result = {}
    for thr in iou_thrs:
        aps, recs = [], []
        for c in range(n_classes):
            scores, tps, n_gt = [], [], 0
            for g, d in zip(gt_annos, dt_annos):
                gb = np.asarray(g["boxes"], dtype=np.float32)[np.asarray(g["labels"]) == c]
                sel = np.asarray(d["labels"]) == c
                db = np.asarray(d["boxes"], dtype=np.float32)[sel]
                ds = np.asarray(d["scores"], dtype=np.float32)[sel]
                n_gt += len(gb)
                if len(db) == 0:
                    continue
                order = np.argsort(-ds, kind="stable")
                db, ds = db[order], ds[order]
                tp = np.zeros(len(db), dtype=bool)
                if len(gb):
                    rot = gb.shape[1] == 7 or db.shape[1] == 7
                    iou = box_iou(torch.from_numpy(db).to(device), torch.from_numpy(gb).to(device), rotated=rot,
                                  mode3d=True).cpu().numpy()
                    used = np.zeros(len(gb), dtype=bool)
                    for i in range(len(db)):
                        j = int(np.argmax(iou[i]))
                        if iou[i, j] > thr and not used[j]:      # strict, like mmdet3d's eval_det_cls
                            tp[i] = True
                            used[j] = True
                scores.append(ds)
                tps.append(tp)
            if n_gt == 0:
                continue
            if scores:
                s = np.concatenate(scores)
                t = np.concatenate(tps)[np.argsort(-s, kind="stable")]
                ctp, cfp = np.cumsum(t), np.cumsum(~t)
                rec, prec = ctp / n_gt, ctp / np.maximum(ctp + cfp, 1e-9)
                aps.append(average_precision(rec, prec))
                recs.append(float(rec[-1]) if len(rec) else 0.0)
            else:
                aps.append(0.0)
                recs.append(0.0)
        result[f"mAP_{thr:.2f}"] = float(np.mean(aps)) if aps else 0.0
        result[f"mAR_{thr:.2f}"] = float(np.mean(recs)) if recs else 0.0
        result[f"AP_{thr:.2f}"] = aps
    return result
