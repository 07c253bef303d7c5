"""ORACLE (test infrastructure only) for the post-processing row: BEV / 3D IoU of (rotated)
boxes by polygon clipping in float64, greedy NMS, AP.  PARITY UNPINNED: the reference calls mmdet3d's pcdet_nms_* /
indoor_eval (OpenPCDet iou3d_nms semantics), which are not under /root/reference; call sites:
post_process/nms_bbox.py:29-35, evaluate_bbox.py:93-100."""
import numpy as np


def _corners(b):
    x, y, dx, dy, a = b[0], b[1], b[3], b[4], (b[6] if len(b) > 6 else 0.0)
    c, s = np.cos(a), np.sin(a)
    loc = np.array([[dx / 2, dy / 2], [-dx / 2, dy / 2], [-dx / 2, -dy / 2], [dx / 2, -dy / 2]])
    return loc @ np.array([[c, s], [-s, c]]) + np.array([x, y])


def _clip(poly, p, q):
    out = []
    for i in range(len(poly)):
        s, t = poly[i], poly[(i + 1) % len(poly)]
        ds = (q[0] - p[0]) * (s[1] - p[1]) - (q[1] - p[1]) * (s[0] - p[0])
        dt = (q[0] - p[0]) * (t[1] - p[1]) - (q[1] - p[1]) * (t[0] - p[0])
        if ds >= 0:
            out.append(s)
        if ds * dt < 0:
            out.append(s + (t - s) * ds / (ds - dt))
    return out


def bev_intersection(a, b):
    poly = list(_corners(a))
    cb = _corners(b)
    for e in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, cb[e], cb[(e + 1) % 4])
    if len(poly) < 3:
        return 0.0
    p = np.array(poly)
    return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - p[:, 1] * np.roll(p[:, 0], -1)))


def iou(a, b, mode3d=True):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    inter = bev_intersection(a, b)
    if not mode3d:
        return inter / max(a[3] * a[4] + b[3] * b[4] - inter, 1e-8)
    zl, zh = max(a[2] - a[5] / 2, b[2] - b[5] / 2), min(a[2] + a[5] / 2, b[2] + b[5] / 2)
    iv = inter * max(zh - zl, 0.0)
    return iv / max(a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - iv, 1e-8)


def nms(boxes, scores, thr):
    order = np.argsort(-scores, kind="stable")
    keep = []
    for i in order:
        if all(iou(boxes[i], boxes[j], mode3d=False) <= thr for j in keep):
            keep.append(i)
    return np.array(keep, dtype=np.int64)
