"""Differentiable IoU of rotated 3D boxes (about z), for the `IoU3DLoss(with_yaw=True)` of the ARKit configuration
(reference: fcaf3d_head.py:39,53-55 builds it through mmdet's loss registry; the FCAF3D fork implements it with
lilanxiao/Rotated_IoU @3bdca6b, doc/install.md:38-47 -- third party, not under /root/reference: parity unpinned, checked
against the float64 polygon-clipping oracle in tests/test_losses_cpu.py).

Boxes [n,7] = (cx, cy, cz, dx, dy, dz, yaw) with cz the gravity centre.  Bird's-eye-view intersection: the candidate
vertices of the intersection polygon are the 16 edge-edge crossings and the 8 corners lying inside the other rectangle;
they are sorted by angle around their centroid and fed to the shoelace formula (invalid candidates collapse onto a valid
vertex and add no area).  Everything is plain torch, so gradients reach all seven parameters of both boxes."""
import torch


def bev_corners(b):
    """[n,7] -> [n,4,2], counter-clockwise"""
    dx, dy, a = b[:, 3:4] / 2, b[:, 4:5] / 2, b[:, 6]
    lx = torch.cat((dx, -dx, -dx, dx), dim=1)
    ly = torch.cat((dy, dy, -dy, -dy), dim=1)
    c, s = torch.cos(a)[:, None], torch.sin(a)[:, None]
    return torch.stack((lx * c - ly * s + b[:, 0:1], lx * s + ly * c + b[:, 1:2]), dim=-1)


def _cross(u, v):
    return u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]


def _inside(pts, rect, eps=1e-9):
    """pts [n,4,2] inside (or on) rect [n,4,2] (corners in order): projections on two adjacent edges"""
    a, ab, ad = rect[:, None, 0], (rect[:, 1] - rect[:, 0])[:, None], (rect[:, 3] - rect[:, 0])[:, None]
    ap = pts - a
    pab, pad = (ap * ab).sum(-1), (ap * ad).sum(-1)
    return (pab >= -eps) & (pab <= (ab * ab).sum(-1) + eps) & (pad >= -eps) & (pad <= (ad * ad).sum(-1) + eps)


def bev_intersection(c1, c2):
    """area [n] of the intersection of the convex quadrilaterals c1, c2 [n,4,2]"""
    n = c1.shape[0]
    p1, p2 = c1[:, :, None], torch.roll(c1, -1, 1)[:, :, None]                   # edges of box 1: [n,4,1,2]
    p3, p4 = c2[:, None], torch.roll(c2, -1, 1)[:, None]                          # edges of box 2: [n,1,4,2]
    d12, d34 = p2 - p1, p4 - p3
    den = _cross(d12, d34)
    ok = den.abs() > 1e-12
    den = torch.where(ok, den, torch.ones_like(den))
    t = _cross(p3 - p1, d34) / den
    u = _cross(p3 - p1, d12) / den
    hit = ok & (t > 0) & (t < 1) & (u > 0) & (u < 1)
    cross_pts = (p1 + t[..., None] * d12).reshape(n, 16, 2)
    verts = torch.cat((cross_pts, c1, c2), dim=1)                                  # [n,24,2]
    mask = torch.cat((hit.reshape(n, 16), _inside(c1, c2), _inside(c2, c1)), dim=1)
    cnt = mask.sum(1, keepdim=True).clamp(min=1)
    centre = (verts * mask[..., None]).sum(1, keepdim=True) / cnt[..., None]
    rel = (verts - centre).detach()
    ang = torch.atan2(rel[..., 1], rel[..., 0])
    ang = torch.where(mask, ang, torch.full_like(ang, 1e3))                        # invalid candidates sort last
    order = torch.argsort(ang, dim=1)
    v = torch.gather(verts, 1, order[..., None].expand(-1, -1, 2))
    m = torch.gather(mask, 1, order)
    v = torch.where(m[..., None], v, v[:, :1])                                     # collapse the rest onto the first vertex
    nxt = torch.roll(v, -1, 1)
    area = 0.5 * (v[..., 0] * nxt[..., 1] - nxt[..., 0] * v[..., 1]).sum(1).abs()
    return torch.where(mask.sum(1) >= 3, area, torch.zeros_like(area))


def rotated_iou_3d(a, b):
    """IoU [n] of paired boxes a, b [n,7]"""
    inter = bev_intersection(bev_corners(a), bev_corners(b))
    zl = torch.max(a[:, 2] - a[:, 5] / 2, b[:, 2] - b[:, 5] / 2)
    zh = torch.min(a[:, 2] + a[:, 5] / 2, b[:, 2] + b[:, 5] / 2)
    iv = inter * (zh - zl).clamp(min=0)
    union = a[:, 3] * a[:, 4] * a[:, 5] + b[:, 3] * b[:, 4] * b[:, 5] - iv
    return iv / union.clamp(min=1e-8)
