"""Rotated IoU3D loss (ARKit configuration) against the float64 polygon-clipping oracle, and its
gradients against finite differences."""
import numpy as np
import torch

from oracle import post_oracle as PO


def _boxes(rng, n, spread=0.6):
    b = np.zeros((n, 7))
    b[:, :3] = rng.randn(n, 3) * spread
    b[:, 3:6] = 0.3 + rng.rand(n, 3) * 1.5
    b[:, 6] = (rng.rand(n) - 0.5) * 2 * np.pi
    return b


def test_rotated_iou_matches_polygon_clipping_oracle():
    from projects.mvsdetection.core.rotated_iou import rotated_iou_3d
    rng = np.random.RandomState(0)
    a, b = _boxes(rng, 400), _boxes(rng, 400)
    # special cases: identical, axis-aligned pair, far apart, one inside the other, same centre other angle
    a[0] = b[0]
    a[1, 6] = b[1, 6] = 0.0
    b[2, :3] = a[2, :3] + 50
    b[3, :3], b[3, 3:6], b[3, 6] = a[3, :3], a[3, 3:6] * 0.4, a[3, 6] + 0.3
    b[4, :6], b[4, 6] = a[4, :6], a[4, 6] + 0.7
    got = rotated_iou_3d(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    exp = np.array([PO.iou(x, y, mode3d=True) for x, y in zip(a, b)])
    np.testing.assert_allclose(got, exp, rtol=1e-9, atol=1e-10)
    assert abs(got[0] - 1) < 1e-12 and got[2] == 0 and abs(got[3] - 0.4 ** 3) < 1e-12 and 0 < got[4] < 1
    assert (got > 0.05).sum() > 50            # the random pairs do overlap


def test_rotated_iou_gradients_match_finite_differences():
    from projects.mvsdetection.core.rotated_iou import rotated_iou_3d
    rng = np.random.RandomState(1)
    a, b = _boxes(rng, 60, spread=0.3), _boxes(rng, 60, spread=0.3)
    ta = torch.from_numpy(a).requires_grad_(True)
    tb = torch.from_numpy(b)
    iou = rotated_iou_3d(ta, tb)
    keep = (iou > 0.05) & (iou < 0.95)
    assert int(keep.sum()) > 20
    iou[keep].sum().backward()
    g = ta.grad.numpy()
    eps = 1e-6
    for i in np.nonzero(keep.numpy())[0][:12]:
        for j in range(7):
            ap, am = a.copy(), a.copy()
            ap[i, j] += eps
            am[i, j] -= eps
            fd = (PO.iou(ap[i], b[i]) - PO.iou(am[i], b[i])) / (2 * eps)
            assert abs(fd - g[i, j]) < 1e-5 * max(1.0, abs(fd)), (i, j, fd, g[i, j])


def test_iou3d_loss_module_axis_aligned_and_rotated():
    from projects.mvsdetection.models.fcaf3d_head import _IoU3DLoss
    rng = np.random.RandomState(2)
    a, b = _boxes(rng, 30, 0.2).astype(np.float32), _boxes(rng, 30, 0.2).astype(np.float32)
    w = torch.rand(30)
    rot = _IoU3DLoss(loss_weight=1.0, with_yaw=True)(torch.from_numpy(a), torch.from_numpy(b), weight=w, avg_factor=7.0)
    exp = sum(float(w[i]) * (1 - PO.iou(a[i], b[i])) for i in range(30)) / 7.0
    assert abs(float(rot) - exp) < 1e-4
    a[:, 6] = b[:, 6] = 0
    ax = _IoU3DLoss(with_yaw=False)(torch.from_numpy(a[:, :6]), torch.from_numpy(b[:, :6]), weight=w, avg_factor=7.0)
    exp = sum(float(w[i]) * (1 - PO.iou(a[i], b[i])) for i in range(30)) / 7.0
    assert abs(float(ax) - exp) < 1e-4
