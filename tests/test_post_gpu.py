"""GPU tests of the post-processing row (NMS + mAP): HIP IoU / suppression mask vs the float64 polygon-clipping oracle."""
import numpy as np
import pytest
import torch

from oracle import post_oracle as PO

pytestmark = pytest.mark.gpu


def _boxes(rng, n, yaw=True):
    b = np.zeros((n, 7), dtype=np.float32)
    b[:, :3] = rng.rand(n, 3) * 4
    b[:, 3:6] = 0.3 + rng.rand(n, 3) * 1.2
    if yaw:
        b[:, 6] = rng.uniform(-3.2, 3.2, n)
    return b


@pytest.mark.parametrize("yaw", [False, True])
@pytest.mark.parametrize("mode3d", [False, True])
def test_iou_matrix_vs_oracle(device, yaw, mode3d):
    from cnrma_amd import postprocess as PP
    rng = np.random.RandomState(1)
    a, b = _boxes(rng, 40, yaw), _boxes(rng, 37, yaw)
    got = PP.box_iou(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), rotated=yaw, mode3d=mode3d).cpu().numpy()
    exp = np.array([[PO.iou(x, y, mode3d) for y in b] for x in a])
    np.testing.assert_allclose(got, exp, atol=2e-5)
    # identical boxes -> IoU 1, disjoint -> 0
    same = PP.box_iou(torch.from_numpy(a).to(device), torch.from_numpy(a).to(device), rotated=yaw, mode3d=mode3d).cpu().numpy()
    np.testing.assert_allclose(np.diag(same), 1.0, atol=1e-5)


@pytest.mark.parametrize("yaw", [False, True])
def test_nms_vs_oracle(device, yaw):
    from cnrma_amd import postprocess as PP
    rng = np.random.RandomState(2)
    b = _boxes(rng, 300, yaw)
    s = rng.rand(300).astype(np.float32)
    keep = PP.nms_single_class(torch.from_numpy(b if yaw else b[:, :6]).to(device), torch.from_numpy(s).to(device), 0.3, yaw)
    exp = PO.nms(b, s, 0.3)
    assert list(keep.cpu().numpy()) == list(exp)
    assert 10 < len(exp) < 300


def test_multiclass_nms_and_map(device):
    from cnrma_amd import postprocess as PP
    rng = np.random.RandomState(3)
    gt = _boxes(rng, 12, yaw=False)[:, :6]
    gl = rng.randint(0, 3, 12)
    # detections: every GT box jittered a few times (+ noise boxes), score = closeness
    det, sc = [], []
    for g, l in zip(gt, gl):
        for _ in range(4):
            d = g + rng.randn(6).astype(np.float32) * 0.03
            row = np.full(3, 0.001, dtype=np.float32)
            row[l] = 0.5 + 0.4 * rng.rand()
            det.append(d); sc.append(row)
    det += list(_boxes(rng, 20, yaw=False)[:, :6] + 10)
    sc += [np.array([0.2, 0.001, 0.001], dtype=np.float32)] * 20
    boxes, scores, labels = PP.nms(torch.tensor(np.array(det)).to(device), torch.tensor(np.array(sc)).to(device), 0.01, 0.5)
    assert boxes.shape[1] == 6 and len(boxes) == len(scores) == len(labels)
    res = PP.indoor_eval([dict(boxes=gt, labels=gl)], [dict(boxes=boxes.cpu().numpy(), scores=scores.cpu().numpy(),
                                                              labels=labels.cpu().numpy())], (0.25, 0.5), n_classes=3)
    assert res["mAP_0.25"] > 0.8 and res["mAR_0.25"] > 0.8 and 0 <= res["mAP_0.50"] <= 1
    # perfect detections -> AP 1
    res = PP.indoor_eval([dict(boxes=gt, labels=gl)], [dict(boxes=gt, scores=np.ones(12, np.float32), labels=gl)], (0.5,), 3)
    assert abs(res["mAP_0.50"] - 1.0) < 1e-6
