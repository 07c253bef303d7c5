"""CPU suite: the sparse-half oracle (parity unpinned: ME is absent) is anchored on dense equivalences
(SURVEY.md 8c): sparse conv == conv3d on the densified grid at the active output sites, etc.; the decoder is
checked against the golden vectors produced by the reference's fcaf3d_head.py."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse_oracle as SO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def random_sparse(rng, D=10, occ=0.3, C=5, ts=1, batch=1, neg=False):
    cs = []
    for b in range(batch):
        m = rng.rand(D, D, D) < occ
        xyz = np.argwhere(m)
        rng.shuffle(xyz)
        if neg:
            xyz = xyz - D // 2
        cs.append(np.concatenate((np.full((len(xyz), 1), b), xyz * ts), axis=1))
    c = np.concatenate(cs).astype(np.int64)
    return c, rng.randn(len(c), C)


def densify(coords, feats, D, ts, fill=0.0, shift=0):
    g = np.full((feats.shape[1], D, D, D), fill)
    i = coords[:, 1:] // ts + shift
    g[:, i[:, 0], i[:, 1], i[:, 2]] = feats.T
    return torch.from_numpy(g)[None]


def dense_kernel(W, k):
    """W[kidx][ci][co] with kidx = ix + k*iy + k*k*iz  ->  [co, ci, kx, ky, kz]"""
    K, ci, co = W.shape
    return torch.from_numpy(W.reshape(k, k, k, ci, co).transpose(4, 3, 2, 1, 0).copy())   # (iz,iy,ix,..) -> (co,ci,ix,iy,iz)


@pytest.mark.parametrize("ts", [1, 2])
def test_conv_k3_stride1_equals_dense_conv3d(ts):
    rng = np.random.RandomState(0)
    c, f = random_sparse(rng, D=9, C=4, ts=ts)
    W = rng.randn(27, 4, 6)
    oc, of = SO.conv(c, f, W, 3, 1, ts)
    dense = F.conv3d(densify(c, f, 9, ts), dense_kernel(W, 3), padding=1)[0].numpy()
    i = oc[:, 1:] // ts
    np.testing.assert_allclose(of, dense[:, i[:, 0], i[:, 1], i[:, 2]].T, atol=1e-10)


def test_conv_k3_stride2_equals_dense_conv3d():
    rng = np.random.RandomState(1)
    c, f = random_sparse(rng, D=10, C=3, ts=1)
    W = rng.randn(27, 3, 5)
    oc, of = SO.conv(c, f, W, 3, 2, 1)
    assert (oc[:, 1:] % 2 == 0).all() and len(np.unique(SO._key(oc))) == len(oc)
    assert set(map(tuple, oc)) == set(map(tuple, np.concatenate((c[:, :1], c[:, 1:] // 2 * 2), axis=1)))
    dense = F.conv3d(densify(c, f, 10, 1), dense_kernel(W, 3), padding=1, stride=2)[0].numpy()
    i = oc[:, 1:] // 2
    np.testing.assert_allclose(of, dense[:, i[:, 0], i[:, 1], i[:, 2]].T, atol=1e-10)


def test_conv_k1_stride2_takes_only_the_coincident_input():
    rng = np.random.RandomState(2)
    c, f = random_sparse(rng, D=8, C=3)
    W = rng.randn(3, 4)
    oc, of = SO.conv(c, f, W, 1, 2, 1)
    look = SO.Lookup(c)
    idx = look(oc)
    exp = np.where((idx >= 0)[:, None], f[np.maximum(idx, 0)] @ W, 0.0)
    np.testing.assert_allclose(of, exp, atol=1e-12)


def test_maxpool_equals_dense_maxpool3d():
    rng = np.random.RandomState(3)
    c, f = random_sparse(rng, D=10, C=4, ts=2)
    oc, of = SO.max_pool(c, f, 2)
    dense = F.max_pool3d(densify(c, f, 10, 2, fill=-np.inf), 2, 2)[0].numpy()
    i = oc[:, 1:] // 4
    np.testing.assert_allclose(of, dense[:, i[:, 0], i[:, 1], i[:, 2]].T)


def test_generative_transpose_equals_dense_conv_transpose3d():
    rng = np.random.RandomState(4)
    c, f = random_sparse(rng, D=6, C=3, ts=4)
    W = rng.randn(8, 3, 5)
    oc, of = SO.conv_transpose_generative(c, f, W, 4)
    assert len(oc) == 8 * len(c) and len(np.unique(SO._key(oc))) == len(oc)
    wd = torch.from_numpy(W.reshape(2, 2, 2, 3, 5).transpose(3, 4, 2, 1, 0).copy())     # [ci, co, ix, iy, iz]
    dense = F.conv_transpose3d(densify(c, f, 6, 4), wd, stride=2)[0].numpy()
    i = oc[:, 1:] // 2
    np.testing.assert_allclose(of, dense[:, i[:, 0], i[:, 1], i[:, 2]].T, atol=1e-10)


def test_negative_coordinates_floor_toward_minus_infinity():
    c = np.array([[0, -1, -1, -1], [0, -2, 0, 3], [0, 1, 1, 1], [0, -3, -4, 5]])
    oc = SO.stride_coords(c, 2)
    assert list(map(tuple, oc)) == [(0, -2, -2, -2), (0, -2, 0, 2), (0, 0, 0, 0), (0, -4, -4, 4)]


def test_known_answers():
    # single active voxel: k3 conv sees only the centre tap (kidx 13)
    c = np.array([[0, 4, 4, 4]])
    f = np.array([[2.0]])
    W = np.arange(27, dtype=np.float64).reshape(27, 1, 1)
    _, o = SO.conv(c, f, W, 3, 1, 1)
    assert o[0, 0] == 2.0 * 13
    # kernel offset order: x fastest.  neighbour at +x is kidx 14, at +y kidx 16, at +z kidx 22
    for d, k in (((1, 0, 0), 14), ((0, 1, 0), 16), ((0, 0, 1), 22)):
        c2 = np.array([[0, 4, 4, 4], [0, 4 + d[0], 4 + d[1], 4 + d[2]]])
        f2 = np.array([[0.0], [1.0]])
        _, o = SO.conv(c2, f2, W, 3, 1, 1)
        assert o[0, 0] == k
    # 2x2x2 full block pooled to one site
    blk = np.array([[0, x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)])
    oc, o = SO.max_pool(blk, np.arange(8, dtype=np.float64).reshape(8, 1), 1)
    assert len(oc) == 1 and o[0, 0] == 7
    # two scenes never mix
    c3 = np.array([[0, 0, 0, 0], [1, 1, 0, 0]])
    _, o = SO.conv(c3, np.ones((2, 1)), np.ones((27, 1, 1)), 3, 1, 1)
    assert (o == 1).all()
    # union-add
    uc, uf = SO.union_add(np.array([[0, 0, 0, 0], [0, 2, 0, 0]]), np.array([[1.0], [2.0]]),
                          np.array([[0, 2, 0, 0], [0, 4, 0, 0]]), np.array([[10.0], [20.0]]))
    assert list(map(tuple, uc)) == [(0, 0, 0, 0), (0, 2, 0, 0), (0, 4, 0, 0)] and list(uf[:, 0]) == [1, 12, 20]
    # interpolation weights on the half-stride lattice are 1, 1/2, 1/4, 1/8; missing corners contribute 0
    sc = np.array([[0, 0, 0, 0], [0, 4, 0, 0]])
    out = SO.interpolate(sc, np.array([[1.0], [3.0]]), 4, np.array([[0, 0, 0, 0], [0, 2, 0, 0], [0, 2, 2, 0], [0, 2, 2, 2]]))
    np.testing.assert_allclose(out[:, 0], [1.0, 2.0, 1.0, 0.5])


def test_decode_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "decode.npz"))
    pts = torch.from_numpy(z["points"])
    for nreg, yaw in ((6, "fcaf3d"), (8, "fcaf3d"), (8, "sin-cos"), (7, "naive")):
        box = SO.decode_boxes(pts, torch.from_numpy(z[f"pred_{nreg}_{yaw}"]), yaw)
        assert torch.equal(box, torch.from_numpy(z[f"box_{nreg}_{yaw}"]))


def test_compute_centerness_matches_reference_golden():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from projects.mvsdetection.models.fcaf3d_head import compute_centerness
    z = np.load(os.path.join(GOLDEN, "decode.npz"))
    assert torch.equal(compute_centerness(torch.from_numpy(z["centerness_in"])), torch.from_numpy(z["centerness_out"]))


def test_fp32_torch_port_matches_the_fp64_oracle():
    """oracle/sparse_torch.py (the multi-threaded fp32 restatement timed as bench.py's cpu_baseline) against the fp64
    numpy oracle on a random surface-like point set: every level's coordinate set identical, head outputs close"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import sparse_torch as ST
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    from projects.mvsdetection.models.fcaf3d_head import FCAF3DHead
    rng = np.random.RandomState(0)
    n = 4000
    uv = rng.randint(0, 160, size=(n, 2))
    z = (40 + 8 * np.sin(uv[:, 0] / 17.0) + rng.randint(0, 3, size=n)).astype(np.int64)
    coords = np.unique(np.concatenate((np.zeros((n, 1), dtype=np.int64), uv, z[:, None]), axis=1), axis=0)
    feats = rng.randn(len(coords), 8).astype(np.float32)
    torch.manual_seed(0)
    backbone = FCAF3DBackbone(8, 34).eval()
    head = FCAF3DHead(18, (64, 128, 256, 512), 128, 6, 0.01, 300, None, test_cfg=dict(nms_pre=50)).eval()
    backbone.init_weights()
    head.init_weights()
    lv64 = SO.backbone_forward(backbone, coords, feats)
    lv32 = ST.backbone_forward(backbone, coords, feats)
    for (c64, f64, _), (cs32, f32) in zip(lv64, lv32):
        assert np.array_equal(c64, cs32.C.numpy())
        np.testing.assert_allclose(f32.numpy(), f64, rtol=2e-3, atol=2e-3)
    r64 = SO.head_forward(head, lv64)
    r32 = ST.head_forward(head, lv32)
    for a, b in zip(r64, r32):
        assert a["coords"].shape == tuple(b["coords"].shape)
        ka = np.lexsort(a["coords"].T[::-1])
        kb = np.lexsort(b["coords"].numpy().T[::-1])
        same = (a["coords"][ka] == b["coords"].numpy()[kb]).all(axis=1)
        assert same.mean() > 0.99              # ties at the pruning threshold may differ
        np.testing.assert_allclose(b["cls_score"].numpy()[kb][same], a["cls_score"][ka][same], rtol=5e-3, atol=5e-3)
    b32, s32 = ST.get_bboxes(head, r32)
    assert torch.isfinite(b32).all() and torch.isfinite(s32).all()
