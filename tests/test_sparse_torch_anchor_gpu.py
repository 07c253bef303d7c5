"""Independent anchor of the sparse operators (MinkowskiEngine is not available anywhere, so rows a9-a11
can only be *pinned* by an ME install or by the authors' checkpoint + mAP).  What can be done without ME: compare the
HIP operators DIRECTLY with torch's dense operators on the densified tensor at the active sites -- no
oracle/sparse_oracle.py in between, so HIP-vs-torch does not route through the builder's own restatement:

    sparse conv (k3, s1)        == F.conv3d(dense, padding=1) at the input sites           fcaf3d_backbone.py:26-31
    sparse conv (k3, s2)        == F.conv3d(dense, stride=2, padding=1) at floor(p/2)*2      fcaf3d_backbone.py:63-70
    max pool (k2, s2)           == F.max_pool3d(dense with -inf at inactive sites, 2, 2)    fcaf3d_backbone.py:53-55
    generative transpose (k2,s2)== F.conv_transpose3d(dense, stride=2)                      fcaf3d_head.py:72-83

torch's operators run in fp64 on the CPU: the tolerance is the north star's 1e-4 (relative to the tensor's scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-4
D = 24


def _case(seed, cin, ts=1, fill=0.25):
    g = torch.Generator().manual_seed(seed)
    occ = torch.rand(D, D, D, generator=g) < fill
    xyz = torch.nonzero(occ).to(torch.int32) * ts
    coords = torch.cat((torch.zeros(len(xyz), 1, dtype=torch.int32), xyz), dim=1)
    feats = torch.randn(len(xyz), cin, generator=g)
    dense = torch.zeros(1, cin, D, D, D, dtype=torch.float64)
    dense[0][:, occ] = feats.t().double()
    return coords, feats, dense, occ


def _dense_weight(w, k):
    """[K,Cin,Cout] in the kernel-offset order of the sparse engine (x fastest) -> conv3d weight [Cout,Cin,kx,ky,kz]"""
    K, cin, cout = w.shape
    wd = torch.zeros(cout, cin, k, k, k, dtype=torch.float64)
    for idx in range(K):
        ix, iy, iz = idx % k, (idx // k) % k, idx // (k * k)
        wd[:, :, ix, iy, iz] = w[idx].t().double()
    return wd


def _st(coords, feats, ts, device):
    from cnrma_amd import sparse as S
    return S.SparseTensor(feats.to(device), S.CoordSet(coords.to(device), ts))


def _at(dense_out, coords, div):
    c = (coords[:, 1:].long() // div)
    return dense_out[0][:, c[:, 0], c[:, 1], c[:, 2]].t()


def _close(got, exp):
    exp = exp.numpy()
    np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), exp, rtol=TOL, atol=TOL * max(1.0, float(np.abs(exp).max())))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64)])
def test_conv_k3_s1_vs_torch_conv3d(device, cin, cout, precision):
    from cnrma_amd import sparse as S
    coords, feats, dense, occ = _case(1, cin)
    w = torch.randn(27, cin, cout, generator=torch.Generator().manual_seed(2)) / np.sqrt(27 * cin)
    y = S.conv(_st(coords, feats, 1, device), w.to(device), kernel_size=3, stride=1, precision=precision)
    assert torch.equal(y.C.cpu(), coords)
    _close(y.F, _at(F.conv3d(dense, _dense_weight(w, 3), padding=1), coords, 1))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_conv_k3_s2_vs_torch_strided_conv3d(device, precision):
    from cnrma_amd import sparse as S
    coords, feats, dense, occ = _case(3, 32)
    w = torch.randn(27, 32, 64, generator=torch.Generator().manual_seed(4)) / np.sqrt(27 * 32)
    y = S.conv(_st(coords, feats, 1, device), w.to(device), kernel_size=3, stride=2, precision=precision)
    out_c = y.C.cpu()
    # output sites: every even site with an active input in its 2x2x2 cell (floor(p/2)*2), nothing else
    cells = torch.unique(coords[:, 1:] // 2 * 2, dim=0)
    assert y.cs.stride == 2 and sorted(map(tuple, out_c[:, 1:].tolist())) == sorted(map(tuple, cells.tolist()))
    _close(y.F, _at(F.conv3d(dense, _dense_weight(w, 3), stride=2, padding=1), out_c, 2))


def test_max_pool_vs_torch_max_pool3d(device):
    from cnrma_amd import sparse as S
    coords, feats, dense, occ = _case(5, 32)
    neg = torch.full_like(dense, -float("inf"))
    neg[0][:, occ] = dense[0][:, occ]
    y = S.max_pool(_st(coords, feats, 1, device), kernel_size=2, stride=2)
    out_c = y.C.cpu()
    exp = _at(F.max_pool3d(neg, 2, 2), out_c, 2)
    assert bool(torch.isfinite(exp).all())
    assert torch.equal(y.F.cpu().double(), exp)              # a maximum of fp32 values: exact


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_generative_transpose_vs_torch_conv_transpose3d(device, precision):
    from cnrma_amd import sparse as S
    coords, feats, dense, occ = _case(7, 64, ts=2, fill=0.15)          # coarse tensor at stride 2 -> children at stride 1
    w = torch.randn(8, 64, 32, generator=torch.Generator().manual_seed(8)) / 8.0
    y = S.conv_transpose_generative(_st(coords, feats, 2, device), w.to(device), precision=precision)
    out_c = y.C.cpu()
    assert y.cs.stride == 1 and len(out_c) == 8 * len(coords) and len(torch.unique(out_c, dim=0)) == len(out_c)
    # conv_transpose3d weight: [Cin, Cout, kx, ky, kz]; child offset k decodes with x fastest
    wt = torch.zeros(64, 32, 2, 2, 2, dtype=torch.float64)
    for k in range(8):
        wt[:, :, k & 1, (k >> 1) & 1, (k >> 2) & 1] = w[k].double()
    up = F.conv_transpose3d(dense, wt, stride=2)                        # dense index = coarse index (coordinate / 2)
    _close(y.F, _at(up, out_c, 1))
