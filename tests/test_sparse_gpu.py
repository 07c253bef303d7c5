"""GPU parity tests (-m gpu) of the sparse half: HIP engine (through the C-ABI) vs oracle/sparse_oracle.py.
Coordinates / index outputs bit-exact; features within 1e-4 (north_star), compared after sorting by coordinate."""
import os

import numpy as np
import pytest
import torch

from helpers import SCENES, elementwise_error, load_golden, t
from oracle import rma_oracle as RO
from oracle import sparse_oracle as SO

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_sparse(rng, n=3000, span=40, C=32, ts=1, batch=1, neg=True):
    xyz = rng.randint(-span if neg else 0, span, size=(n, 3)) * ts
    b = rng.randint(0, batch, size=(n, 1))
    c = np.concatenate((b, xyz), axis=1).astype(np.int64)
    c = c[SO.unique_first(c)]
    return c, rng.randn(len(c), C).astype(np.float32)


def to_st(c, f, ts, device):
    from cnrma_amd import sparse as S
    return S.SparseTensor(torch.from_numpy(f).to(device), S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), ts))


def sort_rows(c, f):
    o = np.argsort(SO._key(c), kind="stable")
    return np.asarray(c)[o], np.asarray(f)[o]


def check(st, oc, of, tol=TOL, same_order=True):
    c = st.C.cpu().numpy().astype(np.int64)
    f = st.F.cpu().numpy()
    assert c.shape == oc.shape, (c.shape, oc.shape)
    if same_order:
        assert (c == oc).all(), "row order differs from the oracle's first-occurrence order"
    c1, f1 = sort_rows(c, f)
    c2, f2 = sort_rows(oc, of)
    assert (c1 == c2).all()
    scale = max(1.0, float(np.abs(f2).max()))
    np.testing.assert_allclose(f1, f2, rtol=tol, atol=tol * scale)


@pytest.mark.parametrize("name", SCENES)
def test_voxelize_vs_golden(device, name):
    from cnrma_amd import sparse as S
    g = load_golden(name)
    st, src = S.voxelize(t(g["sel_coords"], device), t(g["sel_feats"], device), 0.01, row_order="first")
    assert (st.C.cpu().numpy() == g["vox_coords"]).all()
    assert (src.cpu().numpy() == g["vox_src"]).all()
    assert torch.equal(st.F.cpu(), t(g["sel_feats"])[t(g["vox_src"]).long()])
    # Morton row order: same voxels, same representatives, rows sorted by the interleaved-bit key
    sm, srcm = S.voxelize(t(g["sel_coords"], device), t(g["sel_feats"], device), 0.01, row_order="morton")
    cm = sm.C.cpu().numpy().astype(np.int64)
    assert sorted(map(tuple, cm)) == sorted(map(tuple, g["vox_coords"]))
    assert sorted(srcm.cpu().numpy()) == sorted(g["vox_src"])
    assert torch.equal(sm.F.cpu(), t(g["sel_feats"])[srcm.cpu().long()])

    def morton(c):
        k = np.zeros(len(c), dtype=np.uint64)
        for bit in range(16):
            for ax, sh in ((1, 2), (2, 1), (3, 0)):
                k |= (((c[:, ax] + 32768) >> bit) & 1).astype(np.uint64) << np.uint64(3 * bit + sh)
        return k
    mk = morton(cm)
    assert (np.diff(mk.astype(np.float64)) > 0).all()


def test_voxelize_duplicates_negative_and_first_wins(device):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(0)
    coords = (rng.rand(20000, 3).astype(np.float32) - 0.5) * 0.6          # heavy duplication at 1 cm, negatives
    feats = rng.randn(20000, 8).astype(np.float32)
    st, src = S.voxelize(torch.from_numpy(coords).to(device), torch.from_numpy(feats).to(device), 0.01, batch_id=3,
                         row_order="first")
    Cq, Fq, first = RO.voxelize(torch.from_numpy(coords), torch.from_numpy(feats), 0.01, batch_id=3)
    assert (st.C.cpu() == Cq).all() and (src.cpu().long() == first).all() and torch.equal(st.F.cpu(), Fq)
    assert len(st) < 20000


@pytest.mark.parametrize("cin,cout,k,stride,ts", [(32, 64, 3, 1, 1), (64, 64, 3, 1, 2), (32, 64, 3, 2, 1), (64, 128, 3, 2, 4),
                                                   (64, 128, 1, 2, 2), (128, 25, 1, 1, 1), (5, 7, 3, 1, 1), (48, 96, 3, 1, 1),
                                                   (256, 256, 3, 1, 2), (8, 3, 1, 1, 1)])
@pytest.mark.parametrize("precision", ["f32", "bf16x6", "f16x3"])
def test_conv_vs_oracle(device, cin, cout, k, stride, ts, precision):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + cout + k)
    c, f = rand_sparse(rng, n=4000, span=12, C=cin, ts=ts, batch=2)
    W = (rng.randn(k ** 3, cin, cout) / np.sqrt(cin * k ** 3)).astype(np.float32)
    out = S.conv(to_st(c, f, ts, device), torch.from_numpy(W).to(device), k, stride, precision=precision)
    oc, of = SO.conv(c, f, W, k, stride, ts)
    check(out, oc, of, tol=2e-6)   # all three paths are fp32-grade at this length (f16x3: 22-bit operands)
    assert out.cs.stride == ts * stride


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (256, 512)])
def test_forced_tile_shapes_and_prefetch_depths_are_bit_identical(device, cin, cout):
    """every tile shape x prefetch depth of the f16x3 kernel (sparse.conv_tuning) computes a row with the same products
    in the same (offset, channel-slice) order: bit-identical outputs without a split, 1e-6 with one (slab sums reorder)"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + cout)
    c, f = rand_sparse(rng, n=6000, span=14, C=cin, ts=1)
    W = torch.from_numpy((rng.randn(27, cin, cout) / np.sqrt(cin * 27)).astype(np.float32)).to(device)
    x = to_st(c, f, 1, device)
    try:
        S.conv_tuning("64x64", 1, 1)
        ref = S.conv(x, W, 3, 1, act="relu").F.clone()
        seen = set()
        for shape in ("64x64", "128x64", "64x128", "128x128"):
            for pf in (1, 2):
                S.conv_tuning(shape, 1, pf)
                plan = S.conv_plan(x.cs.n, cin, cout, 27)
                assert plan["shape"] == shape and plan["splits"] == 1
                seen.add((plan["shape"], plan["prefetch"]))
                assert torch.equal(S.conv(x, W, 3, 1, act="relu").F, ref), (shape, pf)
        assert ("64x128", 2) in seen and ("128x64", 2) in seen and ("128x128", 1) in seen
        for splits in (3, 9, 27):
            S.conv_tuning("64x128", splits, 2)
            assert S.conv_plan(x.cs.n, cin, cout, 27)["splits"] == splits
            got = S.conv(x, W, 3, 1, act="relu").F
            assert torch.allclose(got, ref, rtol=1e-5, atol=5e-6 * float(ref.abs().max()))
    finally:
        S.conv_tuning()
    oc, of = SO.conv(c, f, W.cpu().numpy(), 3, 1, 1)
    check(S.SparseTensor(ref, x.cs), oc, np.maximum(of, 0.0), tol=2e-6)


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (128, 128), (256, 512)])
def test_warp_specialised_kernel_is_bit_identical_to_the_stage_kernel(device, cin, cout):
    """sparse.conv_tuning(ws=N): the 8-wave producer / consumer form of the f16x3 kernel (N-slot LDS ring, per-slot counters
    instead of block barriers) runs the same stages in the same order -- bit-identical outputs for every tile shape and
    ring depth, with the fused epilogue (residual + ReLU), with a split over the offsets, on a stride-2 map and on the
    identity map (K = 1)"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin * 3 + cout)
    c, f = rand_sparse(rng, n=7000, span=13, C=cin, ts=1)
    W = torch.from_numpy((rng.randn(27, cin, cout) / np.sqrt(cin * 27)).astype(np.float32)).to(device)
    W1 = torch.from_numpy((rng.randn(cin, cout) / np.sqrt(cin)).astype(np.float32)).to(device)
    x = to_st(c, f, 1, device)
    res = torch.from_numpy(rng.randn(len(c), cout).astype(np.float32)).to(device)
    try:
        for shape in ("64x64", "128x64", "64x128", "128x128"):
            for splits in (1, 9):
                S.conv_tuning(shape, splits, 1, 0, 0)
                ref = S.conv(x, W, 3, 1, residual=res, act="relu").F.clone()
                ref2 = S.conv(x, W, 3, 2).F.clone()
                ref1 = S.conv(x, W1, 1, 1, act="elu").F.clone()
                for slots in (2, 3, 4):
                    S.conv_tuning(shape, splits, 1, 0, slots)
                    for _ in range(2):
                        assert torch.equal(S.conv(x, W, 3, 1, residual=res, act="relu").F, ref), (shape, splits, slots)
                    assert torch.equal(S.conv(x, W, 3, 2).F, ref2), (shape, splits, slots, "stride 2")
                    got1 = S.conv(x, W1, 1, 1, act="elu")
                    assert torch.equal(got1.F, ref1), (shape, splits, slots, "K = 1")
                    assert float(got1.amax.max()) == float(ref1.abs().max())
                if shape == "128x128":                      # the 256-row tiles exist in the warp-specialised form only (8 consumer waves)
                    for big in ("256x128", "256x64"):
                        S.conv_tuning(big, splits, 1, 0, 2)
                        assert S.conv_plan(x.cs.n, cin, cout, 27)["shape"] == big
                        assert torch.equal(S.conv(x, W, 3, 1, residual=res, act="relu").F, ref), (big, splits)
                        assert torch.equal(S.conv(x, W, 3, 2).F, ref2), (big, splits, "stride 2")
    finally:
        S.conv_tuning()


@pytest.mark.parametrize("cin,cout,n,span", [(64, 64, 6000, 14), (128, 128, 9000, 20), (256, 512, 1500, 9), (64, 128, 20000, 14)])
def test_gather_once_convolution_vs_oracle_and_stage_kernel(device, cin, cout, n, span):
    """the gather-once form of the 3x3x3 stride-1 convolution (sparse.GO_CONV: per 64-row tile the distinct input rows are
    staged once per channel slice, the 27 offsets run from LDS; weights in MFMA-fragment order) against the fp64 oracle
    (2e-6) and the stage kernel (same products, another summation order), with residual + ReLU epilogue, with a split over
    channel slices (short layer) and on a dense point set in random row order (no locality: every tile needs several offset
    groups).  Also checks the tile unions themselves: every (row, offset) entry resolves to the neighbour table's row."""
    from cnrma_amd import _lib
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + n)
    c, f = rand_sparse(rng, n=n, span=span, C=cin, ts=1)
    W = torch.from_numpy((rng.randn(27, cin, cout) / np.sqrt(cin * 27)).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.randn(len(c), cout).astype(np.float32)).to(device)
    x = to_st(c, f, 1, device)
    prev = S.GO_CONV
    try:
        S.GO_CONV = False
        ref = S.conv(x, W, 3, 1, residual=res, act="relu").F.clone()
        S.GO_CONV = True
        got = S.conv(x, W, 3, 1, residual=res, act="relu")
        plain = S.conv(x, W, 3, 1).F
    finally:
        S.GO_CONV = prev
    assert torch.allclose(got.F, ref, rtol=1e-5, atol=5e-6 * float(ref.abs().max()))
    oc, of = SO.conv(c, f, W.cpu().numpy(), 3, 1, 1)
    check(S.SparseTensor(plain, x.cs), oc, of, tol=2e-6)
    assert float(got.amax.max()) == float(got.F.abs().max())                     # the magnitude bound travels as before
    # ---- the tile unions
    nbr = x.cs.neighbours(x.cs, 3, 1).cpu().numpy()
    tu = S.tile_union(x.cs, x.cs, 3, 1).cpu()
    n_t = (len(c) + 63) // 64
    al = lambda b: (b + 255) // 256 * 256
    hdr = tu[:n_t * 84 * 4].view(torch.int32).view(n_t, 84).numpy()
    rows = tu[al(n_t * 84 * 4):al(n_t * 84 * 4) + n_t * 1728 * 4].view(torch.int32).view(n_t, 1728).numpy()
    o2 = al(n_t * 84 * 4) + al(n_t * 1728 * 4)
    lidx = tu[o2:o2 + n_t * 1728 * 2].view(torch.int16).view(n_t, 64, 27).numpy().astype(np.int64) & 0xFFFF
    for t in rng.choice(n_t, size=min(n_t, 12), replace=False):
        seen = 0
        for g in range(hdr[t, 0]):
            mask, ub, un = int(hdr[t, 1 + 3 * g]) & 0xFFFFFFFF, hdr[t, 2 + 3 * g], hdr[t, 3 + 3 * g]
            assert 0 < un <= S.GO_UMAX and (mask & seen) == 0
            seen |= mask
            u = rows[t, ub:ub + un]
            first = []                                                          # distinct, in the order of first appearance
            for k in range(27):                                                 # over (offset, row of the tile)
                if (mask >> k) & 1:
                    first += [w for w in nbr[64 * t:64 * t + 64, k] if w >= 0]
            assert list(u) == list(dict.fromkeys(first))
            for k in range(27):
                if (mask >> k) & 1:
                    for r in range(min(64, len(c) - 64 * t)):
                        want = nbr[64 * t + r, k]
                        assert (lidx[t, r, k] == S.GO_UMAX) if want < 0 else (u[lidx[t, r, k]] == want)
        live = nbr[64 * t:64 * t + 64]
        assert all(((seen >> k) & 1) == int((live[:, k] >= 0).any()) for k in range(27))
    if n >= 20000:
        assert hdr[:, 0].max() > 1                                              # no locality: several groups per tile


@pytest.mark.parametrize("cin,cout,n,span", [(64, 64, 6000, 14), (128, 128, 9000, 20), (256, 512, 1500, 9), (64, 128, 20000, 14),
                                             (64, 64, 64 * 37 + 5, 12)])
def test_gather_once_second_form_is_bit_identical_to_the_first(device, cin, cout, n, span):
    """sparse.conv_tuning(go=1): the second form of the gather-once kernel (metadata requested up front, row numbers of group
    0 cached in LDS, scalar offset loop with SGPR-based weight loads, local indices one offset ahead, one-dimensional grid
    decoded per XCD, 2 or 4 weight offsets in flight) and go=2, the third form (persistent blocks, two images: the next stage's
    union rows are gathered under the current stage's MFMAs with hand-counted waits, fragments read one offset ahead) keep the
    first form's summation order: bit-identical outputs and magnitude bound in every work order, with residual + ReLU, split
    over channel slices, on a point set without locality (several offset groups per tile) and with a ragged last tile"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + n)
    c, f = rand_sparse(rng, n=n, span=span, C=cin, ts=1)
    W = torch.from_numpy((rng.randn(27, cin, cout) / np.sqrt(cin * 27)).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.randn(len(c), cout).astype(np.float32)).to(device)
    x = to_st(c, f, 1, device)
    prev = S.GO_CONV
    try:
        S.GO_CONV = True
        for splits in (-1, 1, 2):
            S.conv_tuning(splits=splits, go=0)
            ref = S.conv(x, W, 3, 1, residual=res, act="relu")
            ref_f, ref_amax = ref.F.clone(), float(ref.amax.max())
            plain = S.conv(x, W, 3, 1).F.clone()
            # nb = 12: the fragment-look-ahead instantiation of the second form (round 6: A fragments one offset ahead); go = 2: the third form
            for go, nb, xcd in [(1, nb_, x_) for nb_ in (2, 4, 12) for x_ in (0, 1, 2)] + [(2, -1, -1)]:
                if True:
                    if True:
                        S.conv_tuning(splits=splits, go=go, nb=nb, xcd=xcd)
                        got = S.conv(x, W, 3, 1, residual=res, act="relu")
                        assert torch.equal(got.F, ref_f), (splits, go, nb, xcd)
                        assert float(got.amax.max()) == ref_amax
                        assert torch.equal(S.conv(x, W, 3, 1).F, plain), (splits, go, nb, xcd, "plain")
    finally:
        S.conv_tuning()
        S.GO_CONV = prev


@pytest.mark.parametrize("cin,cout,n,span", [(64, 64, 6000, 14), (128, 128, 9000, 20), (256, 512, 1500, 9), (64, 128, 20000, 14),
                                             (64, 64, 64 * 37 + 5, 12)])
def test_exact_fp32_gather_once_convolution_vs_oracle_and_stage_kernel(device, cin, cout, n, span):
    """CONV_PRECISION = "f32" on the gather-once structure (cnrma_sparse_conv_go_f32: raw fp32 union rows in LDS,
    v_mfma_f32_32x32x2_f32, fragment-order fp32 weights) against the fp64 oracle (2e-6, the stage kernel's bound) and against the
    fp32 stage kernel (same products, another summation order), with residual + ReLU, forced splits over channel slices, on a
    point set without locality (several offset groups per tile) and with a ragged last tile; run to run bit-identical"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + n + 1)
    c, f = rand_sparse(rng, n=n, span=span, C=cin, ts=1)
    W = torch.from_numpy((rng.randn(27, cin, cout) / np.sqrt(cin * 27)).astype(np.float32)).to(device)
    res = torch.from_numpy(rng.randn(len(c), cout).astype(np.float32)).to(device)
    x = to_st(c, f, 1, device)
    seen = []
    orig_call = S.call

    def call(name, *a):
        seen.append(name)
        return orig_call(name, *a)
    prev = S.GO_CONV, S.GO_F32
    try:
        S.GO_CONV, S.GO_F32, S.call = True, False, call
        ref = S.conv(x, W, 3, 1, residual=res, act="relu", precision="f32").F.clone()
        assert "cnrma_sparse_conv_f32" in seen and "cnrma_sparse_conv_go_f32" not in seen
        S.GO_F32 = True
        for splits in (-1, 1, 2):
            S.conv_tuning(splits=splits)
            del seen[:]
            got = S.conv(x, W, 3, 1, residual=res, act="relu", precision="f32").F
            assert "cnrma_sparse_conv_go_f32" in seen and "cnrma_sparse_conv_f32" not in seen
            assert torch.allclose(got, ref, rtol=1e-5, atol=2e-6 * float(ref.abs().max())), splits
            assert torch.equal(got, S.conv(x, W, 3, 1, residual=res, act="relu", precision="f32").F)
            plain = S.conv(x, W, 3, 1, precision="f32")
            oc, of = SO.conv(c, f, W.cpu().numpy(), 3, 1, 1)
            check(S.SparseTensor(plain.F, x.cs), oc, of, tol=2e-6)
    finally:
        S.conv_tuning()
        S.GO_CONV, S.GO_F32 = prev
        S.call = orig_call


def test_presplit_companions_give_identical_results(device):
    """bf16x6 with pre-split feature companions (read + written by the conv epilogue) == bf16x6 splitting in the loop"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(11)
    c, f = rand_sparse(rng, n=5000, span=10, C=64)
    W1 = (rng.randn(27, 64, 128) / 40).astype(np.float32)
    W2 = (rng.randn(27, 128, 64) / 60).astype(np.float32)
    Wt = (rng.randn(8, 64, 32) / 20).astype(np.float32)
    outs = []
    for flag in (False, True):
        S.PRESPLIT = flag
        try:
            x = to_st(c, f, 2, device)
            y = S.conv(x, torch.from_numpy(W1).to(device), 3, 1, act="relu", precision="bf16x6")
            z = S.conv(y, torch.from_numpy(W2).to(device), 3, 1, precision="bf16x6")
            u = S.conv_transpose_generative(z, torch.from_numpy(Wt).to(device), act="elu", precision="bf16x6")
            assert (y._split is not None) == flag and (u._split is not None) == flag
            outs.append((y.F.clone(), z.F.clone(), u.F.clone()))
        finally:
            S.PRESPLIT = False
    for a, b in zip(*outs):
        assert torch.equal(a, b)            # the truncation split is exact and deterministic: bit-identical paths
    # the companion written by the epilogue reconstructs the fp32 features exactly: h + m + l == x
    S.PRESPLIT = True
    try:
        y = S.conv(to_st(c, f, 2, device), torch.from_numpy(W1).to(device), 3, 1, act="relu", precision="bf16x6")
    finally:
        S.PRESPLIT = False
    sp = y._split[:-1].float()                       # [n, C/8, 3, 8]
    rec = (sp[:, :, 0] + sp[:, :, 1] + sp[:, :, 2]).reshape(y.F.shape)
    assert torch.equal(rec, y.F) and float(y._split[-1].float().abs().max()) == 0.0


def test_bf16x6_is_fp32_grade_on_wide_dynamic_range(device):
    """the 3-way bf16 split must hold fp32 accuracy for operands spanning many binades (bf16 keeps fp32's exponent)"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(9)
    c, f = rand_sparse(rng, n=6000, span=9, C=64)
    f = (f * np.exp(rng.uniform(-12, 12, size=f.shape))).astype(np.float32)
    W = (rng.randn(27, 64, 64) * np.exp(rng.uniform(-8, 8, size=(27, 64, 64)))).astype(np.float32)
    x = to_st(c, f, 1, device)
    y6 = S.conv(x, torch.from_numpy(W).to(device), 3, 1, precision="bf16x6").F.cpu().numpy().astype(np.float64)
    y32 = S.conv(x, torch.from_numpy(W).to(device), 3, 1, precision="f32").F.cpu().numpy().astype(np.float64)
    y16 = S.conv(x, torch.from_numpy(W).to(device), 3, 1, precision="f16x3").F.cpu().numpy().astype(np.float64)
    _, ref = SO.conv(c, f, W, 3, 1, 1)
    # error relative to sum |a||b| (the natural scale of a dot product's rounding error)
    look = SO.Lookup(c)
    mag = np.zeros_like(ref)
    for k, off in enumerate(SO.kernel_offsets(3, 1)):
        q = c.copy(); q[:, 1:] += off
        idx = look(q); m = idx >= 0
        mag[m] += np.abs(f[idx[m]].astype(np.float64)) @ np.abs(W[k].astype(np.float64))
    e6 = np.abs(y6 - ref).max() / mag.max()
    e32 = np.abs(y32 - ref).max() / mag.max()
    e16 = np.abs(y16 - ref).max() / mag.max()
    print("max error / max sum|a||b|: bf16x6 %.2e  f32 %.2e  f16x3 %.2e" % (e6, e32, e16))
    print("max over outputs of error / sum|a||b|: bf16x6 %.2e  f16x3 %.2e" % (
        (np.abs(y6 - ref) / (mag + 1e-30)).max(), (np.abs(y16 - ref) / (mag + 1e-30)).max()))
    assert e6 < 4e-7 and e32 < 4e-7, (e6, e32)
    assert (np.abs(y6 - ref) / (mag + 1e-30)).max() < 2e-6
    # f16x3: 22-bit operands under ONE power-of-two scale per tensor: full accuracy relative to the tensor's magnitude
    assert e16 < 8e-7, e16


@pytest.mark.parametrize("ts", [1, 4])
def test_structured_kernel_maps_equal_generic(device, ts):
    """the symmetric (stride-1) and input-driven (stride-2 conv k3/k1, pool k2) builders give the generic table"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(ts)
    c, _ = rand_sparse(rng, n=20000, span=16, C=1, ts=ts, batch=2)
    cs = S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), ts, n_batch=2)
    child = cs.strided(2)
    for ksize, out in ((3, cs), (3, child), (2, child), (1, child)):
        fast = cs.neighbours(out, ksize, ts, method="auto").clone()
        cs._nbr.clear()
        ref = cs.neighbours(out, ksize, ts, method="generic").clone()
        cs._nbr.clear()
        assert torch.equal(fast, ref), (ksize, out is cs)
        assert (ref >= 0).any()


def test_conv_fused_epilogue(device):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(5)
    c, f = rand_sparse(rng, n=3000, span=10, C=64)
    W = (rng.randn(27, 64, 64) / 40).astype(np.float32)
    scale, shift = rng.rand(64).astype(np.float32) + 0.5, rng.randn(64).astype(np.float32)
    res = rng.randn(len(c), 64).astype(np.float32)
    x = to_st(c, f, 1, device)
    for act, fn in (("relu", SO.relu), ("elu", SO.elu), (None, lambda v: v)):
        out = S.conv(x, torch.from_numpy(W).to(device), 3, 1, torch.from_numpy(scale).to(device),
                     torch.from_numpy(shift).to(device), torch.from_numpy(res).to(device), act)
        _, o = SO.conv(c, f, W, 3, 1, 1)
        check(out, c, fn(o * scale + shift + res))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("cin,cout,ts,n,span", [(32, 64, 1, 6000, 40), (256, 64, 1, 3000, 30), (64, 128, 2, 5000, 25), (32, 64, 1, 1, 4)])
def test_pair_list_conv_vs_oracle_and_tile_kernel(device, cin, cout, ts, n, span, precision):
    """the stem's kernel (stride 2 on a sparse point sample -> nearly empty kernel map): pair-list path vs the fp64 oracle,
    vs the output-stationary tile kernel, with the fused epilogue, and bit-reproducible from run to run (the pair slots are
    handed out by atomics: the list layout varies, the sums must not) -- in f16x3 and in exact fp32 (cnrma_sparse_conv_pairs_f32)"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + n)
    c, f = rand_sparse(rng, n=n, span=span, C=cin, ts=ts, batch=2)
    W = (rng.randn(27, cin, cout) / np.sqrt(cin * 4)).astype(np.float32)
    scale, shift = rng.rand(cout).astype(np.float32) + 0.5, rng.randn(cout).astype(np.float32)
    oc, of = SO.conv(c, f, W, 3, 2, ts)
    Wd, sd, hd = (torch.from_numpy(a).to(device) for a in (W, scale, shift))
    x = to_st(c, f, ts, device)
    assert S._nearly_empty_map(x.cs, x.cs.strided(2)), "test inputs must be sparse enough to take the pair-list path"
    pairs, tile_entry = ("cnrma_sparse_conv_pairs_f16x3", "cnrma_sparse_conv_f16x3") if precision == "f16x3" else \
        ("cnrma_sparse_conv_pairs_f32", "cnrma_sparse_conv_f32")
    calls = []
    orig, prev_min = S.call, S.PAIR_CONV_MIN_CIN
    S.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    S.PAIR_CONV_MIN_CIN = 32
    try:
        out = S.conv(x, Wd, 3, 2, sd, hd, None, "relu", precision=precision)
        again = S.conv(to_st(c, f, ts, device), Wd, 3, 2, sd, hd, None, "relu", precision=precision)
    finally:
        S.call, S.PAIR_CONV_MIN_CIN = orig, prev_min
    assert calls.count(pairs) == 2 and tile_entry not in calls
    check(out, oc, SO.relu(of * scale + shift), tol=2e-6)
    assert torch.equal(out.F, again.F)
    prev = S.PAIR_CONV
    S.PAIR_CONV = False
    try:
        tile = S.conv(to_st(c, f, ts, device), Wd, 3, 2, sd, hd, None, "relu", precision=precision)
    finally:
        S.PAIR_CONV = prev
    assert torch.equal(tile.C, out.C)
    assert float((tile.F - out.F).abs().max()) <= 2e-6 * max(1.0, float(tile.F.abs().max()))
    # the magnitude bound published for the consumers covers the output (f16x3: the next layer's operand scale)
    if precision == "f16x3":
        assert float(out.amax.max()) == float(out.F.abs().max())


def test_generative_transpose_vs_oracle(device):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(6)
    c, f = rand_sparse(rng, n=800, span=6, C=128, ts=8)
    W = (rng.randn(8, 128, 64) / 30).astype(np.float32)
    out = S.conv_transpose_generative(to_st(c, f, 8, device), torch.from_numpy(W).to(device))
    oc, of = SO.conv_transpose_generative(c, f, W, 8)
    check(out, oc, of, same_order=False)          # the reference's row order is implementation-defined (a hash map's)
    assert out.cs.stride == 4
    # ours: parent-major, the 8 children in Morton order (x the most significant bit): row 8 * i + m
    got = out.C.cpu().numpy().astype(np.int64).reshape(len(c), 8, 4)
    m = np.arange(8)
    off = np.stack((np.zeros(8, np.int64), (m >> 2) & 1, (m >> 1) & 1, m & 1), axis=1) * 4
    assert (got == c[:, None, :] + off[None]).all()


def test_maxpool_and_instnorm_vs_oracle(device):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(7)
    c, f = rand_sparse(rng, n=5000, span=14, C=64, ts=2)
    x = to_st(c, f, 2, device)
    oc, of = SO.max_pool(c, f, 2)
    check(S.max_pool(x), oc, of, tol=0)
    w, b = rng.rand(1, 64).astype(np.float32) + 0.5, rng.randn(1, 64).astype(np.float32)
    y = S.instance_norm(x, torch.from_numpy(w).to(device), torch.from_numpy(b).to(device), relu=True)
    check(y, c, SO.relu(SO.instance_norm(f, w, b)), tol=1e-5)


def test_union_interp_prune_vs_oracle(device):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(8)
    ca, fa = rand_sparse(rng, n=3000, span=10, C=16, ts=4)
    cb, fb = rand_sparse(rng, n=3000, span=10, C=16, ts=4)
    a, b = to_st(ca, fa, 4, device), to_st(cb, fb, 4, device)
    u = a + b
    uc, uf = SO.union_add(ca, fa, cb, fb)
    check(u, uc, uf, tol=1e-6)
    # the union carries a valid coordinate map: a k3 conv over it must match the oracle
    W = (rng.randn(27, 16, 8) / 20).astype(np.float32)
    _, of = SO.conv(uc, uf, W, 3, 1, 4)
    check(S.conv(u, torch.from_numpy(W).to(device), 3, 1), uc, of)
    # interpolation of a coarse score at the finer (half-stride) lattice
    cs, fs = rand_sparse(rng, n=2000, span=5, C=1, ts=8)
    cq, _ = rand_sparse(rng, n=4000, span=10, C=1, ts=4)
    got = S.interpolate(to_st(cs, fs, 8, device), torch.from_numpy(cq.astype(np.int32)).to(device)).cpu().numpy()
    np.testing.assert_allclose(got, SO.interpolate(cs, fs, 8, cq), atol=1e-6)
    # pruning keeps the masked rows in order
    mask = rng.rand(len(ca)) < 0.4
    p = S.prune(a, torch.from_numpy(mask).to(device))
    check(p, ca[mask], fa[mask], tol=0)


def test_decode_and_scores_vs_reference_golden(device):
    from cnrma_amd import sparse as S
    z = np.load(os.path.join(GOLDEN, "decode.npz"))
    pts = t(z["points"], device)
    for nreg, yaw in ((6, "fcaf3d"), (8, "fcaf3d"), (8, "sin-cos"), (7, "naive")):
        box = S.decode_boxes(pts, t(z[f"pred_{nreg}_{yaw}"], device), yaw).cpu().numpy()
        np.testing.assert_allclose(box, z[f"box_{nreg}_{yaw}"], rtol=1e-5, atol=1e-5)
    cls, ctr = torch.randn(500, 18, device=device), torch.randn(500, 1, device=device)
    s, mx = S.class_scores(cls, ctr)
    exp = torch.sigmoid(cls) * torch.sigmoid(ctr)
    assert torch.allclose(s, exp, atol=1e-6) and torch.allclose(mx, exp.max(dim=1)[0], atol=1e-6)


def test_fused_head_post_and_select_decode(device):
    from cnrma_amd import sparse as S
    z = np.load(os.path.join(GOLDEN, "decode.npz"))
    g = torch.Generator().manual_seed(5)
    n, R, nc = 1000, 8, 17
    y = torch.randn(n, 28, generator=g).to(device)
    coords = torch.randint(-300, 300, (n, 4), generator=g, dtype=torch.int32).to(device)
    scale = torch.tensor(1.37, device=device)
    cen, box, cls, mx, pts = S.head_post(y, coords, R, nc, scale, 0.01)
    assert torch.equal(cen, y[:, :1]) and torch.equal(cls, y[:, 1 + R:1 + R + nc])
    assert torch.allclose(box[:, :6], torch.exp(y[:, 1:7] * scale), rtol=1e-6) and torch.equal(box[:, 6:], y[:, 7:9])
    assert torch.equal(mx[:, 0], cls.max(dim=1)[0]) and torch.equal(pts, coords[:, 1:] * 0.01)
    # select + decode against the reference's golden decode vectors, through a permutation
    for nreg, yaw in ((6, "fcaf3d"), (8, "fcaf3d"), (8, "sin-cos"), (7, "naive")):
        pred = t(z[f"pred_{nreg}_{yaw}"], device)
        m = pred.shape[0]
        ids = torch.randperm(m, generator=g)[:100].to(device)
        c2, ct = torch.randn(m, nc, generator=g).to(device), torch.randn(m, 1, generator=g).to(device)
        boxes, scores = S.select_decode(ids, c2, ct, pred, t(z["points"], device), yaw)
        np.testing.assert_allclose(boxes.cpu().numpy(), z[f"box_{nreg}_{yaw}"][ids.cpu().numpy()], rtol=1e-5, atol=1e-5)
        assert torch.allclose(scores, (torch.sigmoid(c2) * torch.sigmoid(ct))[ids], atol=1e-6)
        assert torch.allclose(S.max_scores(c2, ct), (torch.sigmoid(c2) * torch.sigmoid(ct)).max(dim=1)[0], atol=1e-6)


def test_topk_mask_matches_torch_topk(device):
    from cnrma_amd import sparse as S
    g = torch.Generator().manual_seed(1)
    x = torch.randn(270_336, generator=g).to(device)
    x[::7] = x[3]                                           # many exact ties
    x[5] = float("-inf"); x[6] = 0.0; x[8] = -0.0
    for k in (1, 1000, 200_000, 270_335, 270_336, 300_000):
        m = S.topk_mask(x, k).bool()
        assert int(m.sum()) == min(k, x.numel())
        if k < x.numel():
            kth = torch.topk(x, k).values[-1]
            assert bool((x[m] >= kth).all()) and bool((x[~m] <= kth).all())
            # ties on the threshold value are resolved towards the smaller index
            tie = torch.nonzero(x == kth).squeeze(1)
            kept = m[tie]
            assert bool((kept[:-1] >= kept[1:]).all()) if len(tie) > 1 else True


def _randomise(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() >= 2 and "kernel" in n:
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / np.sqrt(p.shape[-2] * (p.shape[0] if p.dim() == 3 else 1))))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2 + (1.0 if "weight" in n or "scale" in n else 0.0))
        for n, b in module.named_buffers():
            if "running_mean" in n:
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            if "running_var" in n:
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)


@pytest.mark.parametrize("n_cls,n_reg,yaw", [(18, 6, "fcaf3d"), (17, 8, "fcaf3d"), (18, 8, "sin-cos")])
def test_fcaf3d_forward_vs_oracle(device, n_cls, n_reg, yaw):
    """whole backbone + neck + head + decode on a small point cloud: every level's coordinate set bit-exact,
    head outputs and decoded boxes within 1e-4 of the float64 oracle."""
    from cnrma_amd import sparse as S
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    from projects.mvsdetection.models.fcaf3d_head import FCAF3DHead
    rng = np.random.RandomState(3)
    # points on a few planes (room-like), 1-cm voxels, ~2 m extent, some negative coordinates
    pts = rng.rand(30000, 3).astype(np.float32) * np.array([2.0, 1.6, 1.2], dtype=np.float32) - 0.3
    pts[:10000, 2] = -0.3 + 0.01 * rng.rand(10000)
    pts[10000:20000, 0] = 1.7 - 0.01 * rng.rand(10000)
    feats = rng.randn(30000, 32).astype(np.float32)
    backbone = FCAF3DBackbone(32, 34)
    head = FCAF3DHead(n_classes=n_cls, in_channels=(64, 128, 256, 512), out_channels=128, n_reg_outs=n_reg, voxel_size=0.01,
                      pts_threshold=1500, assigner=None, yaw_parametrization=yaw, test_cfg=dict(nms_pre=300, iou_thr=.5, score_thr=.01))
    _randomise(backbone, 1)
    _randomise(head, 2)
    torch.manual_seed(7)
    head.init_weights()            # head convs ~ N(0, .01) like the reference (:100-104): keeps exp(reg) finite
    with torch.no_grad():
        for sc in head.scales:
            sc.scale.fill_(1.3)
    backbone.eval(); head.eval()
    Cq, Fq, _ = RO.voxelize(torch.from_numpy(pts), torch.from_numpy(feats), 0.01)
    levels = SO.backbone_forward(backbone, Cq.numpy(), Fq.numpy())
    exp = SO.head_forward(head, levels)
    backbone.to(device); head.to(device)
    with torch.no_grad():
        x, _ = S.voxelize(torch.from_numpy(pts).to(device), torch.from_numpy(feats).to(device), 0.01)
        outs = backbone(x)
        for o, (c, f, ts) in zip(outs, levels):
            check(o, c, f, tol=2e-4, same_order=False)      # Morton row order on the device, first-occurrence in the oracle
            c1, f1 = sort_rows(o.C.cpu().numpy().astype(np.int64), o.F.cpu().numpy())
            # ... and every element within 1e-3 absolutely or relatively (achieved 6.0e-4 -- 4.0e-4 with the stage kernel alone: |f| ~ 2e2 with these random weights;
            # head outputs / boxes below hold 1e-4, as do the features of the benchmark's model at full size)
            err = elementwise_error(f1, sort_rows(c, f)[1])
            assert err <= 1e-3, err
            assert o.cs.stride == ts
        cen, box, cls, points = map(list, head(outs))
    for i in range(4):
        e = exp[i]
        got_pts = points[i][0].cpu().numpy()
        # rows of a pruned level may be any top-k subset under ties; compare through the coordinate key
        ck = SO._key(np.concatenate((np.zeros((len(got_pts), 1)), np.round(got_pts / 0.01)), axis=1).astype(np.int64))
        ek = SO._key(e["coords"])
        assert len(ck) == len(ek)
        common = np.intersect1d(ck, ek)
        assert len(common) >= 0.99 * len(ek)
        gi = np.argsort(ck)[np.searchsorted(np.sort(ck), common)]
        ei = np.argsort(ek)[np.searchsorted(np.sort(ek), common)]
        for got, key in ((cen[i][0], "centerness"), (box[i][0], "bbox_pred"), (cls[i][0], "cls_score")):
            g_, e_ = got.cpu().numpy()[gi].astype(np.float64), e[key][ei]
            if key == "bbox_pred":      # exp(scale * reg): compare the exponent (random weights can overflow fp32)
                with np.errstate(over="ignore", divide="ignore"):
                    g_ = np.concatenate((np.log(g_[:, :6]), g_[:, 6:]), axis=1)
                    e_ = np.concatenate((np.log(e_[:, :6]), e_[:, 6:]), axis=1)
                fin = np.isfinite(e_) & (np.abs(e_) < 80)
                g_, e_ = g_[fin], e_[fin]
            # north star: box regressions / head outputs within 1e-4 of the fp64 oracle, element-wise (absolute or relative)
            assert elementwise_error(g_, e_) <= TOL, (i, key, elementwise_error(g_, e_))
    # decode: compare the boxes of level 3 (no top-k ambiguity at this size)
    b_got = head._bbox_pred_to_bbox(points[3][0], box[3][0]).cpu().numpy()
    b_exp = SO.decode_boxes(torch.from_numpy(exp[3]["points"]), torch.from_numpy(exp[3]["bbox_pred"]), yaw).numpy()
    o1 = np.lexsort(np.round(points[3][0].cpu().numpy() / 0.01).T)
    o2 = np.lexsort(np.round(exp[3]["points"] / 0.01).T)
    ok = np.isfinite(b_exp[o2]).all(axis=1) & (np.abs(b_exp[o2]).max(axis=1) < 1e4)
    assert ok.sum() > 0
    assert elementwise_error(b_got[o1][ok], b_exp[o2][ok]) <= TOL, elementwise_error(b_got[o1][ok], b_exp[o2][ok])   # boxes at 1e-4


def _amax(t):
    return float(t.amax.max()) if t.amax is not None else None


def test_f16x3_magnitude_bounds_travel_with_the_tensors(device):
    """every producer hands its consumers a bound >= max|F| (conv epilogue, split-K reduce, generative transpose,
    pooling / pruning pass-through, absmax pass for the rest), and the bound is tight (it IS the maximum)"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(21)
    c, f = rand_sparse(rng, n=6000, span=12, C=64)
    x = to_st(c, f, 2, device)
    assert x.amax is None
    W1 = torch.from_numpy((rng.randn(27, 64, 128) / 30).astype(np.float32)).to(device)
    W2 = torch.from_numpy((rng.randn(27, 128, 64) / 40).astype(np.float32)).to(device)
    Wt = torch.from_numpy((rng.randn(8, 64, 32) / 8).astype(np.float32)).to(device)
    y = S.conv(x, W1, 3, 1, act="relu", precision="f16x3")               # 6000 rows: split over kernel offsets
    assert abs(float(x.amax.max()) - float(x.F.abs().max())) == 0.0       # absmax pass on first use
    z = S.conv(y, W2, 3, 2, precision="f16x3")
    u = S.conv_transpose_generative(z, Wt, act="elu", precision="f16x3")
    for t in (y, z, u):
        assert _amax(t) == float(t.F.abs().max()), (_amax(t), float(t.F.abs().max()))
    p = S.max_pool(y, 2, 2)
    q = S.prune(y, torch.arange(y.cs.n, device=device) % 3 == 0)
    assert p.amax is y.amax and q.amax is y.amax
    # a chain whose magnitude grows by ~1e3 per layer stays accurate: the scales follow the bounds
    big = torch.from_numpy((rng.randn(27, 64, 64) * 40).astype(np.float32)).to(device)
    t16 = t32 = to_st(c, f, 2, device)
    for _ in range(4):
        t16 = S.conv(t16, big, 3, 1, precision="f16x3")
        t32 = S.conv(t32, big, 3, 1, precision="f32")
    rel = float((t16.F - t32.F).abs().max() / t32.F.abs().max())
    assert float(t32.F.abs().max()) > 1e9 and rel < 5e-6, rel


@pytest.mark.parametrize("fs,ws", [(1e-20, 1e15), (1e20, 1e-15), (3e-30, 1.0), (1.0, 2e25)])
def test_f16x3_scales_are_exact_powers_of_two(device, fs, ws):
    """operands far outside the fp16 range: the per-tensor scales are powers of two, so the result is the unscaled
    result times fs*ws up to fp32 rounding of that factor"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(5)
    c, f = rand_sparse(rng, n=3000, span=10, C=32)
    W = (rng.randn(27, 32, 64) / 20).astype(np.float32)
    ref = S.conv(to_st(c, f, 1, device), torch.from_numpy(W).to(device), 3, 1, precision="f32").F.double()
    got = S.conv(to_st(c, (f * np.float32(fs)).astype(np.float32), 1, device),
                 torch.from_numpy((W * np.float32(ws)).astype(np.float32)).to(device), 3, 1, precision="f16x3").F.double()
    k = float(np.float32(fs)) * float(np.float32(ws))
    err = float((got / k - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, err


def test_f16x3_all_zero_and_empty_inputs(device):
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(2)
    c, f = rand_sparse(rng, n=500, span=6, C=32)
    W = torch.from_numpy(rng.randn(27, 32, 64).astype(np.float32)).to(device)
    y = S.conv(to_st(c, np.zeros_like(f), 1, device), W, 3, 1, precision="f16x3")
    assert float(y.F.abs().max()) == 0.0 and _amax(y) == 0.0
    shift = torch.full((64,), 0.5, device=device)
    y = S.conv(to_st(c, np.zeros_like(f), 1, device), W, 3, 1, shift=shift, precision="f16x3")
    assert float((y.F - 0.5).abs().max()) == 0.0 and _amax(y) == 0.5


@pytest.mark.parametrize("n,k,live", [(5000, 100, None), (300000, 1000, None), (2000, 500, 320), (64, 64, None),
                                      (50000, 2000, None), (300000, 1024, None), (900, 1024, None)])
def test_topk_indices_match_torch_topk(device, n, k, live):
    """decode's nms_pre cut: same rows, same order as torch.topk (ties -> smaller index); with fewer live rows than k the
    live rows come first in score order"""
    from cnrma_amd import sparse as S
    g = torch.Generator().manual_seed(n + k)
    s = torch.rand(n, generator=g).to(device)
    s[::7] = s[3]                                           # ties
    n_dev = None if live is None else torch.tensor([live], dtype=torch.int32, device=device)
    ids = S.topk_indices(s, k, n_dev)
    assert ids.dtype == torch.int64 and ids.shape == (k,)
    m = n if live is None else live
    ref = torch.sort(s[:m], descending=True, stable=True)[1][:k]
    assert torch.equal(ids[:min(k, m)], ref[:min(k, m)])
    assert bool((ids[min(k, m):] == 0).all())


def test_bottleneck_backbone_vs_oracle(device):
    """FCAF3DBackbone(depth=50): ME's Bottleneck blocks (1x1 - 3x3 strided - 1x1 x4, fcaf3d_backbone.py:122-124) against the fp64
    oracle: coordinate sets bit-exact, every feature within 1e-5 of the level's largest (abs or rel; achieved 2e-6 -- 50
    convolutions with random weights drive |f| to ~1e3, so the bound is taken on the features scaled to unit maximum)"""
    from cnrma_amd import sparse as S
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    rng = np.random.RandomState(12)
    pts = rng.rand(6000, 3).astype(np.float32) * np.array([0.9, 0.7, 0.5], dtype=np.float32) - 0.1
    pts[:2000, 2] = -0.1 + 0.01 * rng.rand(2000)
    feats = rng.randn(6000, 16).astype(np.float32)
    backbone = FCAF3DBackbone(16, 50)
    _randomise(backbone, 3)
    backbone.eval()
    Cq, Fq, _ = RO.voxelize(torch.from_numpy(pts), torch.from_numpy(feats), 0.01)
    levels = SO.backbone_forward(backbone, Cq.numpy(), Fq.numpy())
    backbone.to(device)
    with torch.no_grad():
        x, _ = S.voxelize(torch.from_numpy(pts).to(device), torch.from_numpy(feats).to(device), 0.01)
        outs = backbone(x)
    assert [o.F.shape[1] for o in outs] == [256, 512, 1024, 2048]
    for o, (c, f, ts) in zip(outs, levels):
        check(o, c, f, tol=2e-4, same_order=False)
        c1, f1 = sort_rows(o.C.cpu().numpy().astype(np.int64), o.F.cpu().numpy())
        f2 = sort_rows(c, f)[1]
        m = float(np.abs(f2).max())
        assert elementwise_error(f1 / m, f2 / m) <= 1e-5
        assert o.cs.stride == ts


def test_topk_indices_under_heavy_ties(device):
    """thousands of equal scores across the k-th place: the fused select + sort keeps the smallest rows of the tie"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(5)
    s = rng.rand(70000).astype(np.float32)
    s[rng.permutation(70000)[:3000]] = 0.75
    k = min(1000, int((s > 0.75).sum()) + 500)
    ids = S.topk_indices(torch.from_numpy(s).to(device), k).cpu().numpy()
    assert (ids == np.argsort(-s, kind="stable")[:k]).all()


def test_topk_mask_is_exact_under_heavy_ties(device):
    """thousands of exactly equal scores at the threshold: exactly k rows, the tied ones by smallest index
    (the row set of a stable descending sort)"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(4)
    for n, k, n_zero in ((50000, 20000, 40000), (300000, 200000, 150000), (5000, 4999, 5000), (70000, 300, 0)):
        s = rng.rand(n).astype(np.float32)
        s[rng.permutation(n)[:n_zero]] = 0.0                     # a big block of ties (interpolated scores are often 0)
        if n_zero == 0:
            s[rng.permutation(n)[:3000]] = 0.75                  # ties inside the kept range, cut through the middle
            k = int((s > 0.75).sum()) + 1500
        m = S.topk_mask(torch.from_numpy(s).to(device), k).cpu().numpy().astype(bool)
        order = np.argsort(-s, kind="stable")[:k]
        exp = np.zeros(n, dtype=bool); exp[order] = True
        assert m.sum() == k and (m == exp).all()


@pytest.mark.parametrize("cin,cout,k,stride,ts", [(32, 64, 3, 1, 1), (64, 32, 3, 2, 2), (64, 128, 1, 1, 1), (64, 64, 1, 2, 1),
                                                   (5, 7, 3, 1, 1)])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_conv_backward_vs_oracle(device, cin, cout, k, stride, ts, precision):
    """dgrad (forward kernel on the transposed neighbour table) and wgrad (fp32 MFMA over row chunks) of the sparse
    convolution against the fp64 restatement"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + cout + k + stride)
    c, f = rand_sparse(rng, n=5000, span=12, C=cin, ts=ts, batch=2)
    W = (rng.randn(*((k ** 3, cin, cout) if k > 1 else (cin, cout))) / np.sqrt(cin * k ** 3)).astype(np.float32)
    x = to_st(c, f, ts, device)
    x.F.requires_grad_(True)
    Wt = torch.from_numpy(W).to(device).requires_grad_(True)
    y = S.conv_autograd(x, Wt, k, stride, precision=precision)
    oc, of = SO.conv(c, f, W, k, stride, ts)
    G = rng.randn(*of.shape).astype(np.float32)
    assert (y.cs.C.cpu().numpy() == oc).all()                 # same rows in the same (first-occurrence) order
    np.testing.assert_allclose(y.F.detach().cpu().numpy(), of, rtol=2e-6, atol=2e-6 * max(1.0, float(np.abs(of).max())))
    (y.F * torch.from_numpy(G).to(device)).sum().backward()
    gF, gW = SO.conv_backward(c, f, W, G, k, stride, ts)
    tol = 2e-6
    assert np.abs(x.F.grad.cpu().numpy() - gF).max() / np.abs(gF).max() < tol
    assert np.abs(Wt.grad.cpu().numpy() - gW).max() / np.abs(gW).max() < tol


def test_training_paths_equal_the_inference_kernels(device):
    """the differentiable (torch-composed) paths of the non-conv sparse ops give the inference kernels' results"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(31)
    c, f = rand_sparse(rng, n=4000, span=10, C=32, ts=2, batch=2)
    order = np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))          # scene-major rows (instance norm segments)
    c, f = c[order], f[order]

    def both(fn):
        x0 = to_st(c, f, 2, device); x0.cs.n_batch = 2; x0.cs.scene_major = True
        x1 = to_st(c, f, 2, device); x1.cs.n_batch = 2; x1.cs.scene_major = True
        x1.F.requires_grad_(True)
        with torch.no_grad():
            y0 = fn(x0)
        y1 = fn(x1)
        assert y1.F.requires_grad and (y0.C.cpu() == y1.C.cpu()).all()
        np.testing.assert_allclose(y1.F.detach().cpu().numpy(), y0.F.cpu().numpy(), rtol=2e-5, atol=2e-5)
        y1.F.sum().backward()
        assert torch.isfinite(x1.F.grad).all()
        return y0

    w = torch.from_numpy(rng.randn(32).astype(np.float32)).to(device)
    b = torch.from_numpy(rng.randn(32).astype(np.float32)).to(device)
    Wt = torch.from_numpy((rng.randn(8, 32, 16) / 6).astype(np.float32)).to(device)
    both(lambda x: S.max_pool(x, 2, 2))
    both(lambda x: S.instance_norm(x, w, b, 1e-8, relu=True))
    both(lambda x: S.conv_transpose_generative(x, Wt, act="elu"))
    both(lambda x: S.prune(x, torch.arange(x.cs.n, device=device) % 3 != 1))
    c2, f2 = rand_sparse(rng, n=3000, span=10, C=32, ts=2, batch=2)
    other = to_st(c2, f2, 2, device); other.cs.n_batch = 2
    both(lambda x: S.union_add(x, other))


def test_fcaf3d_trains_one_step(device):
    """backbone + head in training mode: forward through the differentiable sparse ops (dgrad / wgrad kernels for the
    convolutions), backward, every parameter receives a finite gradient and SGD lowers the loss on the same batch"""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    from projects.mvsdetection.models.fcaf3d_head import FCAF3DHead
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(7)
    torch.manual_seed(0)
    pts = torch.from_numpy((rng.rand(6000, 3) * np.array([2.0, 1.6, 1.0])).astype(np.float32)).to(device)
    feats = torch.from_numpy(rng.randn(6000, 32).astype(np.float32)).to(device)
    backbone = FCAF3DBackbone(in_channels=32, depth=14).to(device)
    head = FCAF3DHead(n_classes=4, in_channels=(64, 128, 256, 512), out_channels=128, n_reg_outs=6, voxel_size=0.01,
                      pts_threshold=100000, assigner=None).to(device)
    backbone.init_weights(); head.init_weights()
    backbone.train(); head.train()
    params = [p for p in list(backbone.parameters()) + list(head.parameters())]
    opt = torch.optim.SGD(params, lr=1e-3)

    def loss_fn():
        x, _ = S.voxelize(pts, feats, 0.01)
        x.F.requires_grad_(True)
        cen, box, cls, _ = map(list, head(backbone(x)))
        t = sum((c_[0] ** 2).mean() for c_ in cen) + sum((b_[0][:, :6].log() ** 2).mean() for b_ in box) \
            + sum((k_[0] ** 2).mean() for k_ in cls)
        return t, x
    l0, x = loss_fn()
    opt.zero_grad()
    l0.backward()
    assert torch.isfinite(x.F.grad).all() and float(x.F.grad.abs().sum()) > 0
    missing = [n for n, p in list(backbone.named_parameters()) + list(head.named_parameters()) if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in params)
    opt.step()
    l1, _ = loss_fn()
    assert float(l1.detach()) < float(l0.detach()), (float(l0.detach()), float(l1.detach()))


def test_voxelize_reports_out_of_range_coordinates(device):
    """coordinates beyond the 16-bit fields of the voxel key (or NaN) must not alias other voxels silently"""
    from cnrma_amd import _lib
    from cnrma_amd import sparse as S
    pts = torch.rand(1000, 3, device=device)
    f = torch.randn(1000, 8, device=device)
    S.voxelize(pts, f, 0.01)                                   # fine
    for bad in (400.0, float("nan"), -1e9):
        p = pts.clone()
        p[17, 1] = bad
        with pytest.raises(_lib.CnrmaError):
            S.voxelize(p, f, 0.01)


def test_generated_children_table_from_the_parents_table(device):
    """the 3x3x3 table of a generated child set derived from the parents' table (cnrma_sparse_kernel_map_children) equals the
    table the hash-map builder produces for the same rows, entry for entry"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(21)
    c, f = rand_sparse(rng, n=3000, span=9, C=32, ts=4)
    c[:, 1:] -= 16                                              # negative coordinates as well
    W = torch.from_numpy((rng.randn(8, 32, 32) / 8).astype(np.float32)).to(device)
    y = S.conv_transpose_generative(to_st(c, f, 4, device), W)
    assert y.cs._gen_parent is not None and y.cs.stride == 2
    got = y.cs.neighbours(y.cs, 3, 2)
    plain = S.CoordSet(y.C.clone(), 2)
    exp = plain.neighbours(plain, 3, 2)
    assert torch.equal(got, exp)
    # and a convolution over it matches the oracle
    W3 = (rng.randn(27, 32, 64) / 30).astype(np.float32)
    oc, of = SO.conv_transpose_generative(c, f, W.cpu().numpy(), 4)
    _, o3 = SO.conv(oc, of, W3, 3, 1, 2)
    check(S.conv(y, torch.from_numpy(W3).to(device), 3, 1), oc, o3, tol=2e-6, same_order=False)


def test_strided_sets_of_sorted_rows_need_no_hash_table(device):
    """strided coordinate sets of a Morton-sorted set by adjacent comparison (cnrma_sparse_stride_coords_sorted) == the hash
    route, row for row, over a chain of levels; the kernel maps built on them (lazy coordinate map) match as well"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(31)
    pts = (rng.rand(40000, 3).astype(np.float32) - 0.4) * np.array([3.0, 2.0, 1.0], dtype=np.float32)      # negative coordinates too
    x, _ = S.voxelize(torch.from_numpy(pts).to(device), torch.zeros(40000, 4, device=device), 0.02)
    assert x.cs.sorted and x.cs.compact
    plain = S.CoordSet(x.C.clone(), 1)                                     # same rows, not flagged: the hash route
    assert not plain.sorted
    a, b = x.cs, plain
    for level in range(4):
        ca, cb = a.strided(2), b.strided(2)
        assert ca.sorted and ca._map is None and ca.n == cb.n and torch.equal(ca.C, cb.C)
        assert torch.equal(a.neighbours(ca, 3, a.stride), b.neighbours(cb, 3, b.stride))       # stride-2 map: probes ca's lazy map
        assert torch.equal(ca.neighbours(ca, 3, ca.stride), cb.neighbours(cb, 3, cb.stride))
        a, b = ca, cb
    assert 0 < a.n < 2000


def test_instance_norm_relu_max_pool_fused_is_bit_identical(device):
    """the stem's InstanceNorm - ReLU - MaxPool as statistics pass + one pooling pass that normalises its candidates on the fly
    (cnrma_sparse_instnorm_maxpool_f32) == the two operators one after the other, bit for bit; the magnitude bound is exact"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(41)
    c, f = rand_sparse(rng, n=20000, span=20, C=64, ts=2)
    w = torch.from_numpy((rng.rand(1, 64) + 0.5).astype(np.float32)).to(device)
    b = torch.from_numpy((rng.randn(1, 64) * 0.3).astype(np.float32)).to(device)
    x = to_st(c, f * 3.0 + 0.7, 2, device)
    ref = S.max_pool(S.instance_norm(x, w, b, 1e-8, relu=True), 2, 2)
    got = S.instance_norm_max_pool(x, w, b, 1e-8, relu=True, kernel_size=2, stride=2)
    assert got.cs.stride == 4 and torch.equal(got.C, ref.C) and torch.equal(got.F, ref.F)
    assert float(got.amax.max()) == float(got.F.abs().max())
    oc, of = SO.max_pool(c, SO.relu(SO.instance_norm(f * 3.0 + 0.7, w.cpu().numpy(), b.cpu().numpy())), 2)
    check(got, oc, of, tol=1e-5)
