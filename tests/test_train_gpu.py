"""Training-side kernels added for BASELINE configs[4] (bf16 autocast): the bf16 convolution against the fp64 oracle on
bf16-rounded operands, and a detector train_step under torch.autocast(bf16) against the same step in fp32."""
import os
import runpy

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as SO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("stride", [1, 2])
def test_bf16_conv_equals_fp32_accumulation_of_bf16_operands(device, stride):
    """one bf16 piece per operand (round to nearest), products exact in fp32, fp32 accumulation: the result equals the
    fp64 oracle on the ROUNDED operands up to accumulation noise -- and differs from the unrounded result by bf16's 2^-8"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(5 + stride)
    c = np.unique(np.concatenate((np.zeros((9000, 1), dtype=np.int64), rng.randint(-20, 20, size=(9000, 3)) * 2), axis=1), axis=0)
    f = rng.randn(len(c), 64).astype(np.float32)
    W = (rng.randn(27, 64, 96) / 20).astype(np.float32)
    x = S.SparseTensor(torch.from_numpy(f).to(device), S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 2))
    y = S.conv(x, torch.from_numpy(W).to(device), 3, stride, precision="bf16")
    oc, exp = SO.conv(c, _bf16_round(f), _bf16_round(W), 3, stride, 2)
    oc0, exact = SO.conv(c, f, W, 3, stride, 2)
    got_c = y.C.cpu().numpy().astype(np.int64)
    o1, o2 = np.lexsort(got_c.T[::-1]), np.lexsort(oc.T[::-1])
    assert np.array_equal(got_c[o1], oc[o2])
    got = y.F.cpu().numpy()[o1]
    np.testing.assert_allclose(got, exp[o2], rtol=1e-5, atol=1e-5)
    err = np.abs(got - exact[o2]).max() / np.abs(exact).max()
    assert 1e-4 < err < 2e-2                                   # bf16 operands: visible, bounded


@pytest.mark.parametrize("cin,cout,stride", [(64, 96, 1), (32, 64, 2), (40, 24, 1)])
def test_bf16_weight_gradient_equals_fp32_accumulation_of_bf16_operands(device, cin, cout, stride):
    """wgrad of the autocast mode (cnrma_sparse_conv_wgrad_bf16): features and output gradients rounded to bf16, fp32
    accumulation -> equals the fp64 oracle on the ROUNDED operands up to accumulation noise"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + stride)
    c = np.unique(np.concatenate((np.zeros((7000, 1), dtype=np.int64), rng.randint(-16, 16, size=(7000, 3)) * 2), axis=1), axis=0)
    f = rng.randn(len(c), cin).astype(np.float32)
    W = (rng.randn(27, cin, cout) / 20).astype(np.float32)
    x = S.SparseTensor(torch.from_numpy(f).to(device).requires_grad_(True), S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 2))
    Wd = torch.from_numpy(W).to(device).requires_grad_(True)
    y = S.conv_autograd(x, Wd, 3, stride, precision="bf16")
    g = rng.randn(*y.F.shape).astype(np.float32)
    calls = []
    orig = S.call
    S.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        y.F.backward(torch.from_numpy(g).to(device))
    finally:
        S.call = orig
    assert "cnrma_sparse_conv_wgrad_bf16" in calls and "cnrma_sparse_conv_wgrad_f32" not in calls
    # the oracle's output rows in the engine's row order
    oc = y.C.cpu().numpy().astype(np.int64)
    _, gW = SO.conv_backward(c, _bf16_round(f), W, _bf16_round(g), 3, stride, 2, out_coords=oc)
    _, gW_exact = SO.conv_backward(c, f, W, g, 3, stride, 2, out_coords=oc)
    got = Wd.grad.cpu().numpy()
    np.testing.assert_allclose(got, gW, rtol=2e-5, atol=2e-5 * np.abs(gW).max())
    err = np.abs(got - gW_exact).max() / np.abs(gW_exact).max()
    assert 1e-5 < err < 2e-2                                   # bf16 operands: visible, bounded
    # the data gradient: grad_out and W rounded to bf16, fp32 accumulation -- at stride 1 on the forward table with the
    # offsets of the weight image mirrored (cnrma_sparse_conv_prepare_weights_bf16_t), else on the transposed table
    if cout % 32 == 0:
        assert "cnrma_sparse_conv_prepare_weights_bf16_t" in calls
        assert ("cnrma_sparse_kernel_map_transpose" in calls) == (stride != 1)
    gF, _ = SO.conv_backward(c, f, _bf16_round(W), _bf16_round(g), 3, stride, 2, out_coords=oc)
    gF_exact, _ = SO.conv_backward(c, f, W, g, 3, stride, 2, out_coords=oc)
    got_f = x.F.grad.cpu().numpy()
    if cout % 32 == 0:
        np.testing.assert_allclose(got_f, gF, rtol=2e-5, atol=2e-5 * np.abs(gF).max())
    assert np.abs(got_f - gF_exact).max() / np.abs(gF_exact).max() < 2e-2


def test_mirrored_data_gradient_equals_the_transposed_table(device):
    """stride-1 convolution: the data gradient on the forward table with mirrored weight offsets is the one over the
    transposed table, bit for bit (same products, same order over k reversed -> compared at 1e-6), in fp32 and bf16"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(77)
    c = np.unique(np.concatenate((rng.randint(0, 2, size=(6000, 1)), rng.randint(-12, 12, size=(6000, 3))), axis=1), axis=0)
    f = rng.randn(len(c), 64).astype(np.float32)
    W = (rng.randn(27, 64, 64) / 20).astype(np.float32)
    g = rng.randn(len(c), 64).astype(np.float32)
    grads = {}
    for prec in ("f32", "bf16"):
        for mirror in (True, False):
            S.DGRAD_MIRROR = mirror
            try:
                x = S.SparseTensor(torch.from_numpy(f).to(device).requires_grad_(True),
                                   S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 1))
                Wd = torch.from_numpy(W).to(device).requires_grad_(True)
                y = S.conv_autograd(x, Wd, 3, 1, precision=prec)
                y.F.backward(torch.from_numpy(g).to(device))
                grads[prec, mirror] = x.F.grad.cpu().numpy()
            finally:
                S.DGRAD_MIRROR = True
        a, b = grads[prec, True], grads[prec, False]
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), prec


def test_train_step_under_bf16_autocast_tracks_the_fp32_step(device, tmp_path):
    import projects.mvsdetection  # noqa: F401
    from cnrma_amd import synth
    from projects.mvsdetection.registry import build_model
    sc = synth.make_scene("tiny", seed=6)
    C = 32
    g = torch.Generator().manual_seed(0)
    feats0 = torch.randn(sc["features"].shape[0], C, *sc["features"].shape[3:], generator=g)
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_arkit.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path=str(tmp_path / "r"),
             voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]), max_points=None, use_feature_transform=False,
             detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=14))
    ext = np.array(sc["dims"], dtype=np.float32) * 0.04
    boxes = torch.tensor([[0.35 * ext[0], 0.4 * ext[1], 0.1 * ext[2], 0.5, 0.4, 0.5, 0.3],
                          [0.65 * ext[0], 0.6 * ext[1], 0.2 * ext[2], 0.4, 0.6, 0.4, -0.4]], device=device)

    def run(autocast):
        torch.manual_seed(2)
        model = build_model(dict(m))
        model.detection_backbone.init_weights()
        model.detection_head.init_weights()
        model = model.to(device).train()
        feats = feats0.to(device).requires_grad_(True)
        data = dict(features=[feats], projection=[sc["projection"][:, 0].to(device)], tsdf=sc["tsdf"].to(device),
                    offset=[torch.zeros(3, device=device)], gt_bboxes_3d=[boxes.clone()],
                    gt_labels_3d=[torch.tensor([1, 3], device=device)])
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = model.train_step(data, None)
        out["loss"].backward()
        grads = torch.cat([p.grad.flatten() for n, p in sorted(model.named_parameters()) if p.grad is not None])
        return {k: float(v) for k, v in out["log_vars"].items()}, grads, feats.grad.flatten()

    l32, g32, f32 = run(False)
    l16, g16, f16 = run(True)
    assert {"loss_centerness", "loss_bbox", "loss_cls"} <= set(l32)            # rotated IoU loss of the ARKit head included
    for k in l32:
        assert abs(l16[k] - l32[k]) <= 3e-2 * max(1.0, abs(l32[k])), (k, l16[k], l32[k])
    cos = torch.nn.functional.cosine_similarity(g16, g32, dim=0)
    assert torch.isfinite(g16).all() and float(cos) > 0.98, float(cos)
    assert float(torch.nn.functional.cosine_similarity(f16, f32, dim=0)) > 0.95


@pytest.mark.parametrize("cfg_name", ["ray_marching_scannet.py", "ray_marching_arkit.py"])
def test_train_step_with_the_shipped_augmentation(device, tmp_path, cfg_name):
    """All six reference configs train WITH the point / box augmentation (ray_marching.py:339-407, fcaf3d_transforms.py:14-146:
    flips, rotation, scale, translation of the aggregated points and of the GT boxes).  The detector is built from the shipped
    model section with `use_feature_transform` left as shipped (True), the GT boxes come as the box object the dataset
    pipeline hands over; np.random is seeded.  What the transform did on the GPU inside the step == the same
    TransformFeaturesBBoxes on the CPU with the same seed (its helpers are pinned to the reference by
    tests/golden/point_transforms.npz), the transformed boxes reach the assigner, the step yields finite losses and
    gradients that reach the feature maps."""
    import projects.mvsdetection  # noqa: F401
    from cnrma_amd import synth
    from projects.mvsdetection.core.boxes import GTBoxes
    from projects.mvsdetection.datasets.pipelines.fcaf3d_transforms import TransformFeaturesBBoxes
    from projects.mvsdetection.registry import build_model
    sc = synth.make_scene("tiny", seed=6)
    C = 32
    g = torch.Generator().manual_seed(0)
    feats0 = torch.randn(sc["features"].shape[0], C, *sc["features"].shape[3:], generator=g)
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", cfg_name))
    m = dict(cfg["model"])
    assert m["use_feature_transform"] is True and m["feature_transform"]["flip_ratio_horizontal"] == 0.5
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path=str(tmp_path / "r"),
             voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]),
             detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=14))
    with_yaw = cfg_name.endswith("arkit.py")
    ext = np.array(sc["dims"], dtype=np.float32) * 0.04
    raw = torch.tensor([[0.35 * ext[0], 0.4 * ext[1], 0.1 * ext[2], 0.5, 0.4, 0.5, 0.3 if with_yaw else 0.0],
                        [0.65 * ext[0], 0.6 * ext[1], 0.2 * ext[2], 0.4, 0.6, 0.4, -0.4 if with_yaw else 0.0]])
    torch.manual_seed(2)
    model = build_model(dict(m))
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    model = model.to(device).train()
    assert isinstance(model.feature_transform, TransformFeaturesBBoxes)
    rec, inner = {}, model.feature_transform

    def spy(points, gt):
        rec["in_points"], rec["in_gt"] = points.detach().cpu().clone(), gt.tensor.detach().cpu().clone()
        points, gt = inner(points, gt)
        rec["out_points"], rec["out_gt"] = points.detach().cpu().clone(), gt.tensor.detach().cpu().clone()
        return points, gt
    model.feature_transform = spy
    seen = {}
    assign = model.detection_head.assigner.assign
    model.detection_head.assigner.assign = lambda pts, gt, lab: (seen.update(gt=gt.tensor.detach().cpu().clone()), assign(pts, gt, lab))[1]
    feats = feats0.to(device).requires_grad_(True)
    boxes = GTBoxes(raw.clone(), with_yaw=with_yaw).to(device)
    data = dict(features=[feats], projection=[sc["projection"][:, 0].to(device)], tsdf=sc["tsdf"].to(device),
                offset=[torch.zeros(3, device=device)], gt_bboxes_3d=[boxes], gt_labels_3d=[torch.tensor([1, 3], device=device)])
    np.random.seed(5)
    out = model.train_step(data, None)
    out["loss"].backward()
    # the same transform on the CPU, same seed, on what went in
    np.random.seed(5)
    cpu_gt = GTBoxes(rec["in_gt"].clone(), with_yaw=with_yaw)
    cpu_pts, cpu_gt = TransformFeaturesBBoxes(**cfg["model"]["feature_transform"])(rec["in_points"].clone(), cpu_gt)
    np.testing.assert_allclose(rec["out_points"].numpy(), cpu_pts.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(rec["out_gt"].numpy(), cpu_gt.tensor.numpy(), rtol=0, atol=2e-6)
    assert not torch.equal(rec["out_points"], rec["in_points"]) and not torch.equal(rec["out_gt"], rec["in_gt"])   # it did something
    assert torch.equal(seen["gt"], rec["out_gt"])                       # the assigner saw the TRANSFORMED boxes
    assert {"loss_centerness", "loss_bbox", "loss_cls"} <= set(out["log_vars"])
    assert all(np.isfinite(float(v)) for v in out["log_vars"].values())
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(gr).all() for gr in grads)
    assert feats.grad is not None and torch.isfinite(feats.grad).all() and float(feats.grad.abs().sum()) > 0


def test_eval_mode_forward_keeps_gradients_when_asked(device):
    """eval mode (folded BatchNorm, fused epilogues) with autograd on -- frozen-BN fine-tuning, input-gradient analysis --
    must not drop gradients silently: the fused convolution falls back to the differentiable one (same values)"""
    from cnrma_amd import nn as snn
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(3)
    c = np.unique(np.concatenate((np.zeros((3000, 1), dtype=np.int64), rng.randint(0, 24, size=(3000, 3)) * 2), axis=1), axis=0)
    f = torch.from_numpy(rng.randn(len(c), 32).astype(np.float32)).to(device)
    cs = S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 2)
    blk = snn.BasicBlock(32, 32).to(device).eval()
    with torch.no_grad():
        for bn in (blk.norm1.bn, blk.norm2.bn):
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
        ref = blk(S.SparseTensor(f, cs)).F                       # fused path
    x = f.clone().requires_grad_(True)
    y = blk(S.SparseTensor(x, cs)).F                            # grad enabled, eval mode
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    y.square().sum().backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0 and blk.conv1.kernel.grad is not None


@pytest.mark.parametrize("how", ["wrapper", "inner"])
def test_frozen_batchnorm_inside_a_train_mode_block_keeps_its_statistics(device, how):
    """norm_eval fine-tuning: the block stays in train mode, its BatchNorms are frozen with .eval() -- on the
    MinkowskiBatchNorm wrapper or, as mmcv's norm_eval does by walking modules(), on the nn.BatchNorm1d inside it.  The fused
    conv -> BatchNorm node must not be taken: running statistics untouched, output = the eval-mode (folded) block's, gradients
    still reach the input, the kernels and the BatchNorm's affine parameters."""
    from cnrma_amd import nn as snn
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(11)
    c = np.unique(np.concatenate((np.zeros((3000, 1), dtype=np.int64), rng.randint(0, 24, size=(3000, 3)) * 2), axis=1), axis=0)
    f = torch.from_numpy(rng.randn(len(c), 32).astype(np.float32)).to(device)
    cs = S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 2)
    blk = snn.BasicBlock(32, 32).to(device)
    seq = snn.FusedSequential(snn.MinkowskiConvolution(32, 32, kernel_size=3), snn.MinkowskiBatchNorm(32), snn.MinkowskiELU()).to(device)
    norms = [blk.norm1, blk.norm2, seq[1]]
    with torch.no_grad():
        for n in norms:
            n.bn.running_mean.normal_(0, 0.1); n.bn.running_var.uniform_(0.5, 1.5)
            n.bn.weight.uniform_(0.5, 1.5); n.bn.bias.normal_(0, 0.1)
    blk.eval(); seq.eval()
    with torch.no_grad():
        ref_blk, ref_seq = blk(S.SparseTensor(f, cs)).F, seq(S.SparseTensor(f, cs)).F
    blk.train(); seq.train()
    for n in norms:
        (n if how == "wrapper" else n.bn).eval()
    before = [(n.bn.running_mean.clone(), n.bn.running_var.clone(), n.bn.num_batches_tracked.clone()) for n in norms]
    x = f.clone().requires_grad_(True)
    y_blk, y_seq = blk(S.SparseTensor(x, cs)).F, seq(S.SparseTensor(x, cs)).F
    for n, (m, v, k) in zip(norms, before):
        assert torch.equal(n.bn.running_mean, m) and torch.equal(n.bn.running_var, v) and torch.equal(n.bn.num_batches_tracked, k)
    np.testing.assert_allclose(y_blk.detach().cpu().numpy(), ref_blk.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(y_seq.detach().cpu().numpy(), ref_seq.cpu().numpy(), rtol=1e-4, atol=1e-4)
    (y_blk.square().sum() + y_seq.square().sum()).backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0
    assert blk.conv1.kernel.grad is not None and blk.norm2.bn.weight.grad is not None and seq[1].bn.bias.grad is not None
    # and the unfrozen block does update them (the fused node is still what runs in plain training)
    for n in norms:
        n.train()
    blk(S.SparseTensor(f, cs))
    assert not torch.equal(blk.norm1.bn.running_mean, before[0][0])


def test_full_size_train_step_at_scannet_shape(device, tmp_path):
    """BASELINE configs[4] on one GPU at FULL size: RayMarching.train_step at the ScanNet training shape (40 views x 32 ch x
    120x160 -> 192x192x80, 4.14 M aggregated rows, max_points 500 000 drawn on the device, MinkResNet34 + head, the three
    detection losses, SGD) under bf16 autocast -- finite gradients that reach the 2D feature maps, a loss that falls over
    a few optimiser steps, and a first-step gradient that points where the fp32 step's does (cosine > 0.98).
    Reference: ray_marching.py:409-451, :592-623; fcaf3d_head.py:142-214."""
    import projects.mvsdetection  # noqa: F401
    from cnrma_amd import synth
    from projects.mvsdetection.registry import build_model
    sc = synth.make_scene("S", seed=0, boxes=3)
    C = sc["features"].shape[2]
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path=str(tmp_path / "r"),
             voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]), use_feature_transform=False,
             point_sampler="device", detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=34))
    dims = np.array(sc["dims"], dtype=np.float32) * 0.04
    rng = np.random.RandomState(0)
    boxes = torch.tensor([[rng.uniform(.2, .8) * dims[0], rng.uniform(.2, .8) * dims[1], rng.uniform(0, .3) * dims[2], .8, .6, .7]
                          for _ in range(12)], dtype=torch.float32, device=device)
    labels = torch.from_numpy(rng.randint(0, 18, size=12)).to(device)
    feats0 = sc["features"][:, 0].to(device)

    def make():
        torch.manual_seed(3)
        model = build_model(dict(m))
        model.detection_backbone.init_weights()
        model.detection_head.init_weights()
        return model.to(device).train()

    def step(model, autocast, feats):
        data = dict(features=[feats], projection=[sc["projection"][:, 0].to(device)], tsdf=sc["tsdf"].to(device),
                    offset=[torch.zeros(3, device=device)], gt_bboxes_3d=[boxes.clone()], gt_labels_3d=[labels])
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = model.train_step(data, None)
        return out

    # first-step gradients: bf16 autocast vs fp32 (same weights; the device sampler's call counter is rewound so that both
    # steps draw the same 500 000 of the 4.14 M points)
    from cnrma_amd import rma
    grads, losses_first = {}, {}
    for autocast in (False, True):
        rma._SAMPLE_CALLS[0] = 0
        model = make()
        feats = feats0.clone().requires_grad_(True)
        out = step(model, autocast, feats)
        assert out["num_samples"] == 1 and {"loss_centerness", "loss_bbox", "loss_cls"} <= set(out["log_vars"])
        out["loss"].backward()
        g = torch.cat([p.grad.flatten() for n, p in sorted(model.named_parameters()) if p.grad is not None])
        assert bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0
        assert feats.grad is not None and bool(torch.isfinite(feats.grad).all()) and float(feats.grad.abs().sum()) > 0
        assert len(model.points_detection[0]) == 500000
        grads[autocast] = g
        losses_first[autocast] = {k: float(v) for k, v in out["log_vars"].items()}
        if autocast:
            keep = model
    cos = float(torch.nn.functional.cosine_similarity(grads[True], grads[False], dim=0))
    assert cos > 0.98, cos
    # values, not only the direction: the three losses of the bf16 step against the fp32 step on the same weights
    # and the same 500 000 points -- bf16 operands (2^-8 relative) through 35 layers, fp32 accumulation: within 3 %
    for k, v32 in losses_first[False].items():
        assert abs(losses_first[True][k] - v32) <= 3e-2 * max(1.0, abs(v32)), (k, losses_first[True][k], v32)
    rel = float((grads[True] - grads[False]).norm() / grads[False].norm())
    assert rel < 0.2, rel                                    # cos > 0.98 <=> relative gradient error below ~0.2
    # a few optimiser steps under autocast on the same scene: the loss falls
    model = keep
    opt = torch.optim.SGD(model.parameters(), lr=2e-3, momentum=0.9)
    losses = []
    for it in range(6):
        opt.zero_grad()
        out = step(model, True, feats0)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0], losses


@pytest.mark.parametrize("n,C", [(5000, 64), (777, 128), (40000, 32), (300, 256)])
@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (True, True), (False, True), ("elu", False)])
def test_batch_norm_train_kernels_match_torch(device, n, C, relu, with_res):
    """MinkowskiBatchNorm in training mode on the library's kernels (cnrma_bn_train_forward_f32 / _backward_f32), alone and
    fused with the shortcut add and the ReLU behind it: output, running statistics and all gradients (input, residual, weight,
    bias) against nn.BatchNorm1d (+ add + relu) composed in torch"""
    from cnrma_amd import sparse as S
    torch.manual_seed(n + C)
    x = (torch.randn(n, C, device=device) * 3 + 1.5)
    g = torch.randn(n, C, device=device)
    r = torch.randn(n, C, device=device)
    ref = torch.nn.BatchNorm1d(C).to(device).train()
    got = torch.nn.BatchNorm1d(C).to(device).train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.normal_()
        got.load_state_dict(ref.state_dict())
    xr = x.clone().requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if with_res else None
    rg = r.clone().requires_grad_(True) if with_res else None
    yr = ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.nn.functional.elu(yr) if relu == "elu" else torch.relu(yr)
    prev, S.BN_TRAIN_HIP = S.BN_TRAIN_HIP, True
    calls = []
    orig = S.call
    S.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        yg = S.batch_norm_train(xg, got, relu=relu, residual=rg)
        yr.backward(g)
        yg.backward(g)
    finally:
        S.BN_TRAIN_HIP = prev
        S.call = orig
    assert calls == ["cnrma_bn_train_forward_f32", "cnrma_bn_train_backward_f32"]
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-3, atol=2e-5)
    if with_res:
        # rows where the fused output sits within rounding of the ReLU's kink may take the other branch
        mask = (yr.detach().abs() > 1e-5).cpu().numpy() if relu is True else np.ones((n, C), bool)
        np.testing.assert_allclose(rg.grad.cpu().numpy()[mask], rr.grad.cpu().numpy()[mask], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(got.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(got.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert int(got.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("cin,cout,stride,morton", [(64, 64, 1, True), (64, 64, 1, False), (32, 64, 2, True), (128, 96, 1, True),
                                                    (256, 128, 1, False), (36, 20, 1, True)])
def test_gather_once_weight_gradient_equals_the_block_kernel(device, cin, cout, stride, morton):
    """cnrma_sparse_conv_wgrad_go_bf16 (tile unions: compact Morton rows = one group per tile, shuffled rows = several) against
    the block kernel on the same operands -- same bf16 products, another summation order -- and against the fp64 oracle"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + cout + stride)
    c = np.unique(np.concatenate((rng.randint(0, 2, size=(9000, 1)), rng.randint(-14, 14, size=(9000, 3)) * 2), axis=1), axis=0)
    if morton:
        key = (c[:, 0].astype(np.int64) << 60)
        cc = ((c[:, 1:] + 64) // 2).astype(np.int64)
        for b in range(8):
            for a in range(3):
                key |= ((cc[:, a] >> b) & 1) << (3 * b + 2 - a)
        c = c[np.argsort(key, kind="stable")]
    else:
        c = c[rng.permutation(len(c))]
    f = rng.randn(len(c), cin).astype(np.float32)
    W = (rng.randn(27, cin, cout) / 20).astype(np.float32)
    got = {}
    g = None
    for go in (True, False):
        S.WGRAD_GO = go
        try:
            x = S.SparseTensor(torch.from_numpy(f).to(device).requires_grad_(True), S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 2))
            Wd = torch.from_numpy(W).to(device).requires_grad_(True)
            y = S.conv_autograd(x, Wd, 3, stride, precision="bf16")
            if g is None:
                g = rng.randn(*y.F.shape).astype(np.float32)
            calls = []
            orig = S.call
            S.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
            try:
                y.F.backward(torch.from_numpy(g).to(device))
            finally:
                S.call = orig
            assert ("cnrma_sparse_conv_wgrad_go_bf16" in calls) == go and ("cnrma_sparse_conv_wgrad_bf16" in calls) == (not go)
            got[go] = Wd.grad.cpu().numpy()
            oc = y.C.cpu().numpy().astype(np.int64)
        finally:
            S.WGRAD_GO = "auto"
    scale = np.abs(got[False]).max()
    assert np.abs(got[True] - got[False]).max() <= 2e-5 * scale
    _, gW = SO.conv_backward(c, _bf16_round(f), W, _bf16_round(g), 3, stride, 2, out_coords=oc)
    np.testing.assert_allclose(got[True], gW, rtol=2e-5, atol=2e-5 * np.abs(gW).max())


@pytest.mark.parametrize("cin,cout,n_pts", [(64, 64, 9000), (64, 128, 9000), (128, 64, 2500), (256, 256, 700)])
def test_gather_once_bf16_forward_and_data_gradient_equal_the_stage_kernel(device, cin, cout, n_pts):
    """cnrma_sparse_conv_go_bf16 (forward, and the data gradient with the mirrored + transposed fragment image) against
    cnrma_sparse_conv_bf16 on the same operands -- same bf16 products, another fp32 summation order -- and the forward against
    the fp64 oracle on the rounded operands; small sets take the split over channel slices"""
    from cnrma_amd import sparse as S
    rng = np.random.RandomState(cin + cout)
    c = np.unique(np.concatenate((rng.randint(0, 2, size=(n_pts, 1)), rng.randint(-14, 14, size=(n_pts, 3))), axis=1), axis=0)
    f = rng.randn(len(c), cin).astype(np.float32)
    W = (rng.randn(27, cin, cout) / 20).astype(np.float32)
    g = rng.randn(len(c), cout).astype(np.float32)
    res = {}
    for go in (True, False):
        S.TRAIN_GO = go
        try:
            x = S.SparseTensor(torch.from_numpy(f).to(device).requires_grad_(True), S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 1))
            Wd = torch.from_numpy(W).to(device).requires_grad_(True)
            calls = []
            orig = S.call
            S.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
            try:
                y = S.conv_autograd(x, Wd, 3, 1, precision="bf16")
                y.F.backward(torch.from_numpy(g).to(device))
            finally:
                S.call = orig
            assert (calls.count("cnrma_sparse_conv_go_bf16") == 2) == go and ("cnrma_sparse_conv_bf16" in calls) == (not go)
            res[go] = (y.F.detach().cpu().numpy(), x.F.grad.cpu().numpy(), y.C.cpu().numpy().astype(np.int64))
        finally:
            S.TRAIN_GO = "auto"
    for a, b in zip(res[True][:2], res[False][:2]):
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
    oc, exp = SO.conv(c, _bf16_round(f), _bf16_round(W), 3, 1, 1)
    got_c = res[True][2]
    o1, o2 = np.lexsort(got_c.T[::-1]), np.lexsort(oc.T[::-1])
    assert np.array_equal(got_c[o1], oc[o2])
    np.testing.assert_allclose(res[True][0][o1], exp[o2], rtol=1e-5, atol=1e-5 * np.abs(exp).max())
    # the DATA gradient of the gather-once path directly against the fp64 oracle (it was only compared with the stage
    # kernel): grad_in = sum_k G[o] @ W[k]^T over the pairs, with the operands the kernel multiplies -- grad_out and W rounded to bf16
    # (the rows of y follow the input rows here: stride 1, out set = in set)
    assert np.array_equal(got_c, c)
    gF, _ = SO.conv_backward(c, f, _bf16_round(W), _bf16_round(g), 3, 1, 1)
    np.testing.assert_allclose(res[True][1], gF, rtol=2e-5, atol=2e-5 * np.abs(gF).max())


@pytest.mark.parametrize("inplanes,planes,stride,prec", [(64, 64, 1, "bf16"), (64, 128, 2, "bf16"), (32, 64, 1, None)])
def test_fused_conv_batchnorm_node_equals_the_module_composition(device, inplanes, planes, stride, prec):
    """a BasicBlock in training mode with conv -> BatchNorm -> [+ shortcut] -> ReLU as one autograd node
    (sparse.conv_bn_act_train) against the same block composed module by module: outputs, every gradient and the running
    statistics (the same kernels in the same order: equal up to nothing but identical launches -> exact)"""
    import copy
    from cnrma_amd import nn as snn, sparse as S
    rng = np.random.RandomState(inplanes + planes + stride)
    c = np.unique(np.concatenate((rng.randint(0, 2, size=(7000, 1)), rng.randint(-12, 12, size=(7000, 3))), axis=1), axis=0)
    f = rng.randn(len(c), inplanes).astype(np.float32)
    down = None
    if stride != 1 or inplanes != planes:
        down = snn.FusedSequential(snn.MinkowskiConvolution(inplanes, planes, kernel_size=1, stride=stride, dimension=3),
                                   snn.MinkowskiBatchNorm(planes))
    torch.manual_seed(3)
    blk = snn.BasicBlock(inplanes, planes, stride=stride, downsample=down).to(device).train()
    res = {}
    for fused in (True, False):
        b = copy.deepcopy(blk)
        S.FUSE_CONV_BN = fused
        try:
            x = S.SparseTensor(torch.from_numpy(f).to(device).requires_grad_(True), S.CoordSet(torch.from_numpy(c.astype(np.int32)).to(device), 1))
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=prec == "bf16"):
                y = b(x)
            g = torch.from_numpy(np.random.RandomState(1).randn(*y.F.shape).astype(np.float32)).to(device)
            y.F.backward(g)
        finally:
            S.FUSE_CONV_BN = True
        res[fused] = ([y.F.detach().cpu().numpy(), x.F.grad.cpu().numpy()] + [p.grad.cpu().numpy() for p in b.parameters()]
                      + [bf.cpu().numpy().astype(np.float64) for bf in b.buffers()])
    assert len(res[True]) == len(res[False]) > 6
    for a, e in zip(res[True], res[False]):
        np.testing.assert_array_equal(a, e)
