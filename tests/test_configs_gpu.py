"""The BASELINE.json configurations that were not exercised on the GPU before: ARKit (17 classes, 8 regression outputs,
yaw decode) at the full S geometry through the registered detector; the north-star shape NS (480x640 planes, 256
channels, 192^3 grid): the C = 256 kernel paths against the oracle and full-size invariants of the 12.3 M-ray march;
the ScanNet test shape St (50 views, 256 x 256 x 96) end to end."""
import os
import runpy

import numpy as np
import pytest
import torch

from helpers import count_mismatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arkit_config_forward_test_at_full_geometry(device, tmp_path):
    """ray_marching_arkit.py (hot path): V = 40, 32 x 120 x 160 maps, 192 x 192 x 80 grid, max_points = 500 000"""
    import projects.mvsdetection  # noqa: F401
    from cnrma_amd import pipeline, synth
    from oracle import rma_oracle as O
    from projects.mvsdetection.registry import build_model
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_arkit.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None, save_path=str(tmp_path / "r"))
    m.update(point_sampler="numpy")            # the reference's RNG stream: the subset below is reproduced with np.random.seed
    assert m["voxel_dim_test"] == [192, 192, 80] and m["detection_head"]["n_reg_outs"] == 8
    torch.manual_seed(0)
    model = build_model(m)
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    model = model.to(device).eval()
    sc = synth.make_scene("S", seed=3, boxes=3)
    feat, proj, tsdf = sc["features"][:, 0].to(device), sc["projection"][:, 0], sc["tsdf"].to(device)
    data = dict(features=[feat], projection=[proj.to(device)], tsdf=tsdf, offset=[torch.tensor([0.5, -0.25, 0.125], device=device)],
                scene=["41069021"])
    np.random.seed(11)
    with torch.no_grad():
        assert model(return_loss=False, **data) == [{}]
    M = model.points_detection[0].shape[0]
    assert M > 3_000_000
    z = np.load(tmp_path / "r" / "41069021" / "41069021_bbox_raw.npz")
    assert z["bboxes"].shape[1] == 7 and z["scores"].shape[1] == 17 and z["bboxes"].shape[0] == z["scores"].shape[0] <= 4000
    assert np.isfinite(z["bboxes"]).all() and (z["scores"] >= 0).all() and (z["scores"] <= 1).all()
    assert (z["bboxes"][:, 3:6] > 0).all() and (np.abs(z["bboxes"][:, 6]) <= np.pi / 2 + 1e-6).all()    # 0.5 * atan2
    # the same scene through the fused pipeline with the same numpy mask: identical raw boxes
    np.random.seed(11)
    mask = O.sample_mask_numpy(M, 500000)
    pcfg = pipeline.SceneConfig(sc["dims"], stride=4, max_points=500000, sampler="numpy")
    out = pipeline.forward_scene(pcfg, model.detection_backbone, model.detection_head, feat, proj, tsdf[0, 0],
                                 offset=(0.5, -0.25, 0.125), mask=mask)
    assert out["M"] == M and out["bboxes"].shape == z["bboxes"].shape
    np.testing.assert_allclose(z["bboxes"], out["bboxes"].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(z["scores"], out["scores"].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_dense_unprojection_c256_planes_480x640(device):
    """the 256-channel path of the dense kernel (8 channel sweeps) on north-star planes against the oracle, bit for bit"""
    from cnrma_amd import rma, synth
    from oracle import rma_oracle as O
    sc = synth.make_scene((3, 256, 480, 640, (64, 64, 48), 1), seed=2)
    feat, proj = sc["features"][:, 0], sc["projection"][:, 0]
    vol, cnt = rma.backproject_accum(rma.to_nhwc(feat.to(device)), proj, sc["dims"], 0.04, sc["origin"], 1)
    evol, ecnt = O.backproject_accum(sc["dims"], 0.04, sc["origin"], proj, feat, 1)
    assert torch.equal(cnt.cpu().long(), ecnt) and count_mismatch(vol, evol) == 0
    assert int((ecnt > 0).sum()) > 10000


def test_rma_one_northstar_view_c256_vs_oracle(device):
    """one 480 x 640 view with 256 channels into the 192^3 TSDF: kept set, order, places and features bit-exact, weights
    within 1 ulp of the oracle (the one-wave-per-row emission path of C = 256)"""
    from cnrma_amd import rma, synth
    from oracle import rma_oracle as O
    sc = synth.make_scene((1, 256, 480, 640, (192, 192, 192), 1), seed=1, boxes=2)
    feat, proj, tsdf = sc["features"][:, 0], sc["projection"][:, 0], sc["tsdf"][0, 0]
    exp = O.rma_neus_view(O.scale_projection(proj[0], 1), feat[0], tsdf, sc["dims"], 0.04, sc["origin"])
    pinv = rma.projection_inverse(proj, 1).to(device)
    rows, per_view = rma.rma_view_rows(rma.to_nhwc(feat.to(device)), pinv, tsdf.to(device), sc["dims"], 0.04, sc["origin"])
    assert rows.shape == exp.shape and rows.shape[0] > 1_000_000 and int(per_view[0]) == exp.shape[0]
    rows = rows.cpu()
    assert count_mismatch(rows[:, :3], exp[:, :3]) == 0 and count_mismatch(rows[:, 4:], exp[:, 4:]) == 0
    ulp = (rows[:, 3].view(torch.int32) - exp[:, 3].view(torch.int32)).abs()
    assert int(ulp.max()) <= 1


def test_northstar_march_full_size_invariants(device):
    """12.3 M rays (40 x 480 x 640) through 192^3: the table-driven single march == the two-pass march that evaluates the
    sigmoid and the IEEE divisions at every step (same rows, same order, same bits), and it is deterministic"""
    from cnrma_amd import rma, synth
    sc = synth.make_scene((40, 4, 480, 640, (192, 192, 192), 1), seed=0, boxes=3)
    nhwc = rma.to_nhwc(sc["features"][:, 0].to(device))
    pinv = rma.projection_inverse(sc["projection"][:, 0], 1).to(device)
    tsdf = sc["tsdf"][0, 0].to(device)
    # the free-space skip table is a default of this size only ("auto": from MARCH_SKIP_MIN_RAYS rays on): assert that it WAS in
    # use -- table builder and march both handed a non-null skip pointer -- or the comparison below proves nothing about it
    assert rma.MARCH_SKIP == "auto" and 40 * 480 * 640 >= rma.MARCH_SKIP_MIN_RAYS
    seen, orig = {}, rma.call

    def spy(name, *args):
        if name == "cnrma_rma_march_tables_f32":
            seen["tables_skip"] = args[5]
        elif name == "cnrma_rma_neus_march_f32":
            seen["march_skip"] = args[-2]
        return orig(name, *args)
    rma.call = spy
    try:
        a, pa = rma.rma_view_rows(nhwc, pinv, tsdf, sc["dims"], 0.04, sc["origin"], single_march=True)
    finally:
        rma.call = orig
    assert seen.get("tables_skip") and seen.get("march_skip") and seen["tables_skip"] == seen["march_skip"], seen
    assert a.shape[0] > 60_000_000 and bool((pa > 1_000_000).all())
    b, pb = rma.rma_view_rows(nhwc, pinv, tsdf, sc["dims"], 0.04, sc["origin"], single_march=False)
    assert torch.equal(pa, pb) and torch.equal(a.view(torch.int32), b.view(torch.int32))
    del b
    c, _ = rma.rma_view_rows(nhwc, pinv, tsdf, sc["dims"], 0.04, sc["origin"], single_march=True)
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    w = a[:, 3]
    assert float(w.min()) >= 0.05 and float(w.max()) <= 1.0


def test_scannet_test_shape_end_to_end(device):
    """St: 50 views into 256 x 256 x 96 (ray_marching_scannet.py voxel_dim_test): eager forward and its static graph"""
    from bench import build_model
    from cnrma_amd import pipeline, synth
    sc = synth.make_scene("St", seed=2, boxes=4)
    feat, proj, tsdf = sc["features"][:, 0].to(device), sc["projection"][:, 0], sc["tsdf"][0, 0].to(device)
    backbone, head = build_model(32, device)
    cfg = pipeline.SceneConfig(sc["dims"], stride=4, max_points=500000, sample_seed=5)
    st = pipeline.StaticScene(cfg, backbone, head, device)
    eager = st.build(feat, proj, tsdf)
    assert eager["M"] > 4_000_000 and eager["M_selected"] == 500000 and tuple(eager["volume"].shape) == (32, 256, 256, 96)
    st.seed_dev.zero_()
    out = st.run(feat, proj, tsdf)
    torch.cuda.synchronize()
    b, s, info = pipeline.StaticScene.detections(out)
    assert info["M"] == eager["M"] and info["level_rows"] == eager["level_rows"] and b.shape == eager["bboxes"].shape
    assert torch.equal(out["volume"], eager["volume"]) and torch.isfinite(b).all()
