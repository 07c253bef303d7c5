"""GPU test of the drop-in boundary: the registered RayMarching detector (reference API: forward(return_loss=False, **data)
-> [{}], raw boxes dumped to {save_path}/{scene}/{scene}_bbox_raw.npz) against the oracle / the scene pipeline."""
import os
import runpy

import numpy as np
import pytest
import torch

from helpers import count_mismatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(tmp_path, dims, device, max_points, **kw):
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None)      # hot path only: features / TSDF come in
    m.update(save_path=str(tmp_path / "results"), voxel_dim_test=list(dims), voxel_dim_train=list(dims), max_points=max_points)
    m.update(kw)
    m["detection_backbone"] = dict(type="FCAF3DBackbone", in_channels=8, depth=34)
    model = build_model(m)
    torch.manual_seed(0)
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    return model.to(device).eval()


def test_raymarching_forward_test_matches_oracle_and_pipeline(device, tmp_path):
    from cnrma_amd import pipeline, synth
    from oracle import rma_oracle as O
    sc = synth.make_scene("tiny", seed=0)
    V = sc["features"].shape[0]
    model = _model(tmp_path, sc["dims"], device, max_points=400, point_sampler="numpy")     # the reference's RNG stream (parity switch)
    feats = [sc["features"][:, 0].to(device)]                                 # per sample: [V, C, H', W']
    data = dict(features=feats, projection=[sc["projection"][:, 0].to(device)], tsdf=sc["tsdf"].to(device),
                offset=[torch.tensor([0.25, -0.5, 0.125], device=device)], scene=["scene0000_00"])
    np.random.seed(3)
    with torch.no_grad():
        ret = model(return_loss=False, **data)
    assert ret == [{}]                                                       # reference contract (ray_marching.py:521)
    # dense half: volume / valid as the reference leaves them on the module (B x C x X x Y x Z, B x 1 x X x Y x Z bool)
    vol, cnt = O.backproject_accum(sc["dims"], 0.04, sc["origin"], sc["projection"][:, 0], sc["features"][:, 0], sc["stride"])
    assert count_mismatch(model.volume[0], vol) == 0 and torch.equal(model.valid[0, 0].cpu(), cnt > 0)
    # aggregated points == oracle
    pts = O.aggregate_rma(sc["projection"][:, 0], sc["features"][:, 0], sc["tsdf"][0, 0], sc["dims"], 0.04, sc["origin"], sc["stride"])
    got = model.points_detection[0].cpu()
    assert count_mismatch(got[:, :3], pts[:, :3]) == 0
    np.testing.assert_allclose(got[:, 3:].numpy(), pts[:, 3:].numpy(), rtol=1e-4, atol=1e-4)
    # raw boxes dumped with the reference's file layout and keys; same as the fused scene pipeline given the same mask
    z = np.load(tmp_path / "results" / "scene0000_00" / "scene0000_00_bbox_raw.npz")
    assert set(z.files) == {"bboxes", "scores"} and z["bboxes"].shape[1] == 6 and z["scores"].shape[1] == 18
    np.random.seed(3)
    mask = O.sample_mask_numpy(pts.shape[0], 400)
    cfg = pipeline.SceneConfig(sc["dims"], stride=sc["stride"], max_points=400, sampler="numpy")
    out = pipeline.forward_scene(cfg, model.detection_backbone, model.detection_head, sc["features"][:, 0].to(device),
                                 sc["projection"][:, 0], sc["tsdf"][0, 0].to(device), offset=(0.25, -0.5, 0.125), mask=mask)
    np.testing.assert_allclose(z["bboxes"], out["bboxes"].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(z["scores"], out["scores"].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_module_level_functions_keep_the_reference_signatures(device):
    from cnrma_amd import synth
    from oracle import rma_oracle as O
    from projects.mvsdetection.models import ray_marching as RM
    sc = synth.make_scene("tiny", seed=2)
    proj = O.scale_projection(sc["projection"][0], sc["stride"]).to(device)          # [B=1,3,4]
    feat = sc["features"][0].to(device)                                             # [B=1,C,H,W]
    vol, valid = RM.backproject(sc["dims"], 0.04, torch.tensor(sc["origin"]).view(1, 3), proj, feat)
    ov, ovalid, _, _ = O.backproject_view(sc["dims"], 0.04, sc["origin"], proj[0].cpu(), feat[0].cpu())
    assert tuple(vol.shape) == (1, feat.shape[1], *sc["dims"]) and tuple(valid.shape) == (1, 1, *sc["dims"])
    assert count_mismatch(vol[0].reshape(feat.shape[1], -1), ov) == 0
    assert torch.equal(valid[0, 0].reshape(-1).cpu(), ovalid)
    o, d = RM.get_ray_parameter(proj, feat)
    oo, od = O.ray_params(proj[0].cpu(), feat.shape[2], feat.shape[3])
    assert tuple(o.shape) == (1, 3, feat.shape[2] * feat.shape[3])
    assert count_mismatch(d[0], od) == 0 and count_mismatch(o[0, :, 0], oo) == 0


def test_raymarching_with_atlas3d_predicts_its_own_tsdf(device, tmp_path):
    """config with backbone_3d + tsdf_head: dense unprojection -> 3D U-Net -> TSDF head ->
    ray marching on the PREDICTED scene_tsdf_004 -> FCAF3D.  The TSDF used by the march equals the CPU evaluation
    of the same torch modules on the same volume; the aggregated points equal the oracle's on that TSDF."""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    from cnrma_amd import synth
    from oracle import rma_oracle as O
    sc = synth.make_scene("tiny", seed=5)
    C = sc["features"].shape[2]
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None)      # hot path only: features / TSDF come in
    m.update(save_path=str(tmp_path / "r"), voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]), max_points=None,
             detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=34),
             backbone_3d=dict(type="AtlasBackbone3D", channels=[C, 16, 32], layers_down=[1, 1, 1], layers_up=[1, 1], drop=0.0,
                              zero_init_residual=False, cond_proj=False, norm="BN"),
             tsdf_head=dict(type="AtlasTSDFHead", input_channels=[C, 16], n_scales=2, voxel_size=0.04, label_smoothing=1.05,
                            sparse_threshold=[0.99]))
    torch.manual_seed(1)
    model = build_model(m)
    model.detection_head.init_weights()
    model = model.to(device).eval()
    data = dict(features=[sc["features"][:, 0].to(device)], projection=[sc["projection"][:, 0].to(device)],
                offset=[torch.zeros(3, device=device)], scene=["s"])
    with torch.no_grad():
        assert model(return_loss=False, **data) == [{}]
        vol = model.volume.cpu()
        net, head = model.backbone3d.cpu().float(), model.tsdf_head.cpu()
        tsdf_cpu = head(net(vol))[0]["scene_tsdf_004"]
    assert tuple(tsdf_cpu.shape) == (1, 1, *sc["dims"])
    # the march ran on the GPU prediction; rebuild it there and compare with the CPU evaluation of the same modules
    model = model.to(device)
    with torch.no_grad():
        tsdf_gpu = model.tsdf_head(model.backbone3d(model.volume))[0]["scene_tsdf_004"].cpu()
    np.testing.assert_allclose(tsdf_gpu.numpy(), tsdf_cpu.numpy(), rtol=1e-3, atol=2e-4)
    pts = O.aggregate_rma(sc["projection"][:, 0], sc["features"][:, 0], tsdf_gpu[0, 0], sc["dims"], 0.04, sc["origin"], sc["stride"])
    got = model.points_detection[0].cpu()
    assert got.shape == pts.shape and count_mismatch(got[:, :3], pts[:, :3]) == 0
    assert os.path.exists(tmp_path / "r" / "s" / "s_bbox_raw.npz")


@pytest.mark.parametrize("sampler,max_points", [("numpy", None), ("device", 3000), ("numpy", 3000)])
def test_raymarching_train_step(device, tmp_path, sampler, max_points):
    """(point_sampler="device": the max_points selection is drawn on the GPU and fused into the aggregation)
    SURVEY.md 8f rank 3 end to end: train_step of the registered detector on a synthetic scene with ground-truth boxes --
    differentiable aggregation (gradient reaches the 2D feature maps), sparse network on the dgrad / wgrad kernels, FCAF3D
    assignment + centerness / IoU / focal losses; a few SGD steps on the same scene lower the loss"""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    from cnrma_amd import synth
    sc = synth.make_scene("tiny", seed=6)
    C = sc["features"].shape[2]
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None)      # hot path only: features / TSDF come in
    m.update(save_path=str(tmp_path / "r"), voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]), max_points=max_points,
             point_sampler=sampler, use_feature_transform=False, detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=14))
    torch.manual_seed(2)
    np.random.seed(3)
    model = build_model(m)
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    model = model.to(device).train()
    dims = np.array(sc["dims"], dtype=np.float32) * 0.04
    boxes = torch.tensor([[0.35 * dims[0], 0.4 * dims[1], 0.1 * dims[2], 0.5, 0.4, 0.5, 0.0],
                          [0.65 * dims[0], 0.6 * dims[1], 0.2 * dims[2], 0.4, 0.6, 0.4, 0.0]], device=device)
    feats = sc["features"][:, 0].to(device).requires_grad_(True)
    data = dict(features=[feats], projection=[sc["projection"][:, 0].to(device)], tsdf=sc["tsdf"].to(device),
                offset=[torch.zeros(3, device=device)], gt_bboxes_3d=[boxes], gt_labels_3d=[torch.tensor([1, 3], device=device)])
    opt = torch.optim.SGD(model.parameters(), lr=2e-3)
    losses = []
    for it in range(4):
        out = model.train_step(dict(data), None)
        assert set(out["log_vars"]) >= {"loss_centerness", "loss_bbox", "loss_cls", "total_loss"}
        opt.zero_grad()
        if feats.grad is not None:
            feats.grad = None
        out["loss"].backward()
        if max_points is not None:
            assert model.points_detection[0].shape[0] <= max(max_points, 0) or sampler == "numpy"
        if it == 0:
            assert feats.grad is not None and torch.isfinite(feats.grad).all() and float(feats.grad.abs().sum()) > 0
            assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_raymarching_train_step_with_atlas3d(device, tmp_path):
    """joint training step with the Atlas 3D network in the loop: TSDF losses (their gradient reaches the 2D feature maps
    through the dense unprojection's backward) + detection losses (through the aggregation's backward)"""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    from cnrma_amd import synth
    sc = synth.make_scene("tiny", seed=9)
    C = sc["features"].shape[2]
    X, Y, Z = sc["dims"]
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    m = dict(cfg["model"])
    m.update(backbone2d=None, feature_2d=None, backbone_3d=None, tsdf_head=None)      # hot path only: features / TSDF come in
    m.update(save_path=str(tmp_path / "r"), voxel_dim_test=list(sc["dims"]), voxel_dim_train=list(sc["dims"]), max_points=None,
             use_feature_transform=False, detection_backbone=dict(type="FCAF3DBackbone", in_channels=C, depth=14),
             backbone_3d=dict(type="AtlasBackbone3D", channels=[C, 16, 32], layers_down=[1, 1, 1], layers_up=[1, 1], drop=0.0,
                              zero_init_residual=False, cond_proj=False, norm="BN"),
             tsdf_head=dict(type="AtlasTSDFHead", input_channels=[C, 16], n_scales=2, voxel_size=0.04, label_smoothing=1.05,
                            sparse_threshold=[0.99]))
    torch.manual_seed(4)
    model = build_model(m)
    model.detection_backbone.init_weights(); model.detection_head.init_weights()
    model = model.to(device).train()
    dims = np.array(sc["dims"], dtype=np.float32) * 0.04
    boxes = torch.tensor([[0.5 * dims[0], 0.5 * dims[1], 0.1 * dims[2], 0.6, 0.5, 0.5, 0.0]], device=device)
    feats = sc["features"][:, 0].to(device).requires_grad_(True)
    gt = sc["tsdf"].to(device).clamp(-1, 1)
    tsdf_list = {"tsdf_gt_004": gt, "tsdf_gt_008": torch.nn.functional.avg_pool3d(gt, 2)}
    data = dict(features=[feats], projection=[sc["projection"][:, 0].to(device)], offset=[torch.zeros(3, device=device)],
                gt_bboxes_3d=[boxes], gt_labels_3d=[torch.tensor([2], device=device)], tsdf_list=tsdf_list)
    out = model.train_step(dict(data), None)
    assert {"tsdf_loss_004", "tsdf_loss_008", "loss_cls", "loss_bbox", "loss_centerness"} <= set(out["log_vars"])
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and torch.isfinite(feats.grad).all() and float(feats.grad.abs().sum()) > 0
    g3d = [p.grad for p in model.backbone3d.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in g3d)


def test_forward_test_graph_path_writes_the_same_files_as_the_eager_path(device, tmp_path):
    """the inference fast path of the plugin (point_sampler="device": scenes of a repeating shape as replayed HIP graphs,
    several in flight, files written when a slot is reused / at flush()) against the eager path of the same detector on
    the same scenes: same files, same keys, same row counts, boxes / scores within 1e-4 -- per-scene offsets included"""
    from cnrma_amd import synth
    scenes = []
    for i in range(7):
        sc = synth.make_scene("tiny", seed=i, boxes=i % 3)
        scenes.append(dict(features=[sc["features"][:, 0].to(device)], projection=[sc["projection"][:, 0].to(device)],
                           tsdf=sc["tsdf"].to(device), offset=[torch.tensor([0.25 * i, -0.5, 0.125 * (i % 2)], device=device)],
                           scene=[f"scene{i:04d}_00"]))
    dims = synth.SHAPES["tiny"][4]
    fast = _model(tmp_path / "fast", dims, device, max_points=100000)
    fast.point_sampler, fast.static_slots, fast.static_calibration = "device", 2, 3
    slow = _model(tmp_path / "slow", dims, device, max_points=100000)
    slow.point_sampler, slow.static_test = "device", False
    slow.load_state_dict(fast.state_dict())
    with torch.no_grad():
        for d in scenes:
            assert fast(return_loss=False, **d) == [{}]
            assert slow(return_loss=False, **d) == [{}]
    ctx = next(iter(fast._static.values()))
    assert ctx["built"] and len(ctx["slots"]) == 2 and ctx["k"] == 4            # 3 calibration scenes, 4 graph replays
    assert sum(p is not None for p in ctx["pending"]) == 2                      # two scenes still in flight
    fast.flush()
    assert getattr(fast, "static_fallbacks", 0) == 0
    for d in scenes:
        n = d["scene"][0]
        a = np.load(tmp_path / "fast" / "results" / n / f"{n}_bbox_raw.npz")
        b = np.load(tmp_path / "slow" / "results" / n / f"{n}_bbox_raw.npz")
        assert set(a.files) == set(b.files) == {"bboxes", "scores"} and a["bboxes"].shape == b["bboxes"].shape
        ka = np.lexsort(tuple(np.round(a["bboxes"][:, i], 3) for i in range(5, -1, -1)))
        kb = np.lexsort(tuple(np.round(b["bboxes"][:, i], 3) for i in range(5, -1, -1)))
        np.testing.assert_allclose(a["bboxes"][ka], b["bboxes"][kb], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(a["scores"][ka], b["scores"][kb], rtol=1e-4, atol=1e-6)


def _tiny_scenes(device, n, boxes=lambda i: i % 3):
    from cnrma_amd import synth
    out = []
    for i in range(n):
        sc = synth.make_scene("tiny", seed=i, boxes=boxes(i))
        out.append(dict(features=[sc["features"][:, 0].to(device)], projection=[sc["projection"][:, 0].to(device)],
                        tsdf=sc["tsdf"].to(device), offset=[torch.tensor([0.25 * i, -0.5, 0.125 * (i % 2)], device=device)],
                        scene=[f"scene{i:04d}_00"]))
    return out


def _same_files(dir_a, dir_b, names, tol=1e-4):
    for n in names:
        a = np.load(dir_a / "results" / n / f"{n}_bbox_raw.npz")
        b = np.load(dir_b / "results" / n / f"{n}_bbox_raw.npz")
        assert set(a.files) == set(b.files) == {"bboxes", "scores"} and a["bboxes"].shape == b["bboxes"].shape, n
        ka = np.lexsort(tuple(np.round(a["bboxes"][:, i], 3) for i in range(5, -1, -1)))
        kb = np.lexsort(tuple(np.round(b["bboxes"][:, i], 3) for i in range(5, -1, -1)))
        np.testing.assert_allclose(a["bboxes"][ka], b["bboxes"][kb], rtol=tol, atol=tol)
        np.testing.assert_allclose(a["scores"][ka], b["scores"][kb], rtol=tol, atol=1e-6)


def test_shipped_config_takes_the_graph_path_unmodified(device, tmp_path):
    """VERDICT round 3, missing #1: the detector built from the shipped config's model section -- no sampler / static keywords
    -- runs forward_test on the graph path (reference ray_marching.py:339-407,456-521: the max_points subset is part of
    forward_test; here it is drawn on the device unless point_sampler="numpy" asks for the reference's RNG stream)"""
    from cnrma_amd import synth
    dims = synth.SHAPES["tiny"][4]
    model = _model(tmp_path, dims, device, max_points=500000)
    assert model.point_sampler == "device" and model.static_test
    scenes = _tiny_scenes(device, 6)
    with torch.no_grad():
        for d in scenes:
            assert model._static_eligible(model.data_converter(dict(d)))
            assert model(return_loss=False, **d) == [{}]
    ctx = next(iter(model._static.values()))
    assert ctx["built"] and ctx["k"] == 6 - model.static_calibration and getattr(model, "static_fallbacks", 0) == 0
    model.flush()
    for d in scenes:
        n = d["scene"][0]
        z = np.load(tmp_path / "results" / n / f"{n}_bbox_raw.npz")
        assert z["bboxes"].shape[0] == z["scores"].shape[0] > 0
    # module state after a graph replay: what the eager path leaves (reference :247-257, :289-307)
    assert model.valid.dtype == torch.bool and tuple(model.valid.shape) == (1, 1, *dims)
    assert tuple(model.volume.shape)[:2] == (1, 8) and model.points_detection[0].shape[1] == 3 + 8


def test_a_deleted_detector_frees_its_graphs(device, tmp_path):
    """the result-writer thread blocks in q.get() between scenes: nothing of a finished scene -- its slot's static buffers, the
    detector itself -- may stay bound in that frame (round 6: a deleted detector kept ~70 GB of graphs alive at the north-star
    shape).  After `del model` the module, its slots and their device memory are gone; the thread ends by itself."""
    import gc
    import threading
    import weakref
    from cnrma_amd import synth
    dims = synth.SHAPES["tiny"][4]
    gc.collect()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    model = _model(tmp_path, dims, device, max_points=500000)
    scenes = _tiny_scenes(device, 6)
    with torch.no_grad():
        for d in scenes:
            model(return_loss=False, **d)
    model.flush()
    ctx = next(iter(model._static.values()))
    assert ctx["built"]
    m_ref, s_ref = weakref.ref(model), weakref.ref(ctx["slots"][0])
    writer = model._writer[1]
    held = torch.cuda.memory_allocated() - base
    assert writer.is_alive() and held > 0
    del model, ctx, scenes, d
    gc.collect()
    torch.cuda.empty_cache()
    assert m_ref() is None and s_ref() is None
    # what stays are the process-wide grow-only scratch buffers of cnrma_amd.sparse / rma (workspaces per stream, counters, bounds)
    left = torch.cuda.memory_allocated() - base
    assert left <= 64 << 20 and left < 0.5 * held, (left / 2 ** 20, held / 2 ** 20)
    writer.join(timeout=12.0)                                  # it polls for the detector's death every 5 s
    assert not writer.is_alive() and not [t for t in threading.enumerate() if t is writer]


def test_graphs_never_replay_stale_weights(device, tmp_path):
    """captured graphs hold raw pointers to prepared weight images.  train() / eval() toggles,
    load_state_dict() and in-place weight updates after the capture must all lead to results of the CURRENT weights"""
    from cnrma_amd import synth
    dims = synth.SHAPES["tiny"][4]
    scenes = _tiny_scenes(device, 9)
    fast = _model(tmp_path / "fast", dims, device, max_points=100000)
    slow = _model(tmp_path / "slow", dims, device, max_points=100000, static_test=False)
    slow.load_state_dict(fast.state_dict())
    with torch.no_grad():
        for d in scenes[:4]:
            fast(return_loss=False, **d)
        assert next(iter(fast._static.values()))["built"]
        fast.train()
        assert fast._static == {}                                  # toggling the mode drops the graphs (and flushes)
        fast.eval()
        for d in scenes[:4]:
            fast(return_loss=False, **d)                           # re-calibrated, re-captured
        assert next(iter(fast._static.values()))["built"]
        # in-place update (what an optimiser step does) with graphs alive: the next scene must see the new weights
        for p_ in fast.detection_head.parameters():
            p_.mul_(1.05)
        slow.load_state_dict(fast.state_dict())
        for d in scenes[4:]:
            fast(return_loss=False, **d)
            slow(return_loss=False, **d)
        fast.flush()
    _same_files(tmp_path / "fast", tmp_path / "slow", [d["scene"][0] for d in scenes[4:]])
    # a scene graph used directly refuses to replay after a weight change
    from cnrma_amd import _lib
    ctx = next(iter(fast._static.values()))
    st = ctx["slots"][0]
    with torch.no_grad():
        next(fast.detection_backbone.parameters()).add_(0.0)
    with pytest.raises(_lib.CnrmaError):
        st.run(scenes[0]["features"][0], scenes[0]["projection"][0].cpu(), scenes[0]["tsdf"][0, 0])


def test_scenes_that_outgrow_the_plan_fall_back_and_the_graphs_are_recaptured(device, tmp_path):
    """a size plan that is too small for every scene (margin 0.3): each static scene is re-run eagerly with the right result,
    the outgrown sizes are merged and the graphs re-captured after 4 fall-backs"""
    from cnrma_amd import synth
    dims = synth.SHAPES["tiny"][4]
    scenes = _tiny_scenes(device, 9)
    fast = _model(tmp_path / "fast", dims, device, max_points=100000)
    fast.static_margin = 0.3
    slow = _model(tmp_path / "slow", dims, device, max_points=100000, static_test=False)
    slow.load_state_dict(fast.state_dict())
    with torch.no_grad():
        for d in scenes:
            fast(return_loss=False, **d)
            slow(return_loss=False, **d)
        fast.flush()
    assert fast.static_fallbacks >= 4 and getattr(fast, "static_rebuilds", 0) >= 1
    _same_files(tmp_path / "fast", tmp_path / "slow", [d["scene"][0] for d in scenes])


_CRASH_SCRIPT = """
import os, sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import test_plugin_gpu as T
from pathlib import Path
dev = torch.device("cuda:0")
from cnrma_amd import synth
model = T._model(Path({out!r}), synth.SHAPES["tiny"][4], dev, max_points=100000)
scenes = T._tiny_scenes(dev, {n})
with torch.no_grad():
    for d in scenes:
        model(return_loss=False, **d)
torch.cuda.synchronize()
import time; time.sleep(0.5)          # the writer thread is not waited for: no flush(), no atexit
os._exit(17)
"""


def test_result_files_survive_a_crash(device, tmp_path):
    """fcaf3d_head.py:266-271 writes {scene}_bbox_raw.npz inside forward_test.  Here the file is written by a writer thread as
    soon as the scene's graph has finished: a process that dies after N scenes (os._exit: no flush, no atexit) leaves at
    least N - static_slots complete files"""
    import subprocess
    import sys
    n = 9
    code = _CRASH_SCRIPT.format(root=ROOT, out=str(tmp_path), n=n)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 17, r.stderr[-2000:]
    done = 0
    for i in range(n):
        f = tmp_path / "results" / f"scene{i:04d}_00" / f"scene{i:04d}_00_bbox_raw.npz"
        if f.exists():
            z = np.load(f)                                  # complete: a half-written file would not load (renamed into place)
            assert z["bboxes"].shape[0] == z["scores"].shape[0]
            done += 1
    assert done >= n - 3, done
    assert not list((tmp_path / "results").rglob("*.tmp*"))
