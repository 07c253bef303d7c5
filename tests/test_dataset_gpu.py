"""The whole forward_test contract once: dataset -> loading pipeline -> collate -> registered
detector built from the reference-shaped config -> 2D network -> hot path -> files on disk."""
import os
import runpy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = [48, 48, 24]


def _sample(tmp_path, cfg, device):
    import projects.mvsdetection  # noqa: F401
    from cnrma_amd import synth
    from projects.mvsdetection.core.data_container import collate
    from projects.mvsdetection.registry import DATASETS
    root = str(tmp_path / "data")
    ann = synth.write_scannet_like(root, n_scenes=1, V=4, dims=tuple(DIMS), img_hw=(120, 160))
    d = dict(cfg["data"]["test"], data_root=root, ann_file=ann, num_frames=4)
    d["pipeline"] = [dict(t, voxel_dim=DIMS) if "voxel_dim" in t else dict(t) for t in d["pipeline"]]
    ds = DATASETS.build(d)
    return collate([ds[0]], device)


def test_images_to_raw_boxes_through_the_registered_detector(device, tmp_path):
    from projects.mvsdetection.registry import build_model
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "ray_marching_scannet.py"))
    batch = _sample(tmp_path, cfg, device)
    m = dict(cfg["model"])
    # full 2D network of the config (ResNet-50 FPN + AtlasFPNFeature, random weights); no 3D network: the march runs on the
    # scene's ground-truth TSDF, so the aggregation is well defined without trained weights
    m.update(save_path=str(tmp_path / "results"), voxel_dim_test=DIMS, voxel_dim_train=DIMS, backbone_3d=None, tsdf_head=None)
    torch.manual_seed(0)
    model = build_model(m).to(device).eval()
    model.detection_backbone.init_weights()
    model.detection_head.init_weights()
    seen = {}
    def hook(mod, i, o):
        seen.setdefault("f", o.shape)                                     # (a hook that returns a value would replace the output)
    model.feature_2d.register_forward_hook(hook)
    with torch.no_grad():
        assert model(return_loss=False, **batch) == [{}]
    assert tuple(seen["f"]) == (1, 32, 120, 160)                          # one view at a time (use_batchnorm_test=False)
    assert tuple(model.volume.shape) == (1, 32, *DIMS) and model.valid.dtype == torch.bool
    pts = model.points_detection[0]
    assert pts.shape[1] == 3 + 32 and pts.shape[0] > 1000 and torch.isfinite(pts).all()
    z = np.load(tmp_path / "results" / "scene0000_00" / "scene0000_00_bbox_raw.npz")
    assert z["bboxes"].shape[1] == 6 and z["scores"].shape[1] == 18 and np.isfinite(z["bboxes"]).all()
    # the boxes are in the world frame: the volume's offset was added to the points (ray_marching.py:364)
    off = batch["offset"][0].cpu().numpy()
    ext = np.array(DIMS) * 0.04
    c = z["bboxes"][:, :3]
    assert (c.min(0) > off - 1.0).all() and (c.max(0) < off + ext + 1.0).all()


def test_atlas_reconstruction_detector_writes_the_tsdf(device, tmp_path):
    from projects.mvsdetection.datasets.tsdf import TSDF
    from projects.mvsdetection.registry import build_model
    cfg = runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", "atlas_recon_scannet.py"))
    batch = _sample(tmp_path, cfg, device)
    m = dict(cfg["model"], save_path=str(tmp_path / "recon"), voxel_dim_test=DIMS, voxel_dim_train=DIMS)
    m["backbone_3d"] = dict(m["backbone_3d"], channels=[32, 48, 64, 96], layers_down=[1, 1, 1, 1], layers_up=[1, 1, 1])
    m["tsdf_head"] = dict(m["tsdf_head"], input_channels=[32, 48, 64])
    torch.manual_seed(0)
    model = build_model(m).to(device).eval()
    with torch.no_grad():
        assert model(return_loss=False, **batch) == [{}]
    assert set(model.last_losses) == {"tsdf_loss_016", "tsdf_loss_008", "tsdf_loss_004"}
    t = TSDF.load(str(tmp_path / "recon" / "scene0000_00" / "scene0000_00.npz"))
    assert tuple(t.tsdf_vol.shape) == tuple(DIMS) and float(t.tsdf_vol.abs().max()) <= 1.05
    np.testing.assert_allclose(t.origin.view(-1).numpy(), batch["offset"][0].cpu().view(-1).numpy(), atol=1e-6)
    # training step of the reconstruction stage: losses with gradients into the (unfrozen) 2D network
    model.train()
    out = model.train_step(dict(batch), None)
    out["loss"].backward()
    g = model.fpn.fpn_output2.weight.grad
    assert torch.isfinite(out["loss"]) and g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
