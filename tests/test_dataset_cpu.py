"""Dataset + loading pipeline (second half) on a synthetic scene directory in the reference's
ScanNet layout: the sample that reaches the detector has the reference's keys, shapes and geometry."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def dataset_root(tmp_path_factory):
    from cnrma_amd import synth
    root = str(tmp_path_factory.mktemp("scannet_like"))
    ann = synth.write_scannet_like(root, n_scenes=2, V=6, dims=(48, 48, 24), img_hw=(120, 160))
    return root, ann


def _dataset(root, ann, pipeline, test_mode=True, frames=4):
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import DATASETS
    return DATASETS.build(dict(type="AtlasScanNetDataset", data_root=root, ann_file=ann, classes=["a", "b", "c", "d"],
                               pipeline=pipeline, test_mode=test_mode, num_frames=frames, voxel_size=0.04, select_type="unit"))


def test_detection_pipeline_sample_contract(dataset_root):
    import runpy
    root, ann = dataset_root
    cfg = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "projects", "configs",
                                      "mvsdetection", "ray_marching_scannet.py"))
    pipe = [dict(t) for t in cfg["data"]["test"]["pipeline"]]
    pipe[2] = dict(pipe[2], voxel_dim=[48, 48, 24])
    ds = _dataset(root, ann, pipe)
    assert len(ds) == 2
    s = ds[1]
    assert set(s) == {"imgs", "projection", "tsdf_dict", "scene", "image_ids", "offset", "gt_bboxes_3d", "gt_labels_3d",
                      "axis_align_matrix"}
    imgs, proj = s["imgs"].data, s["projection"].data
    assert tuple(imgs.shape) == (4, 3, 480, 640) and imgs.dtype == torch.float32 and 0 <= float(imgs.min()) and float(imgs.max()) <= 255
    assert tuple(proj.shape) == (4, 3, 4)
    vols = s["tsdf_dict"].data
    assert [tuple(vols[k].tsdf_vol.shape) for k in ("tsdf_gt_004", "tsdf_gt_008", "tsdf_gt_016")] == [(48, 48, 24), (24, 24, 12), (12, 12, 6)]
    assert s["scene"].data == "scene0001_00" and s["image_ids"].data == [0, 1, 2, 3]
    # test mode 'origin': the volume starts floor(0.5 / 0.04) = 12 voxels before the scene's TSDF origin (the margin the
    # training crops leave); offset = position of the volume's voxel (0,0,0) in the world frame
    origin = torch.tensor([0.2 - 0.4, 0.3, -0.1])
    np.testing.assert_allclose(s["offset"].data.view(-1).numpy(), (origin - 0.48).numpy(), atol=1e-5)
    # geometry: the images were resized 4x, so the projection is the synthetic one with its intrinsic rows scaled by 4,
    # expressed in the shifted frame: a point of the grid frame projects like in the generator
    from cnrma_amd import synth
    P0, K, poses = synth.camera_projections(6, (48, 48, 24), img_hw=(120, 160), return_parts=True)
    pt = torch.tensor([1.0, 0.9, 0.6, 1.0])                 # in the volume's frame = generator frame + 0.48 m
    ref = P0[0] @ (pt - torch.tensor([0.48, 0.48, 0.48, 0.0]))
    got = proj[0] @ pt
    np.testing.assert_allclose((got[:2] / got[2]).numpy(), 4.0 * (ref[:2] / ref[2]).numpy(), rtol=1e-4)
    # resampled TSDF == the generator's volume (pure translation by whole... the origin is a multiple of nothing: nearest
    # / trilinear resampling of an identical lattice returns the samples themselves inside the truncation band)
    full = synth.room_tsdf((48, 48, 24), boxes=2, seed=1)[0, 0]
    got_vol = vols["tsdf_gt_004"].tsdf_vol[12:, 12:, 12:]            # voxel i of the new volume = voxel i - 12 of the file
    want = full[:36, :36, :12]
    # the reference normalises the sampling grid for align_corners=True but samples with align_corners=False
    # (datasets/tsdf.py:150-166): the volume is read half a voxel low and stretched by n/(n-1) -- kept, so values are
    # interpolated neighbours of the file's, not copies
    band = want.abs() < 0.99
    assert float((got_vol[band] - want[band]).abs().mean()) < 0.15
    assert float(((got_vol[band] > 0) == (want[band] > 0)).float().mean()) > 0.95
    assert bool((vols["tsdf_gt_004"].tsdf_vol[:11] == 1).all())       # outside the scene's volume: empty (+1)
    boxes = s["gt_bboxes_3d"].data
    assert len(boxes) == 2 and tuple(s["gt_labels_3d"].data.tolist()) == (1, 3)


def test_train_pipeline_moves_boxes_with_the_scene(dataset_root):
    import runpy
    root, ann = dataset_root
    cfg = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "projects", "configs",
                                      "mvsdetection", "ray_marching_scannet.py"))
    pipe = [dict(t) for t in cfg["data"]["train"]["pipeline"]]
    pipe[2] = dict(pipe[2], voxel_dim=[48, 48, 24])
    ds = _dataset(root, ann, pipe, test_mode=False)
    s = ds[0]
    # mode 'middle' with a volume of exactly the scene's size: translation = -origin, boxes land in the grid frame
    ext = np.array([48, 48, 24], dtype=np.float32) * 0.04
    centre = s["gt_bboxes_3d"].data.gravity_center[0].numpy()
    np.testing.assert_allclose(centre, [0.35 * ext[0], 0.4 * ext[1], 0.3 * ext[2]], atol=1e-4)
    assert torch.equal(s["offset"].data, torch.zeros(3))


def test_recon_pipelines_and_arkit_helpers(dataset_root):
    root, ann = dataset_root
    pipe = [dict(type="AtlasResizeImage", size=(160, 120)), dict(type="AtlasToTensor"),
            dict(type="AtlasRandomTransformSpaceRecon", voxel_dim=[32, 32, 16], random_rotation=True, random_translation=True,
                 paddingXY=0.2, paddingZ=0.1),
            dict(type="AtlasIntrinsicsPoseToProjection"), dict(type="AtlasCollectData")]
    torch.manual_seed(0)
    s = _dataset(root, ann, pipe, test_mode=False)[0]
    assert tuple(s["tsdf_dict"].data["tsdf_gt_004"].tsdf_vol.shape) == (32, 32, 16) and tuple(s["offset"].data.shape) == (3,)
    pipe[2] = dict(type="AtlasTestTransformSpaceRecon", voxel_dim=[48, 48, 24], origin=[0, 0, 0])
    s = _dataset(root, ann, pipe)[0]
    assert tuple(s["tsdf_dict"].data["tsdf_gt_008"].tsdf_vol.shape) == (24, 24, 12)
    # ARKit trajectory parsing: axis-angle -> matrix against scipy
    from scipy.spatial.transform import Rotation
    from projects.mvsdetection.datasets.arkit_dataset import rodrigues, traj_line_to_pose
    v = np.array([0.3, -1.1, 0.7])
    np.testing.assert_allclose(rodrigues(v), Rotation.from_rotvec(v).as_matrix(), atol=1e-12)
    ts, pose = traj_line_to_pose("123.456 0.3 -1.1 0.7 1.0 2.0 3.0")
    w2c = np.eye(4); w2c[:3, :3] = Rotation.from_rotvec(v).as_matrix(); w2c[:3, 3] = [1, 2, 3]
    np.testing.assert_allclose(pose @ w2c, np.eye(4), atol=1e-12)


def test_tsdf_container_roundtrip(tmp_path):
    from projects.mvsdetection.datasets.tsdf import TSDF
    vol = torch.rand(8, 6, 4) * 2 - 1
    t = TSDF(0.04, torch.tensor([[0.1, 0.2, 0.3]]), vol)
    t.save(str(tmp_path / "t.npz"))
    u = TSDF.load(str(tmp_path / "t.npz"))
    assert u.voxel_size == 0.04 and torch.equal(u.tsdf_vol, vol) and torch.allclose(u.origin, t.origin)
    # a constant volume is a fixed point of the resampling in its interior; the border voxels map to |g| = 1 and count as
    # outside (+1), like every sample beyond the old volume
    c = TSDF(0.04, torch.zeros(1, 3), torch.full((8, 6, 4), 0.25))
    w = c.transform().tsdf_vol
    assert torch.allclose(w[1:-1, 1:-1, 1:-1], torch.full((6, 4, 2), 0.25), atol=1e-6) and bool((w[0] == 1).all())
    shifted = torch.eye(4)
    shifted[0, 3] = 10.0
    assert bool((c.transform(shifted).tsdf_vol == 1).all())
    with pytest.raises(ImportError):
        t.get_mesh()
