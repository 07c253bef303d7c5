"""CPU suite: plugin surface (registries, constructor signatures, state-dict key names, configs) and the multi-GPU
path on a world-size-2 gloo group."""
import importlib.util
import os
import runpy

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(name):
    return runpy.run_path(os.path.join(ROOT, "projects", "configs", "mvsdetection", name))


def test_import_registers_the_reference_names():
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection import registry as R
    if R.HAVE_MMDET:
        pytest.skip("real mmdet registries in use")
    assert R.DETECTORS.get("RayMarching") is not None
    assert R.BACKBONES.get("FCAF3DBackbone") is not None
    assert R.HEADS.get("FCAF3DHead") is not None
    assert R.BBOX_ASSIGNERS.get("FCAF3DAssigner") is not None
    assert R.PIPELINES.get("TransformFeaturesBBoxes") is not None


@pytest.mark.parametrize("name,n_cls,n_reg", [("ray_marching_scannet.py", 18, 6), ("ray_marching_arkit.py", 17, 8)])
def test_configs_build_the_detector(name, n_cls, n_reg, tmp_path):
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    cfg = _cfg(name)
    m = dict(cfg["model"])
    m["save_path"] = str(tmp_path / "results")
    model = build_model(m)
    assert type(model).__name__ == "RayMarching"
    assert model.detection_head.n_classes == n_cls and model.detection_head.n_reg_outs == n_reg
    assert model.max_points == 500000 and model.neus_threshold == 0.05 and model.backbone2d_stride == 4
    for meth in ("forward", "forward_train", "forward_test", "train_step", "val_step", "parse_losses", "data_converter",
                 "init_weights", "aggregate_2d_features", "clear_3d_features", "aggregate_2d_features_ray_marching",
                 "fcaf3d_detection", "switch_pointcloud", "ray_projection_neus", "ray_projection_depth"):
        assert callable(getattr(model, meth))
    keys = set(model.state_dict())
    # MinkowskiEngine / mmcv parameter names of the reference's checkpoints
    for k in ("detection_backbone.conv1.0.kernel", "detection_backbone.conv1.1.weight",
              "detection_backbone.layer1.0.conv1.kernel", "detection_backbone.layer1.0.norm1.bn.running_mean",
              "detection_backbone.layer1.0.downsample.0.kernel", "detection_backbone.layer1.0.downsample.1.bn.weight",
              "detection_backbone.layer4.2.conv2.kernel", "detection_head.up_block_1.0.kernel",
              "detection_head.up_block_3.4.bn.running_var", "detection_head.out_block_0.0.kernel",
              "detection_head.centerness_conv.kernel", "detection_head.reg_conv.kernel",
              "detection_head.cls_conv.kernel", "detection_head.cls_conv.bias", "detection_head.scales.3.scale"):
        assert k in keys, k
    sd = model.state_dict()
    assert tuple(sd["detection_backbone.conv1.0.kernel"].shape) == (27, 32, 64)
    assert tuple(sd["detection_backbone.layer1.0.downsample.0.kernel"].shape) == (64, 64)      # K == 1 -> [Cin, Cout]
    assert tuple(sd["detection_head.up_block_1.0.kernel"].shape) == (8, 128, 64)
    assert tuple(sd["detection_head.cls_conv.bias"].shape) == (1, n_cls)


def test_sample_points_uses_numpy_global_rng_like_reference():
    from projects.mvsdetection.datasets.pipelines.fcaf3d_transforms import sample_points
    from oracle import rma_oracle as O
    pts = torch.zeros(1000, 3)
    np.random.seed(3)
    a = sample_points(pts, max_points=100)
    np.random.seed(3)
    b = O.sample_mask_numpy(1000, 100)
    assert a.dtype == torch.bool and int(a.sum()) == 100 and (a.numpy() == b).all()
    assert bool(sample_points(pts, max_points=5000).all())


def test_point_transform_helpers_match_reference_golden():
    from projects.mvsdetection.datasets.pipelines import fcaf3d_transforms as T
    z = np.load(os.path.join(ROOT, "tests", "golden", "point_transforms.npz"))
    pts = torch.from_numpy(z["points"])
    assert torch.equal(T.rotate_points(pts.clone(), 0.0731), torch.from_numpy(z["rot"]))
    assert torch.equal(T.flip_points(pts.clone(), "horizontal"), torch.from_numpy(z["flip_h"]))
    assert torch.equal(T.flip_points(pts.clone(), "vertical"), torch.from_numpy(z["flip_v"]))
    assert torch.equal(T.scale_points(pts.clone(), 1.0625), torch.from_numpy(z["scale"]))
    assert torch.equal(T.translate_points(pts.clone(), np.array([0.1, -0.05, 0.2], dtype=np.float32)), torch.from_numpy(z["trans"]))
    np.random.seed(21)
    assert (T.sample_points(torch.zeros(1000, 3), max_points=123).numpy() == z["sample_mask_seed21"]).all()


def test_coordinates_order():
    from projects.mvsdetection.datasets.tsdf import coordinates
    c = coordinates((2, 3, 4))
    assert c.shape == (3, 24) and c[:, 1].tolist() == [0, 0, 1] and c[:, 4].tolist() == [0, 1, 0] and c[:, 12].tolist() == [1, 0, 0]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    from cnrma_amd import pipeline
    g = torch.Generator().manual_seed(rank)
    k = 5 + 3 * rank                                   # ragged: every rank has a different number of detections
    boxes, scores = torch.rand(k, 6, generator=g), torch.rand(k, 18, generator=g)
    out = pipeline.gather_detections(boxes, scores)
    ok = len(out) == world
    for r, (b, s) in enumerate(out):
        gr = torch.Generator().manual_seed(r)
        eb, es = torch.rand(5 + 3 * r, 6, generator=gr), torch.rand(5 + 3 * r, 18, generator=gr)
        ok = ok and torch.equal(b, eb) and torch.equal(s, es)
    # the per-step exchange of bench.py: fixed-size padded blocks + live counts, no host read of any count
    det = torch.full((3, 7, 4), float(rank)) + torch.arange(3).view(3, 1, 1)
    val = torch.tensor([[rank, 1], [2, rank], [rank, rank]], dtype=torch.int32)
    det_all, val_all = pipeline.gather_padded_detections(det, val)
    ok = ok and tuple(det_all.shape) == (world, 3, 7, 4) and tuple(val_all.shape) == (world, 3, 2)
    for r in range(world):
        ok = ok and torch.equal(det_all[r], torch.full((3, 7, 4), float(r)) + torch.arange(3).view(3, 1, 1))
        ok = ok and torch.equal(val_all[r], torch.tensor([[r, 1], [2, r], [r, r]], dtype=torch.int32))
    # scene sharding: scene i -> rank i mod W covers every scene exactly once
    mine = [i for i in range(7) if i % world == rank]
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    ok = ok and sorted(sum(allv, [])) == list(range(7))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_detection_all_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_gather_is_identity():
    from cnrma_amd import pipeline
    b, s = torch.rand(4, 6), torch.rand(4, 18)
    out = pipeline.gather_detections(b, s)
    assert len(out) == 1 and out[0][0] is b and out[0][1] is s


def test_every_name_the_reference_registers_is_registered():
    """projects/mvsdetection/__init__.py:2-23 of the reference"""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection import registry as R
    if R.HAVE_MMDET:
        pytest.skip("real mmdet registries in use")
    want = dict(DETECTORS=["RayMarching", "Atlas"],
                BACKBONES=["FCAF3DBackbone", "AtlasBackbone3D", "AtlasFPNFeature", "FPNDetectron", "ResNetDetectron"],
                HEADS=["FCAF3DHead", "AtlasTSDFHead"], BBOX_ASSIGNERS=["FCAF3DAssigner"],
                PIPELINES=["AtlasResizeImage", "AtlasIntrinsicsPoseToProjection", "AtlasRandomTransformSpaceRecon",
                           "AtlasTestTransformSpaceRecon", "AtlasToTensor", "AtlasCollectData", "AtlasTransformSpaceDetection",
                           "TransformFeaturesBBoxes"],
                DATASETS=["AtlasScanNetDataset", "AtlasARKitDataset"])
    for reg, names in want.items():
        for n in names:
            assert getattr(R, reg).get(n) is not None, f"{n} missing from {reg}"
    from projects.mvsdetection import TSDF  # noqa: F401


@pytest.mark.parametrize("name,kind", [("scannet_middle.py", "RayMarching"), ("arkit_middle.py", "RayMarching"),
                                       ("atlas_recon_scannet.py", "Atlas"), ("atlas_recon_arkit.py", "Atlas")])
def test_stage_configs_build(name, kind, tmp_path):
    """the four remaining config files of the reference: full key set, built through the registry"""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    cfg = _cfg(name)
    m = dict(cfg["model"])
    m["save_path"] = str(tmp_path / "results")
    if "middle_save_path" in m:
        m["middle_save_path"] = str(tmp_path / "middle")
    model = build_model(m)
    assert type(model).__name__ == kind
    keys = set(model.state_dict())
    # the 2D / 3D networks carry the reference's checkpoint prefixes
    for k in ("fpn.bottom_up.stem.conv1.weight", "fpn.fpn_lateral5.weight", "fpn.fpn_output2.norm.running_mean",
              "feature_2d.p5.4.norm.weight", "backbone3d.layers_down.0.0.conv1.weight", "tsdf_head.decoders.0.weight"):
        assert k in keys, k
    for split in ("train", "val", "test"):
        assert cfg["data"][split]["type"] in ("AtlasScanNetDataset", "AtlasARKitDataset")
        assert [t["type"] for t in cfg["data"][split]["pipeline"]][0] == "AtlasResizeImage"


def test_save_middle_result_writes_the_pretraining_points(tmp_path):
    """reference ray_marching.py:959-991: {scene}_vert.npy = [M', 3 + C] with the offset added and at most max_points rows"""
    import projects.mvsdetection  # noqa: F401
    from projects.mvsdetection.registry import build_model
    m = dict(_cfg("scannet_middle.py")["model"])
    m.update(save_path=str(tmp_path / "r"), middle_save_path=str(tmp_path / "mid"), backbone2d=None, feature_2d=None,
             backbone_3d=None, tsdf_head=None, max_points=50)
    model = build_model(m)
    pts = torch.arange(200 * 5, dtype=torch.float32).view(200, 5)
    np.random.seed(1)
    model.save_middle_result("sceneX", pts, torch.tensor([[1.0, 2.0, 3.0]]), m["middle_save_path"], str(tmp_path / "vis"))
    out = np.load(tmp_path / "mid" / "sceneX_vert.npy")
    assert out.shape == (50, 5)
    rows = (out[:, 3] - 3) / 5                       # feature column 3 of source row i is 5 i + 3
    assert np.all(np.diff(rows) > 0)                 # order preserved
    np.testing.assert_allclose(out[:, :3], pts.numpy()[rows.astype(int), :3] + [1, 2, 3])
    assert (tmp_path / "vis" / "sceneX" / "sceneX_points.ply").exists()


def test_host_tensors_are_rejected_before_they_reach_a_kernel():
    """a CPU tensor's address handed to a HIP kernel is a GPU memory fault; the binding refuses it instead"""
    import torch
    from cnrma_amd import _lib
    with pytest.raises(_lib.CnrmaError, match="must live on the GPU"):
        _lib.ptr(torch.zeros(4))
    assert _lib.ptr(None) is None


def test_integration_md_ctypes_stub_matches_the_abi():
    """the ctypes stub INTEGRATION.md tells a maintainer to paste was an ABI-v2 one.  The fenced snippet is
    executed here against a recording stand-in of the library: every `argtypes` list it declares must equal the binding
    table (cnrma_amd._lib.SIGNATURES, itself checked against include/cnrma.h), and every call in its function body must
    pass exactly that many arguments"""
    import ast
    import ctypes
    import re
    from cnrma_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "ctypes.CDLL" in b]
    assert len(stub) == 1
    code = stub[0]

    class Fn:
        def __init__(self, name):
            self.name, self.argtypes = name, None

        def __call__(self, *a):
            return _lib.ABI_VERSION if self.name == "cnrma_abi_version" else 0

    class Lib:
        def __init__(self):
            self.fns = {}

        def __getattr__(self, name):
            return self.__dict__["fns"].setdefault(name, Fn(name))
    lib = Lib()
    real_cdll = ctypes.CDLL
    ctypes.CDLL = lambda path: lib
    try:
        exec(compile(code, "INTEGRATION.md", "exec"), {})
    finally:
        ctypes.CDLL = real_cdll
    declared = {n: f.argtypes for n, f in lib.fns.items() if f.argtypes is not None}
    assert {"cnrma_nchw_to_nhwc_f32", "cnrma_backproject_accum_f32"} <= set(declared)
    for name, args in declared.items():
        assert list(args) == list(_lib.SIGNATURES[name][1]), name
    calls = [n for n in ast.walk(ast.parse(code)) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)
             and isinstance(n.func.value, ast.Name) and n.func.value.id == "lib" and n.func.attr.startswith("cnrma_")]
    seen = set()
    for c in calls:
        assert not any(isinstance(a, ast.Starred) for a in c.args)
        assert len(c.args) == len(_lib.SIGNATURES[c.func.attr][1]), c.func.attr
        seen.add(c.func.attr)
    assert {"cnrma_nchw_to_nhwc_f32", "cnrma_backproject_accum_f32", "cnrma_abi_version"} <= seen


@pytest.mark.parametrize("depth,blocks,widths", [(50, (4, 3, 6, 3), (256, 512, 1024, 2048)), (101, (3, 4, 23, 3), (256, 512, 1024, 2048)),
                                                 (34, (3, 4, 6, 3), (64, 128, 256, 512))])
def test_backbone_depths_follow_the_reference_table(depth, blocks, widths):
    """fcaf3d_backbone.py:112-127: BasicBlock for 14 / 18 / 34, ME's Bottleneck (expansion 4) for 50 / 101, ME's parameter names"""
    from projects.mvsdetection.models.fcaf3d_backbone import FCAF3DBackbone
    bb = FCAF3DBackbone(32, depth)
    keys = set(bb.state_dict())
    for i, (n, w) in enumerate(zip(blocks, widths)):
        layer = getattr(bb, f"layer{i + 1}")
        assert len(layer) == n
        last = layer[-1]
        out = last.conv3.out_channels if depth >= 50 else last.conv2.out_channels
        assert out == w
        assert f"layer{i + 1}.0.downsample.0.kernel" in keys and f"layer{i + 1}.0.downsample.1.bn.running_var" in keys
    if depth >= 50:
        assert bb.layer1[0].conv1.kernel.shape == (64, 64) and bb.layer1[0].conv2.kernel.shape == (27, 64, 64)
        assert bb.layer1[0].conv3.kernel.shape == (64, 256) and bb.layer1[1].conv1.kernel.shape == (256, 64)
        assert {"layer1.0.conv3.kernel", "layer1.0.norm3.bn.weight", "layer3.5.norm2.bn.running_mean"} <= keys
    with pytest.raises(ValueError):
        FCAF3DBackbone(32, 20)
