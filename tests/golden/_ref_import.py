"""Container-only helper: import the reference's hot-path Python with stubbed third-party modules.

Used ONLY by ``tests/golden/make_golden.py`` (fixture generation, run in the build container where
``/root/reference`` is mounted). Nothing under ``tests/test_*.py``, ``bench.py`` or ``__graft_entry__``
imports this file: ``/root/reference`` does not exist on the GPU box. No reference source is copied --
the reference modules are imported from where they lie and only their *outputs* are saved as fixtures.

Stubbing recipe follows SURVEY.md section 8(c).
"""
import importlib
import importlib.util
import sys
import types

REF = "/root/reference"


class _Registry:
    def register_module(self, *a, **k):
        if len(a) == 1 and isinstance(a[0], type) and not k:      # bare `@HEADS.register_module`
            return a[0]

        def deco(cls):
            return cls
        return deco


def _identity_decorator(*a, **k):
    def deco(f):
        return f
    return deco


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch.nn as nn

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    _mod("open3d")
    me = _mod("MinkowskiEngine", SparseTensor=_Any, MinkowskiConvolution=_Any, MinkowskiBatchNorm=_Any,
              MinkowskiELU=_Any, MinkowskiReLU=_Any, MinkowskiPruning=_Any, MinkowskiInstanceNorm=_Any,
              MinkowskiMaxPooling=_Any, MinkowskiGenerativeConvolutionTranspose=_Any)
    me.utils = _mod("MinkowskiEngine.utils")
    _mod("MinkowskiEngine.modules")
    _mod("MinkowskiEngine.modules.resnet_block", BasicBlock=_Any, Bottleneck=_Any)
    reg = _Registry()
    _mod("mmdet")
    _mod("mmdet.models", DETECTORS=reg, BACKBONES=reg, HEADS=reg)
    _mod("mmdet.models.builder", build_backbone=_Any(), build_head=_Any(), build_neck=_Any(), HEADS=reg,
         build_loss=lambda cfg: None)
    _mod("mmdet.datasets")
    _mod("mmdet.datasets.builder", PIPELINES=reg)
    _mod("mmdet.core", BaseAssigner=object, reduce_mean=lambda x: x, build_assigner=lambda cfg: None)
    _mod("mmdet.core.bbox")
    _mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=reg)
    _mod("mmdet3d")
    _mod("mmdet3d.core", bbox3d2result=None)
    _mod("mmdet3d.core.bbox", DepthInstance3DBoxes=_Any)
    _mod("mmdet3d.core.bbox.structures", rotation_3d_in_axis=None)
    _mod("mmdet3d.ops")
    _mod("mmdet3d.ops.pcdet_nms", pcdet_nms_gpu=None, pcdet_nms_normal_gpu=None)
    _mod("mmcv")
    _mod("mmcv.runner", auto_fp16=_identity_decorator, force_fp32=_identity_decorator)
    _mod("mmcv.parallel", DataContainer=_Any)

    class Scale(nn.Module):
        def __init__(self, scale=1.0):
            import torch
            super().__init__()
            self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

        def forward(self, x):
            return x * self.scale

    _mod("mmcv.cnn", Scale=Scale, bias_init_with_prob=lambda p: 0.0)
    _mod("skimage", measure=None)
    _mod("trimesh")
    _mod("cv2")
    # namespace packages pointing into the reference (bypass the package __init__ chain)
    for name, sub in [("projects", "projects"), ("projects.mvsdetection", "projects/mvsdetection"),
                      ("projects.mvsdetection.datasets", "projects/mvsdetection/datasets"),
                      ("projects.mvsdetection.datasets.pipelines", "projects/mvsdetection/datasets/pipelines"),
                      ("projects.mvsdetection.models", "projects/mvsdetection/models")]:
        m = types.ModuleType(name)
        m.__path__ = [f"{REF}/{sub}"]
        sys.modules[name] = m


def load_reference():
    """Returns (ray_marching module, fcaf3d_head module, fcaf3d_transforms module, tsdf module)."""
    for k in [k for k in sys.modules if k == "projects" or k.startswith("projects.")]:
        del sys.modules[k]
    install_stubs()
    rm = importlib.import_module("projects.mvsdetection.models.ray_marching")
    head = importlib.import_module("projects.mvsdetection.models.fcaf3d_head")
    tr = importlib.import_module("projects.mvsdetection.datasets.pipelines.fcaf3d_transforms")
    ts = importlib.import_module("projects.mvsdetection.datasets.tsdf")
    return rm, head, tr, ts


def load_reference_atlas3d():
    """Returns (backbone3d module, atlas_head module) of the reference (dense 3D U-Net + TSDF head)."""
    install_stubs()
    b3 = importlib.import_module("projects.mvsdetection.models.backbone3d")
    ah = importlib.import_module("projects.mvsdetection.models.atlas_head")
    return b3, ah


def load_reference_2d():
    """Returns (fpn module, backbone2d module) of the reference (ResNet-FPN + pyramid-to-one-map head)."""
    install_stubs()
    fpn = importlib.import_module("projects.mvsdetection.models.fpn")
    b2 = importlib.import_module("projects.mvsdetection.models.backbone2d")
    return fpn, b2


def make_raymarching(rm, voxel_dim, voxel_size=0.04, origin=(0.0, 0.0, 0.0), stride=4, rtype="neus",
                     thr=0.05, depth_points=None, max_points=None):
    import torch
    obj = rm.RayMarching.__new__(rm.RayMarching)
    torch.nn.Module.__init__(obj)
    obj.voxel_dim = list(voxel_dim)
    obj.voxel_size = voxel_size
    obj.origin = torch.tensor(origin, dtype=torch.float32).view(1, 3)
    obj.backbone2d_stride = stride
    obj.ray_marching_type = rtype
    obj.neus_threshold = thr
    obj.depth_points = depth_points
    obj.max_points = max_points
    obj.feature_transform = None
    obj.points_detection = []
    obj.volume = 0
    obj.valid = 0
    return obj
