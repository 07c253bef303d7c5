"""Registries of the plugin.  With mmdet / mmdet3d installed the classes register into THEIR registries (so the
reference's test.py / train.py build them through mmdet3d.models.build_model unchanged); without them (this
image has no mmcv/mmdet) a minimal registry with the same register_module()/build() surface is used.
Reference: projects/mvsdetection/__init__.py:2-23 and the decorators in each model file."""
import inspect


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        cfg = dict(cfg)
        typ = cfg.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        if not inspect.isclass(cls):
            return cls(**cfg)
        return cls(**cfg)


try:  # real mmdetection stack
    from mmdet.models import DETECTORS, BACKBONES, HEADS            # noqa: F401
    from mmdet.models.builder import build_backbone, build_head      # noqa: F401
    from mmdet.core.bbox.builder import BBOX_ASSIGNERS               # noqa: F401
    from mmdet.core import build_assigner                            # noqa: F401
    from mmdet.datasets.builder import PIPELINES, DATASETS           # noqa: F401
    HAVE_MMDET = True
except Exception:  # shim
    DETECTORS, BACKBONES, HEADS = _Registry("detector"), _Registry("backbone"), _Registry("head")
    BBOX_ASSIGNERS, PIPELINES, DATASETS = _Registry("bbox_assigner"), _Registry("pipeline"), _Registry("dataset")
    HAVE_MMDET = False

    def build_backbone(cfg):
        return BACKBONES.build(cfg)

    def build_head(cfg):
        return HEADS.build(cfg)

    def build_assigner(cfg):
        return BBOX_ASSIGNERS.build(cfg)


def build_model(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d.models.build_model equivalent for the shim registries."""
    return DETECTORS.build(cfg, train_cfg=train_cfg, test_cfg=test_cfg) if not HAVE_MMDET else \
        __import__("mmdet3d.models", fromlist=["build_model"]).build_model(cfg, train_cfg=train_cfg, test_cfg=test_cfg)
