from .fcaf3d_backbone import FCAF3DBackbone
from .fcaf3d_head import FCAF3DAssigner, FCAF3DHead
from .ray_marching import RayMarching

__all__ = ["FCAF3DBackbone", "FCAF3DHead", "FCAF3DAssigner", "RayMarching"]
