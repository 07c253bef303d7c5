from .atlas_head import AtlasTSDFHead
from .backbone3d import AtlasBackbone3D
from .fcaf3d_backbone import FCAF3DBackbone
from .fcaf3d_head import FCAF3DAssigner, FCAF3DHead
from .ray_marching import RayMarching

__all__ = ["AtlasBackbone3D", "AtlasTSDFHead", "FCAF3DBackbone", "FCAF3DHead", "FCAF3DAssigner", "RayMarching"]
