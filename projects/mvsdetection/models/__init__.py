from .atlas import Atlas
from .atlas_head import AtlasTSDFHead
from .backbone2d import AtlasFPNFeature
from .backbone3d import AtlasBackbone3D
from .fcaf3d_backbone import FCAF3DBackbone
from .fcaf3d_head import FCAF3DAssigner, FCAF3DHead
from .fpn import FPNDetectron
from .ray_marching import RayMarching
from .resnet import ResNetDetectron

__all__ = ["AtlasTSDFHead", "Atlas", "AtlasFPNFeature", "AtlasBackbone3D", "FPNDetectron", "ResNetDetectron",
           "FCAF3DBackbone", "FCAF3DHead", "FCAF3DAssigner", "RayMarching"]
