from .atlas_transforms import (AtlasCollectData, AtlasIntrinsicsPoseToProjection, AtlasRandomTransformSpaceRecon,  # noqa: F401
                               AtlasResizeImage, AtlasTestTransformSpaceRecon, AtlasToTensor)
from .fcaf3d_transforms import AtlasTransformSpaceDetection, TransformFeaturesBBoxes, sample_points  # noqa: F401
