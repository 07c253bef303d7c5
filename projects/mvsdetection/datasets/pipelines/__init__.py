from .fcaf3d_transforms import TransformFeaturesBBoxes, sample_points  # noqa: F401
