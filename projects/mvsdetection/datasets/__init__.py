from .arkit_dataset import AtlasARKitDataset  # noqa: F401
from .scannet_dataset import AtlasScanNetDataset  # noqa: F401
from .tsdf import TSDF, coordinates  # noqa: F401
