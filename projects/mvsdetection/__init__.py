"""mmdetection3d plugin surface of CN-RMA on MI355X.  Importing this package registers the hot-path classes
(reference: projects/mvsdetection/__init__.py:2-23 -- the reference's test.py:18 / train.py:33 import it
unconditionally): every name the reference registers is registered here -- the two detectors, the 2D and 3D networks
(plain torch modules: MIOpen executes them on ROCm; they are outside the measured hot path, whose inputs are their
outputs), the datasets and the pipeline transforms."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:          # `cnrma_amd` (shim for the cn-rma_amd/ directory) lives at the repo root
    sys.path.insert(0, _ROOT)

from .datasets import TSDF, AtlasARKitDataset, AtlasScanNetDataset                       # noqa: E402,F401
from .datasets.pipelines import (AtlasCollectData, AtlasIntrinsicsPoseToProjection,              # noqa: E402,F401
                                 AtlasRandomTransformSpaceRecon, AtlasResizeImage, AtlasTestTransformSpaceRecon,
                                 AtlasToTensor, AtlasTransformSpaceDetection, TransformFeaturesBBoxes)
from .models import (Atlas, AtlasBackbone3D, AtlasFPNFeature, AtlasTSDFHead, FCAF3DAssigner,   # noqa: E402,F401
                     FCAF3DBackbone, FCAF3DHead, FPNDetectron, RayMarching, ResNetDetectron)
