"""mmdetection3d plugin surface of CN-RMA on MI355X.  Importing this package registers the hot-path classes
(reference: projects/mvsdetection/__init__.py:2-23 -- the reference's test.py:18 / train.py:33 import it
unconditionally).  The 2D / Atlas-3D networks and the dataset classes of the reference are outside the hot-path
scope and are not re-implemented here."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:          # `cnrma_amd` (shim for the cn-rma_amd/ directory) lives at the repo root
    sys.path.insert(0, _ROOT)

from .datasets.pipelines import TransformFeaturesBBoxes  # noqa: E402,F401
from .models.fcaf3d_backbone import FCAF3DBackbone       # noqa: E402,F401
from .models.fcaf3d_head import FCAF3DAssigner, FCAF3DHead  # noqa: E402,F401
from .models.ray_marching import RayMarching             # noqa: E402,F401
