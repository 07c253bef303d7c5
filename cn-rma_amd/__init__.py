"""cn-rma_amd: MI355X-native hot path of CN-RMA (ray-marching aggregation + sparse FCAF3D forward).

Imported as ``cnrma_amd``: the package ``../cnrma_amd/`` lists this directory in its ``__path__``.
Layout:
  csrc/        hand-written HIP (gfx950) kernels + the C-ABI (declared in ../include/cnrma.h)
  _lib.py      ctypes loader of csrc/libcnrma_hip.so (fails loudly when missing)
  rma.py       host mirror of the aggregation functions of ray_marching.py (a1-a8)
  sparse.py    sparse tensor + operators replacing the MinkowskiEngine surface (a9-a11)
  synth.py     deterministic synthetic scene generator
"""
__version__ = "0.2.0"
