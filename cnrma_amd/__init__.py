"""``cnrma_amd`` -- the importable name of the product package.

The sources live in the sibling directory ``cn-rma_amd/`` (the name the project layout prescribes; a hyphen is not a
valid Python identifier), so this package simply lists that directory in its module search path:

    import cnrma_amd                # this file
    from cnrma_amd import rma       # -> cn-rma_amd/rma.py
"""
import os as _os

PACKAGE_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cn-rma_amd")
if not _os.path.isdir(PACKAGE_DIR):
    raise ImportError(f"cnrma_amd: source directory {PACKAGE_DIR} is missing")
__path__.append(PACKAGE_DIR)
__version__ = "0.2.0"
