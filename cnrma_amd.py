"""Import shim: the product package lives in the directory ``cn-rma_amd/`` (a hyphen is not a valid Python
identifier), so this root-level module turns itself into a package whose submodules are loaded from there.

    import cnrma_amd                # this file
    from cnrma_amd import rma       # -> cn-rma_amd/rma.py
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cn-rma_amd")]
PACKAGE_DIR = __path__[0]

with open(_os.path.join(PACKAGE_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(PACKAGE_DIR, "__init__.py"), "exec"))
