"""Import shim: `import altair_raytracing_amd as isx` -> the package in ./altair-raytracing_amd/."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("altair-raytracing_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
__all__ = list(_pkg.__all__)
