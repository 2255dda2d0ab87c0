"""MI355X-native 3D -> ASCII rasterizer: Python harness side.

The product is the HIP library (csrc/, C ABI in include/ascii_rast.h) and its JavaScript host
(js/).  This package holds the ctypes binding and the Python mirrors of the host logic that the
pytest suite and bench.py use to drive the same C ABI.
"""
from . import binding, geometry, host, scenes  # noqa: F401
from .binding import AsciiRastError, Group, Renderer, load_library, strip_rows  # noqa: F401

__all__ = ["binding", "geometry", "host", "scenes", "Renderer", "Group", "AsciiRastError", "load_library", "strip_rows"]
