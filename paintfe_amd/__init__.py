"""paintfe_amd — MI355X-native raster pixel pipeline behind PaintFE's operator surface.

The product is ``libpfx.so`` (hand-written gfx950 HIP kernels + a C ABI, see ``include/pfx.h``); this package is
the thin host-side mirror of the reference's operator interface used by the tests and the bench harness.
Importing it does not load the library; constructing a ``GpuRenderer`` does, and fails loudly when the library
or the GPU is missing.  There is no CPU fallback anywhere in this package.
"""
from ._lib import LIB_PATH, PfxError, load  # noqa: F401
from .renderer import (ADJUST_OPS, BLEND_MODES, DENSE, FROM_FLAT, IN_PLACE, RHAI_OPS, GpuRenderer, png_decode, script_check)  # noqa: F401

from .project import PfeError, Project  # noqa: F401,E402

__all__ = ["GpuRenderer", "PfxError", "load", "LIB_PATH", "BLEND_MODES", "ADJUST_OPS", "RHAI_OPS", "DENSE", "FROM_FLAT",
           "IN_PLACE", "script_check", "png_decode", "Project"]
