"""ctypes binding of libpfx.so (the C ABI declared in include/pfx.h).

This is the reference-side binding a maintainer would write in Rust as an ``extern "C"`` block;
here it is Python because the tests and the bench harness are.  It is a *binding only*: no pixel arithmetic
happens in this package, and it fails loudly when the HIP library is missing — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFX_LIB_PATH: development override for A/B timing of two builds of the same ABI (never a fallback: the file must exist)
LIB_PATH = os.environ.get("PFX_LIB_PATH") or os.path.join(_HERE, "libpfx.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_SCRIPT = 0, -1, -2, -3, -4, -5, -6
STATUS_NAMES = {0: "PFX_OK", -1: "PFX_ERR_INVALID", -2: "PFX_ERR_NO_DEVICE", -3: "PFX_ERR_HIP", -4: "PFX_ERR_OOM",
                -5: "PFX_ERR_UNSUPPORTED", -6: "PFX_ERR_SCRIPT"}


class PfxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class LayerInfo(C.Structure):
    _fields_ = [("layer_idx", C.c_uint32), ("opacity", C.c_float), ("visible", C.c_uint8), ("blend_mode", C.c_uint8),
                ("kind", C.c_uint8), ("_pad", C.c_uint8), ("adj", C.c_float * 16)]


class Brush(C.Structure):
    _fields_ = [("size", C.c_float), ("hardness", C.c_float), ("flow", C.c_float), ("color", C.c_float * 4),
                ("anti_aliased", C.c_int32), ("is_eraser", C.c_int32), ("mode", C.c_int32)]


class ScriptResult(C.Structure):
    _fields_ = [("error", C.c_char * 512), ("error_line", C.c_int32), ("error_col", C.c_int32),
                ("ops_executed", C.c_uint32), ("console", C.c_char * 2048)]


class BrushDynamics(C.Structure):
    _fields_ = [("scatter", C.c_float), ("hue_jitter", C.c_float), ("brightness_jitter", C.c_float), ("stamp_counter", C.c_uint32),
                ("tip_mask", C.c_void_p), ("tip_mask_size", C.c_uint32), ("tip_rotation", C.c_float), ("tip_random_rotation", C.c_int32),
                ("tip_rotation_lo", C.c_float), ("tip_rotation_hi", C.c_float)]


class DispDab(C.Structure):
    _fields_ = [("mode", C.c_int32), ("cx", C.c_float), ("cy", C.c_float), ("delta_x", C.c_float), ("delta_y", C.c_float), ("radius", C.c_float),
                ("strength", C.c_float)]


class Preview(C.Structure):
    _fields_ = [("active_layer", C.c_uint32), ("blend_mode", C.c_uint8), ("is_eraser", C.c_uint8), ("replaces_layer", C.c_uint8), ("_pad", C.c_uint8)]


class ChainOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("op", C.c_int32), ("n_params", C.c_uint32), ("params", C.c_float * 12), ("lut", C.c_void_p)]


class CanvasOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("w", C.c_uint32), ("h", C.c_uint32), ("anchor_x", C.c_uint32), ("anchor_y", C.c_uint32)]


_lib = None


def load() -> C.CDLL:
    """Load libpfx.so or raise.  Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C paintfe_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.pfx_last_error.restype = C.c_char_p
    lib.pfx_last_error.argtypes = [C.c_void_p]
    lib.pfx_ctx_stream.restype = C.c_void_p
    lib.pfx_ctx_stream.argtypes = [C.c_void_p]
    lib.pfx_layer_memory.restype = C.c_size_t
    lib.pfx_layer_count.restype = C.c_uint32
    _lib = lib
    return lib
