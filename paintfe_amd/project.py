"""PFE project documents (`.pfe`): host-side mirror of `load_pfe` / `save_pfe` and of `CanvasState::composite` on a
loaded document (ref: src/io.rs:242-499, src/cli.rs:222-308).  Binding only — parsing, serialisation and the device work
are libpfx.so's (`paintfe_amd/csrc/pfx_project.cpp`).  Loading and saving need no GPU; compositing and scripts do.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import PfxError

KINDS = ["raster", "exposure", "brightness_contrast", "invert", "channel_mixer"]  # pfx_layer_kind


class _ProjectLayer(C.Structure):
    _fields_ = [("name", C.c_char_p), ("visible", C.c_uint8), ("effectively_visible", C.c_uint8), ("blend_mode", C.c_uint8),
                ("layer_type", C.c_uint8), ("opacity", C.c_float), ("folder_id", C.c_int64), ("n_chunks", C.c_uint32),
                ("kind", C.c_uint8), ("_pad", C.c_uint8 * 3), ("adj", C.c_float * 16)]


def _bind():
    lib = _lib.load()
    if getattr(lib, "_pfx_project_bound", False):
        return lib
    lib.pfx_project_load.restype = C.c_void_p
    lib.pfx_project_load.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.pfx_project_load_file.restype = C.c_void_p
    lib.pfx_project_load_file.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    lib.pfx_project_new.restype = C.c_void_p
    lib.pfx_project_new.argtypes = [C.c_uint32, C.c_uint32]
    lib.pfx_project_free.restype = None
    lib.pfx_project_free.argtypes = [C.c_void_p]
    for f in ("width", "height", "layer_count", "active_layer"):
        getattr(lib, "pfx_project_" + f).restype = C.c_uint32
        getattr(lib, "pfx_project_" + f).argtypes = [C.c_void_p]
    lib.pfx_project_version.argtypes = [C.c_void_p]
    lib.pfx_project_layer_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_ProjectLayer)]
    lib.pfx_project_layer_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.pfx_project_add_layer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_float, C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
    lib.pfx_project_set_active_layer.argtypes = [C.c_void_p, C.c_uint32]
    lib.pfx_project_set_layer_folder.argtypes = [C.c_void_p, C.c_uint32, C.c_int64]
    lib.pfx_project_add_folder.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint8]
    lib.pfx_project_set_layer_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.pfx_project_save.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.pfx_project_save_file.argtypes = [C.c_void_p, C.c_char_p]
    lib.pfx_bytes_free.restype = None
    lib.pfx_bytes_free.argtypes = [C.c_void_p]
    lib.pfx_project_composite.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pfx_project_composite_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pfx_project_run_script.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(_lib.ScriptResult)]
    lib._pfx_project_bound = True
    return lib


class PfeError(ValueError):
    """PfeError of the reference (io.rs:211-240): the message carries its Display text"""


class Project:
    def __init__(self, handle: int):
        self._lib = _bind()
        self._h = handle

    # ---- constructors
    @classmethod
    def load_bytes(cls, raw: bytes) -> "Project":          # load_pfe_from_bytes, io.rs:477
        lib = _bind()
        err = C.create_string_buffer(512)
        buf = (C.c_uint8 * max(len(raw), 1)).from_buffer_copy(raw if raw else b"\0")
        h = lib.pfx_project_load(buf, len(raw), err, 512)
        if not h:
            raise PfeError(err.value.decode("utf-8", "replace"))
        return cls(h)

    @classmethod
    def load(cls, path: str) -> "Project":                 # load_pfe, io.rs:469
        lib = _bind()
        err = C.create_string_buffer(512)
        h = lib.pfx_project_load_file(str(path).encode(), err, 512)
        if not h:
            raise PfeError(err.value.decode("utf-8", "replace"))
        return cls(h)

    @classmethod
    def new(cls, w: int, h: int) -> "Project":
        lib = _bind()
        hd = lib.pfx_project_new(w, h)
        if not hd:
            raise PfeError(f"bad document size {w}x{h}")
        return cls(hd)

    def close(self):
        if self._h:
            self._lib.pfx_project_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accessors
    @property
    def width(self) -> int:
        return self._lib.pfx_project_width(self._h)

    @property
    def height(self) -> int:
        return self._lib.pfx_project_height(self._h)

    @property
    def version(self) -> int:
        return self._lib.pfx_project_version(self._h)

    @property
    def active_layer(self) -> int:
        return self._lib.pfx_project_active_layer(self._h)

    def __len__(self) -> int:
        return self._lib.pfx_project_layer_count(self._h)

    def _ok(self, st: int, what: str):
        if st != 0:
            raise PfxError(st, what)

    def layer(self, i: int) -> dict:
        L = _ProjectLayer()
        self._ok(self._lib.pfx_project_layer_get(self._h, i, C.byref(L)), f"layer {i}")
        return {"name": (L.name or b"").decode("utf-8", "replace"), "visible": bool(L.visible), "effectively_visible": bool(L.effectively_visible),
                "blend_mode": int(L.blend_mode), "layer_type": int(L.layer_type), "opacity": float(L.opacity),
                "folder_id": None if L.folder_id < 0 else int(L.folder_id), "n_chunks": int(L.n_chunks), "kind": int(L.kind),
                "adj": [float(v) for v in L.adj]}

    def layer_pixels(self, i: int) -> np.ndarray:          # Layer::pixels.to_rgba_image()
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._ok(self._lib.pfx_project_layer_pixels(self._h, i, out.ctypes.data_as(C.c_void_p)), f"layer {i}")
        return out

    # ---- editing
    def add_layer(self, name: str, rgba: Optional[np.ndarray], opacity: float = 1.0, blend_mode: int = 0, visible: bool = True,
                  kind=0, adj: Sequence[float] = ()):
        k = KINDS.index(kind) if isinstance(kind, str) else int(kind)
        px = None
        if rgba is not None:
            px = np.ascontiguousarray(rgba, np.uint8)
            assert px.shape == (self.height, self.width, 4), px.shape
        a = (C.c_float * 16)(*(list(adj) + [0.0] * (16 - len(adj))))
        self._ok(self._lib.pfx_project_add_layer(self._h, name.encode(), None if px is None else px.ctypes.data_as(C.c_void_p), opacity, blend_mode,
                                                 1 if visible else 0, k, a), "add_layer")

    def set_active_layer(self, i: int):
        self._ok(self._lib.pfx_project_set_active_layer(self._h, i), "set_active_layer")

    def add_folder(self, folder_id: int, name: str, visible: bool = True):
        self._ok(self._lib.pfx_project_add_folder(self._h, folder_id, name.encode(), 1 if visible else 0), "add_folder")

    def set_layer_folder(self, i: int, folder_id: Optional[int]):
        self._ok(self._lib.pfx_project_set_layer_folder(self._h, i, -1 if folder_id is None else folder_id), "set_layer_folder")

    def set_layer_pixels(self, i: int, rgba: np.ndarray):
        px = np.ascontiguousarray(rgba, np.uint8)
        assert px.shape == (self.height, self.width, 4), px.shape
        self._ok(self._lib.pfx_project_set_layer_pixels(self._h, i, px.ctypes.data_as(C.c_void_p)), "set_layer_pixels")

    # ---- serialisation
    def save_bytes(self) -> bytes:                          # build_pfe + bincode::serialize, io.rs:254-465
        p, n = C.c_void_p(), C.c_size_t()
        self._ok(self._lib.pfx_project_save(self._h, C.byref(p), C.byref(n)), "save")
        try:
            return C.string_at(p, n.value)
        finally:
            self._lib.pfx_bytes_free(p)

    def save(self, path: str):                              # save_pfe, io.rs:242
        self._ok(self._lib.pfx_project_save_file(self._h, str(path).encode()), f"save {path}")

    # ---- device operations (need a GpuRenderer)
    def composite(self, renderer) -> np.ndarray:            # CanvasState::composite(), canvas_state.rs:482
        out = np.zeros((self.height, self.width, 4), np.uint8)
        renderer._check(self._lib.pfx_project_composite(renderer.handle, self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def composite_dev(self, renderer, dst_ptr: int):
        renderer._check(self._lib.pfx_project_composite_dev(renderer.handle, self._h, C.c_void_p(dst_ptr)))

    def run_script(self, renderer, source: str):            # run_one's script step, cli.rs:238-270
        res = _lib.ScriptResult()
        st = self._lib.pfx_project_run_script(renderer.handle, self._h, source.encode(), C.byref(res))
        if st != 0:
            msg = res.error.decode("utf-8", "replace") or self._lib.pfx_last_error(renderer.handle).decode("utf-8", "replace")
            raise PfxError(st, msg)
