"""ctypes mirror of the pfx_group_* C ABI (include/pfx.h): one document across the GPUs of a node, one process.

Binding only — band geometry, peer copies and kernels live in libpfx.so (paintfe_amd/csrc/pfx_group.cpp)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib as L


def band_rows(h: int, world: int, rank: int) -> Tuple[int, int]:
    lib = L.load()
    y0, y1 = C.c_uint32(), C.c_uint32()
    lib.pfx_band_rows(C.c_uint32(h), C.c_uint32(world), C.c_uint32(rank), C.byref(y0), C.byref(y1))
    return int(y0.value), int(y1.value)


class GpuGroup:
    def __init__(self, devices: Sequence[int]):
        self.lib = L.load()
        self.lib.pfx_group_last_error.restype = C.c_char_p
        self.lib.pfx_group_last_error.argtypes = [C.c_void_p]
        for name in ("pfx_group_layer_band_dev", "pfx_group_result_band_dev", "pfx_group_gathered_dev", "pfx_group_ctx"):
            getattr(self.lib, name).restype = C.c_void_p
        self.g = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        st = self.lib.pfx_group_create(arr, C.c_uint32(len(devices)), C.byref(self.g))
        if st != L.OK:
            raise L.PfxError(st, "pfx_group_create failed")
        self.n = len(devices)
        self.w = self.h = 0

    def _check(self, st: int):
        if st != L.OK:
            raise L.PfxError(st, (self.lib.pfx_group_last_error(self.g) or b"").decode())

    def close(self):
        if self.g:
            self.lib.pfx_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_document(self, w: int, h: int, n_layers: int):
        self._check(self.lib.pfx_group_set_document(self.g, C.c_uint32(w), C.c_uint32(h), C.c_uint32(n_layers)))
        self.w, self.h = w, h

    def band(self, rank: int) -> Tuple[int, int]:
        y0, y1 = C.c_uint32(), C.c_uint32()
        self._check(self.lib.pfx_group_band(self.g, C.c_uint32(rank), C.byref(y0), C.byref(y1)))
        return int(y0.value), int(y1.value)

    def upload_layer(self, index: int, rgba: np.ndarray):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert a.shape == (self.h, self.w, 4)
        self._check(self.lib.pfx_group_upload_layer(self.g, C.c_uint32(index), a.ctypes.data_as(C.c_void_p)))

    def layer_band_ptr(self, rank: int, index: int) -> int:
        return int(self.lib.pfx_group_layer_band_dev(self.g, C.c_uint32(rank), C.c_uint32(index)) or 0)

    def flatten_blur(self, infos: List[Tuple[int, float, bool, int]], sigma: float, all_gather: bool = False):
        arr = (L.LayerInfo * len(infos))()
        for k, (idx, op, vis, mode) in enumerate(infos):
            arr[k].layer_idx, arr[k].opacity, arr[k].visible, arr[k].blend_mode, arr[k].kind = idx, op, 1 if vis else 0, mode, 0
        self._check(self.lib.pfx_group_flatten_blur(self.g, arr, C.c_uint32(len(infos)), C.c_float(sigma), C.c_int(1 if all_gather else 0)))

    BAND_NONE, BAND_GAUSSIAN, BAND_BOX, BAND_MEDIAN = 0, 1, 2, 3
    PEER, RCCL, STAGED = 0, 1, 2

    def flatten_filter(self, infos: List[Tuple[int, float, bool, int]], filt: int, param: float, all_gather: bool = False):
        """pfx_group_flatten_filter: flatten, then a band filter (BAND_GAUSSIAN sigma | BAND_BOX radius | BAND_MEDIAN radius)"""
        arr = (L.LayerInfo * len(infos))()
        for k, (idx, op, vis, mode) in enumerate(infos):
            arr[k].layer_idx, arr[k].opacity, arr[k].visible, arr[k].blend_mode, arr[k].kind = idx, op, 1 if vis else 0, mode, 0
        self._check(self.lib.pfx_group_flatten_filter(self.g, arr, C.c_uint32(len(infos)), C.c_int(filt), C.c_float(param), C.c_int(1 if all_gather else 0)))

    def _infos(self, infos):
        arr = (L.LayerInfo * len(infos))()
        for k, (idx, op, vis, mode) in enumerate(infos):
            arr[k].layer_idx, arr[k].opacity, arr[k].visible, arr[k].blend_mode, arr[k].kind = idx, op, 1 if vis else 0, mode, 0
        return arr

    def flatten_warp_displacement(self, infos, disp: np.ndarray):
        """pfx_group_flatten_warp_displacement: flatten, replicate the flattened image, every member warps its band; disp = (h, w, 2) float32"""
        d = np.ascontiguousarray(disp, dtype=np.float32)
        assert d.shape == (self.h, self.w, 2)
        self._check(self.lib.pfx_group_flatten_warp_displacement(self.g, self._infos(infos), C.c_uint32(len(infos)), d.ctypes.data_as(C.c_void_p)))

    def flatten_warp_mesh(self, infos, orig_pts, deformed_pts, cols: int, rows: int):
        """pfx_group_flatten_warp_mesh: orig_pts may be None (uniform original grid); points are (rows + 1, cols + 1, 2) float32"""
        dp = np.ascontiguousarray(deformed_pts, dtype=np.float32)
        op = None if orig_pts is None else np.ascontiguousarray(orig_pts, dtype=np.float32)
        self._check(self.lib.pfx_group_flatten_warp_mesh(self.g, self._infos(infos), C.c_uint32(len(infos)), None if op is None else op.ctypes.data_as(C.c_void_p),
                                                         dp.ctypes.data_as(C.c_void_p), C.c_uint32(cols), C.c_uint32(rows)))

    def set_watchdog(self, timeout_ms: int, calls: int = 0xFFFFFFFF):
        self._check(self.lib.pfx_group_set_watchdog(self.g, C.c_uint32(timeout_ms), C.c_uint32(calls)))

    def synchronize_timeout(self, timeout_ms: int):
        self._check(self.lib.pfx_group_synchronize_timeout(self.g, C.c_uint32(timeout_ms)))

    def set_exact(self, on: bool):
        """pfx_ctx_set_exact on every member's context (the bit-exact Gaussian)"""
        for k in range(self.n):
            self._check(self.lib.pfx_ctx_set_exact(C.c_void_p(self.ctx(k)), C.c_int(1 if on else 0)))

    def set_phase_timing(self, on: bool):
        self._check(self.lib.pfx_group_set_phase_timing(self.g, C.c_int(1 if on else 0)))

    def phase_ms(self, rank: int):
        """(flatten, halo wait, filter, gather) of the last pipeline call on member `rank`, in ms (pfx_group_phase_ms)"""
        out = (C.c_double * 4)()
        self._check(self.lib.pfx_group_phase_ms(self.g, C.c_uint32(rank), out))
        return {"flatten": out[0], "halo_wait": out[1], "filter": out[2], "gather": out[3]}

    def set_transport(self, transport: int):
        self._check(self.lib.pfx_group_set_transport(self.g, C.c_int(transport)))

    def transport(self) -> int:
        return int(self.lib.pfx_group_transport(self.g))

    def synchronize(self):
        self._check(self.lib.pfx_group_synchronize(self.g))

    def result_band_ptr(self, rank: int) -> int:
        return int(self.lib.pfx_group_result_band_dev(self.g, C.c_uint32(rank)) or 0)

    def gathered_ptr(self, rank: int) -> int:
        return int(self.lib.pfx_group_gathered_dev(self.g, C.c_uint32(rank)) or 0)

    def ctx(self, rank: int) -> int:
        return int(self.lib.pfx_group_ctx(self.g, C.c_uint32(rank)) or 0)

    def download(self) -> np.ndarray:
        out = np.empty((self.h, self.w, 4), np.uint8)
        self._check(self.lib.pfx_group_download(self.g, out.ctypes.data_as(C.c_void_p)))
        return out

    def download_gathered(self, rank: int) -> np.ndarray:
        """the all-gathered image as member `rank` holds it"""
        out = np.empty((self.h, self.w, 4), np.uint8)
        self.synchronize()
        st = self.lib.pfx_dev_download(C.c_void_p(self.ctx(rank)), out.ctypes.data_as(C.c_void_p), C.c_void_p(self.gathered_ptr(rank)),
                                       C.c_size_t(out.nbytes))
        if st != L.OK:
            raise L.PfxError(st, "download of the gathered image failed")
        return out
