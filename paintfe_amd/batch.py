"""ctypes mirror of pfx_batch_pipeline (include/pfx.h): a batch of independent images streamed through the GPUs of a node
(the reference CLI's file loop, src/cli.rs:159-216).  Binding only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence

import numpy as np

from . import _lib as L


class BatchParams(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("sigma", C.c_float), ("hue", C.c_float), ("saturation", C.c_float),
                ("lightness", C.c_float), ("n_overlays", C.c_uint32), ("overlays_host", C.POINTER(C.c_void_p)),
                ("overlay_modes", C.c_void_p), ("overlay_opacity", C.c_void_p), ("slots", C.c_uint32), ("n_keep", C.c_uint32),
                ("keep_indices", C.c_void_p), ("keep_out", C.POINTER(C.c_void_p)), ("out_of_contract_fast_gaussian", C.c_uint32)]


class BatchStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("images_per_s", C.c_double), ("h2d_gbs", C.c_double), ("d2h_gbs", C.c_double),
                ("kernel_ms_per_image", C.c_double), ("images", C.c_uint32), ("devices", C.c_uint32)]


def run_batch(devices: Sequence[int], n_images: int, pool: List[np.ndarray], overlays: List[np.ndarray], overlay_modes: Sequence[int],
              sigma: float, hsl=(30.0, -20.0, 10.0), slots: int = 3, keep: Sequence[int] = (), fast: bool = False) -> Dict:
    """Runs the S4 pipeline on `n_images` images (image i = pool[i % len(pool)]); returns the stats and the kept results.
    fast=False (default): the bit-exact Gaussian, results equal the CPU path; fast=True: the default-mode Gaussian (+-1 LSB before HSL)."""
    lib = L.load()
    h, w = pool[0].shape[:2]
    pool = [np.ascontiguousarray(p, dtype=np.uint8) for p in pool]
    overlays = [np.ascontiguousarray(o, dtype=np.uint8) for o in overlays]
    assert all(p.shape == (h, w, 4) for p in pool + overlays)
    P = BatchParams()
    P.w, P.h, P.sigma = w, h, sigma
    P.hue, P.saturation, P.lightness = hsl
    P.n_overlays = len(overlays)
    ov_ptrs = (C.c_void_p * max(len(overlays), 1))(*[o.ctypes.data for o in overlays])
    P.overlays_host = C.cast(ov_ptrs, C.POINTER(C.c_void_p))
    modes = np.asarray(overlay_modes, np.uint8)
    P.overlay_modes = modes.ctypes.data
    P.overlay_opacity = None
    P.slots = slots
    P.out_of_contract_fast_gaussian = 1 if fast else 0
    keep_idx = np.asarray(list(keep), np.uint32)
    outs = [np.zeros((h, w, 4), np.uint8) for _ in keep_idx]
    out_ptrs = (C.c_void_p * max(len(outs), 1))(*[o.ctypes.data for o in outs])
    P.n_keep = len(outs)
    P.keep_indices = keep_idx.ctypes.data if len(outs) else None
    P.keep_out = C.cast(out_ptrs, C.POINTER(C.c_void_p))
    pool_ptrs = (C.c_void_p * len(pool))(*[p.ctypes.data for p in pool])
    st = BatchStats()
    err = C.create_string_buffer(512)
    devs = (C.c_int * len(devices))(*devices)
    rc = lib.pfx_batch_pipeline(devs, C.c_uint32(len(devices)), C.c_uint32(n_images), C.byref(P), C.cast(pool_ptrs, C.POINTER(C.c_void_p)),
                                C.c_uint32(len(pool)), C.byref(st), err, C.c_size_t(512))
    if rc != L.OK:
        raise L.PfxError(rc, err.value.decode() or "pfx_batch_pipeline failed")
    return {"seconds": st.seconds, "images_per_s": st.images_per_s, "h2d_gbs": st.h2d_gbs, "d2h_gbs": st.d2h_gbs,
            "kernel_ms_per_image": st.kernel_ms_per_image, "images": int(st.images), "devices": int(st.devices),
            "kept": {int(i): o for i, o in zip(keep_idx, outs)}}
