#!/usr/bin/env python3
"""Convert the reference's committed golden PNGs into one raw-RGBA fixture file.

Run HERE (build container, where /root/reference exists); the GPU box only ever
reads the resulting ``golden.npz``.  The PNGs are *data files held by the
reference's own tests* (``tests/golden/<category>/<name>.png``, written by
``assert_golden`` in ``tests/common/mod.rs:211-263``); nothing but pixels is
copied.  Every entry is a ``(h, w, 4) uint8`` array keyed ``"<category>/<name>"``.

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("PFX_REFERENCE", "/root/reference")
CATEGORIES = ["blend", "filters", "adjustments", "scripting", "transform", "transforms", "tools"]


def main() -> int:
    src = os.path.join(REF, "tests", "golden")
    out = {}
    for cat in CATEGORIES:
        d = os.path.join(src, cat)
        for fn in sorted(os.listdir(d)):
            if not fn.endswith(".png"):
                continue
            img = Image.open(os.path.join(d, fn)).convert("RGBA")
            out[f"{cat}/{fn[:-4]}"] = np.asarray(img, dtype=np.uint8).copy()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz")
    np.savez_compressed(dst, **out)
    print(f"wrote {dst}: {len(out)} images, {os.path.getsize(dst)} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
