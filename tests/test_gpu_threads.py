"""include/pfx.h: "a context is not thread-safe; distinct contexts are independent" — the second half, tested.  The reference runs its filters on worker threads
while the UI thread composites (src/app.rs spawn_filter_job / src/ops/scripting.rs:1733 on a script thread), so a drop-in is called from several threads at once, each
with its own renderer.  Four host threads, one context each, run different mixes of the bank at the same time (ctypes drops the GIL for the duration of a call);
every result must equal the one the same call gives when nothing else is running — which tests/test_gpu_parity.py pins to the oracle.  What this exercises in the
library: the per-thread device binding, the process-wide dynamic-LDS grant table (k_common.h grant_lds_for), first-use initialisation racing from several threads,
the script runtime's own evaluation threads."""
import threading

import numpy as np
import pytest

from . import inputs as I

pytestmark = pytest.mark.gpu

SCRIPT = """
apply_invert();
let k = 3;
map_channels(|r, g, b, a| [255 - r, (g * k) % 256, b, a]);
apply_brightness_contrast(10.0, 5.0);
print_line(`${width()}x${height()}`);
"""


def layer_stack(seed, w, h, n):
    rng = np.random.default_rng(seed)
    layers = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for _ in range(n)]
    info = [(k, float(rng.uniform(0.2, 1.0)), True, int(rng.integers(0, 25))) for k in range(n)]
    return layers, info


def workload(r, seed):
    """one thread's calls; returns the list of results (arrays / lists), deterministic in (seed)"""
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(200, 700)), int(rng.integers(200, 700))
    img = I.random_rgba(w, h, seed)
    out = []
    out.append(r.gaussian_blur_core(img, float(rng.uniform(0.6, 18.0))))
    out.append(r.adjust(img, "hsl", (float(rng.uniform(-90, 90)), 20.0, -5.0)))
    out.append(r.median_core(img, int(rng.integers(1, 5))))
    out.append(r.box_blur_core(img, float(rng.integers(1, 30))))
    out.append(r.sharpen_core(img, 1.0, float(rng.uniform(0.5, 3.0))))
    out.append(r.glow_core(img, 2.0, 0.5))
    out.append(r.resize_image(img, w // 2 + 3, h // 2 + 1, "lanczos3"))
    layers, info = layer_stack(seed + 100, w, h, 5)
    for k, l in enumerate(layers):
        r.ensure_layer_texture(k, l, 1)
    out.append(r.composite(w, h, info))
    r.clear_layers()
    px, console = r.execute_script_sync(SCRIPT, img)
    out.append(px)
    out.append(np.frombuffer("\n".join(console).encode(), dtype=np.uint8))
    return out


def test_four_contexts_on_four_threads_give_the_serial_results():
    from paintfe_amd import GpuRenderer
    seeds = [11, 12, 13, 14]
    serial = []
    for s in seeds:                      # the answers, one context at a time
        r = GpuRenderer(0)
        serial.append(workload(r, s))
        r.close()
    for attempt in range(3):             # fresh contexts each round: first-use paths race again
        got, errors = [None] * len(seeds), []
        start = threading.Barrier(len(seeds))

        def body(k):
            try:
                r = GpuRenderer(0)
                start.wait()
                res = None
                for _ in range(4):
                    res = workload(r, seeds[k])
                got[k] = res
                r.close()
            except Exception as e:       # noqa: BLE001 — reported below with the thread's number
                errors.append((k, repr(e)))

        threads = [threading.Thread(target=body, args=(k,)) for k in range(len(seeds))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for k in range(len(seeds)):
            assert len(got[k]) == len(serial[k])
            for j, (a, b) in enumerate(zip(got[k], serial[k])):
                assert a.shape == b.shape and np.array_equal(a, b), f"round {attempt}, thread {k}, call {j}: differs from the serial run"
