"""tools/lab/vm_rate.py — warm kernel time of the script VM (k_script.hip) at 8K for closures of growing length: per-pixel and per-instruction cost"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
img = np.random.default_rng(1).integers(0, 256, size=(4320, 7680, 4), dtype=np.uint8)
cases = [("[r, g, b, a]", "identity"),
         ("[255 - r, g, b, a]", "1 op + 1 literal"),
         ("[255 - r, g / 2, (b * 3 + a) / 4, a]", "bench_ops' closure: 6 ops + 4 literals"),
         ("[(r * 3 + g * 5 + b * 7) / 15, (r + g + b) / 3, (r * r + 1) / 256, a]", "12 ops + literals"),
         ("[clamp((r - 128) * 2 + 128, 0, 255), clamp((g - 128) * 2 + 128, 0, 255), clamp((b - 128) * 2 + 128, 0, 255), a]", "contrast: 12 ops, many literals")]
for body, what in cases:
    src = f"map_channels(|r, g, b, a| {body});"
    r.execute_script_sync(src, img)
    ts = []
    for _ in range(4):
        r.timing_reset(); r.timing_enable(True)
        r.execute_script_sync(src, img)
        r.timing_enable(False)
        ts.append(r.timing_read("script_vm")[0])
    print(f"{what:42s} {sorted(ts)[1]:.3f} ms   {body}")
