#!/usr/bin/env python3
"""tools/lab/clock_probe.py [key=value ...] — what the box reports about clock and power while the compositor runs (amdsmi gpu_metrics, sampled from a
thread), for the bench stack.  Prints the metric keys that exist and the sampled series' statistics; bench.py's sampler (clock_power_sampler) uses the
same calls."""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from paintfe_amd import GpuRenderer

r = GpuRenderer(0); r.set_stream(torch.cuda.current_stream().cuda_stream)
for kv in sys.argv[1:]:
    k, v = kv.split("="); r.tune(k, int(v))
w, h, n = 7680, 4320, 32
dev = torch.device("cuda", 0)
stack, modes, opac = bench.synth_stack(torch, dev, w, h, n, seed=0x5EED0002)
flat = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
ptrs = [stack[k].data_ptr() for k in range(n)]
info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]

import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
h0 = hs[0]
m = amdsmi.amdsmi_get_gpu_metrics_info(h0)
print("gpu_metrics keys:", sorted(m.keys()))
print({k: m[k] for k in m if any(s in k for s in ("gfxclk", "power", "throttle", "temperature_hotspot", "activity", "uclk", "socclk"))})
for fn in ("amdsmi_get_power_info", "amdsmi_get_power_cap_info"):
    try:
        print(fn, getattr(amdsmi, fn)(h0))
    except Exception as e:
        print(fn, "failed", e)
try:
    print("clk gfx", amdsmi.amdsmi_get_clock_info(h0, amdsmi.AmdSmiClkType.GFX))
    print("clk mem", amdsmi.amdsmi_get_clock_info(h0, amdsmi.AmdSmiClkType.MEM))
except Exception as e:
    print("clock_info failed", e)

samples = []
stop = False
def sampler():
    while not stop:
        t = time.perf_counter()
        try:
            mm = amdsmi.amdsmi_get_gpu_metrics_info(h0)
            samples.append((t, mm.get("current_gfxclk"), mm.get("average_gfxclk_frequency"), mm.get("current_socket_power"), mm.get("average_socket_power"),
                            mm.get("current_gfxclks"), mm.get("current_uclk"), mm.get("indep_throttle_status")))
        except Exception as e:
            samples.append((t, str(e)))
        time.sleep(0.002)
th = threading.Thread(target=sampler); th.start()
time.sleep(0.1)
t0 = time.perf_counter()
for _ in range(600): r.flatten_dev(ptrs, info, w, h, flat.data_ptr())
torch.cuda.synchronize()
t1 = time.perf_counter()
time.sleep(0.1)
stop = True; th.join()
print("loop", t1 - t0, "s for 600 launches ->", (t1 - t0) / 600 * 1e3, "ms each;", len(samples), "samples")
busy = [s for s in samples if t0 + 0.15 < s[0] < t1 - 0.02 and len(s) > 2]
idle = [s for s in samples if s[0] < t0 and len(s) > 2]
def col(rows, i):
    v = [x[i] for x in rows if isinstance(x[i], (int, float))]
    return (min(v), sum(v) / len(v), max(v)) if v else None
for name, rows in (("idle", idle), ("busy", busy)):
    print(name, "n", len(rows), "current_gfxclk", col(rows, 1), "avg_gfxclk", col(rows, 2), "cur_power", col(rows, 3), "avg_power", col(rows, 4), "uclk", col(rows, 6))
if busy:
    print("per-xcd gfxclks sample:", busy[len(busy) // 2][5], "throttle:", busy[len(busy) // 2][7])
print("first 12 busy samples:", [(round(s[0] - t0, 4),) + tuple(s[1:5]) for s in busy[:12]])
