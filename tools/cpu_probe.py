#!/usr/bin/env python3
"""tools/cpu_probe.py — how many host cores the box really gives the CPU baseline: affinity, cgroup quota and the
oracle compositor's thread scaling on a 1920x1080x32 stack."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O  # noqa: E402

print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
w, h = 1920, 1080
rng = np.random.default_rng(0)
st = rng.integers(0, 256, (32, h, w, 4), dtype=np.uint8)
st[0, ..., 3] = 255
modes = (np.arange(32) % 25).astype(np.uint8)
op = np.ones(32, np.float32)
op[1::2] = 0.6
for th in (1, 2, 4, 8, 16, 32, 0):
    t = time.perf_counter()
    O.flatten_stack(st, modes, op, threads=th)
    dt = time.perf_counter() - t
    print(f"threads={th}: {dt:.2f} s, {w * h / dt / 1e6:.2f} Mpx/s")
