#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/prof.sh) into a short per-kernel table: average duration from the kernel
trace and per-launch PMC counter means for our kernels.  FETCH_SIZE is doubled for wide coalesced reads as
guides/MI355X_MICROARCH.md §HBM prescribes for gfx950 (it tallies 128-B requests at 64 B)."""
import csv
import glob
import os
import sys
from collections import defaultdict

OURS = ("flatten_srt_kernel", "flatten_dle_kernel", "flatten_stream_kernel", "flatten_kernel", "gauss_strip64_kernel", "gauss_strip_kernel", "gauss_fused_exact_kernel", "pointwise_chain_kernel", "gauss_planarize_kernel", "gauss_mfma_kernel", "gauss_h_kernel", "gauss_v_kernel", "pointwise_kernel", "box_", "median_kernel", "warp_", "mesh_kernel",
        "brush_kernel", "chunk_kernel")


def short(name: str) -> str:
    for k in OURS:
        if k in name:
            i = name.find(k)
            return name[i:].split("(")[0][:48]
    return ""


def main(root: str) -> None:
    stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
    for f in stats:
        print(f"# kernel stats ({os.path.relpath(f, root)})")
        for row in csv.DictReader(open(f)):
            n = short(row.get("Name", ""))
            if n:
                print(f"{n:50s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total_ns={row.get('TotalDurationNs')} pct={row.get('Percentage')}")
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                n = short(row.get("Kernel_Name", ""))
                if n:
                    acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print(f"# counters ({os.path.relpath(f, root)}) — mean per launch")
            for n, cs in acc.items():
                for c, vals in cs.items():
                    m = sum(vals) / len(vals)
                    extra = ""
                    if c == "FETCH_SIZE":
                        extra = f"  -> HBM read bytes/launch (KB x 1024 x 2 gfx950 correction) = {m * 1024 * 2:.4g}"
                    if c == "WRITE_SIZE":
                        extra = f"  -> HBM write bytes/launch (KB x 1024) = {m * 1024:.4g}"
                    print(f"{n:50s} {c:24s} {m:.6g} (n={len(vals)}){extra}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
