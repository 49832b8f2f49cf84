#!/usr/bin/env python3
"""tools/design_tables.py [--check] — DESIGN.md's measured figures are GENERATED from the committed profiles, not typed in:

    profiles/rNN_ops.txt        tools/bench_ops.py      per-operation kernel times (HIP events on the launch stream)
    profiles/rNN_bench_n1.json  bench.py                the headline line with every BASELINE configuration
    profiles/rNN_pmc.json       tools/prof.sh + tools/pmc_json.py   rocprofv3 averages and PMC figures of the two headline kernels

The newest rNN of each is used.  The blocks between `<!-- BEGIN GENERATED:name -->` and `<!-- END GENERATED:name -->` in DESIGN.md (headline) and
profiles/ops_table.md (operation table) are replaced;
`--check` exits 1 if DESIGN.md is not what the profiles produce (tests/test_host_logic.py runs it, so a stale table fails the CPU gate).
Prose outside the blocks carries no measured figure that a profile holds; history lives in profiles/rNN_tuning.md."""
import ast, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files: raise SystemExit(f"no profiles/{pattern}")
    return files[-1]


def ops_rows(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if line.startswith("{'op'"):
            try: rows.append(ast.literal_eval(line))
            except Exception: pass
    return rows


def fmt_ms(v): return f"{v:.3f}" if v < 10 else f"{v:.2f}"


def block_headline(b, p, names):
    r, f, g = b["roofline"], p["flatten"], p["gauss_strip"]
    km = r["kernel_ms"]
    cp = r.get("clock_power", {})
    cb = b.get("cpu_baseline", {})
    L = []
    L.append(f"Generated from `{names['bench']}` (bench.py, N = 1, driver-style run) and `{names['pmc']}` (rocprofv3 passes of the same command, commit `{p.get('commit')}`).")
    L.append("")
    L.append("| figure | value |")
    L.append("|---|---|")
    L.append(f"| `value` | **{b['value']:,.0f} Mpixels/s** = {b['ms_per_step']:.4f} ms per step ({b['steps']} timed steps; HIP-event step times min / median / max "
             f"{b['step_ms_hip_events']['min']:.3f} / {b['step_ms_hip_events']['median']:.3f} / {b['step_ms_hip_events']['max']:.3f} ms) |")
    L.append(f"| kernels per step (HIP events) | flatten {km['flatten']:.4f} ms, Gaussian {km.get('gauss_mfma', 0):.4f} ms |")
    if b.get("value_exact_f32"):
        xk = (b.get("exact_f32_leg") or {}).get("kernel_ms", {})
        L.append(f"| `value_exact_f32` (same step, bit-exact f32 Gaussian) | {b['value_exact_f32']:,.0f} Mpixels/s = {b['ms_per_step_exact']:.4f} ms per step, pipeline {b['pipeline_frac_exact']:.3f} of 8 TB/s; "
                 f"Gaussian {sum(v for k, v in xk.items() if k.startswith('gauss')):.3f} ms; whole frame bit-exact: {(b.get('exact_f32_leg') or {}).get('gaussian_whole_frame_bitexact')} |")
    L.append(f"| `roofline` (compositor, 132 B/px) | {r['achieved']:,.0f} GB/s = **{r['frac']:.3f} of 8 TB/s**; pipeline (140 B/px) {r['pipeline_achieved_GBs']:,.0f} GB/s = {r['pipeline_frac']:.3f} |")
    L.append(f"| rocprofv3 averages (under the profiler) | `{f['kernel']}` {f['avg_ns_profiled'] / 1e6:.4f} ms, `{g['kernel']}` {g['avg_ns_profiled'] / 1e6:.4f} ms |")
    L.append(f"| compositor HBM traffic (FETCH_SIZE × 2 + WRITE_SIZE) | {f['hbm_bytes'] / 1e9:.3f} GB per launch = {f['hbm_bytes'] / f['algorithmic_bytes']:.3f} × the algorithmic {f['algorithmic_bytes'] / 1e9:.3f} GB |")
    L.append(f"| compositor instruction counts per launch | {f['valu_wave_insts'] / 1e8:.2f}·10⁸ VALU ({f['valu_insts_per_layer_px']:.1f} per layer-pixel of the full stack), "
             f"{f['salu_wave_insts'] / 1e8:.2f}·10⁸ SALU, {f.get('vmem_rd_wave_insts', 0) / 1e7:.2f}·10⁷ typed loads |")
    L.append(f"| compositor unit occupancy (profiled, {f['clock_ghz']:.2f} GHz) | VALU issue {f.get('valu_issue_frac_inmix', 0):.2f} (in-mix costs; {f['valu_issue_frac_profiled']:.2f} with every instruction at 2 cycles), "
             f"scalar unit {f['salu_unit_frac_profiled']:.2f}, texture path ~{f.get('texture_path_frac_profiled', 0):.2f} |")
    L.append(f"| Gaussian (profiled) | MFMA pipe {g['mfma_pipe_frac_profiled']:.2f} busy, LDS {g['lds_busy_frac_profiled']:.2f}, HBM traffic {g['hbm_bytes'] / g['algorithmic_bytes']:.2f} × algorithmic |")
    if cp: L.append(f"| clock and power during the timed workload | {cp.get('clock_ghz_sustained')} GHz sustained (spec {cp.get('clock_ghz_max_spec')}), {cp.get('socket_power_w')} W of a {cp.get('power_cap_w')} W cap |")
    ck = b.get("check", {})
    L.append(f"| parity checks of the timed results | flatten whole frame bit-exact: {ck.get('flatten_whole_frame_bitexact')}; Gaussian whole frame max abs diff {ck.get('gaussian_whole_frame_max_diff')}, "
             f"{ck.get('gaussian_channels_off_by_one')} of the channels off by one |")
    if cb: L.append(f"| `cpu_baseline` | {cb.get('value')} {cb.get('unit')} on {cb.get('cores')} cores ({cb.get('kind')}; with the reference's serial write-back {cb.get('faithful', {}).get('value')}): {str(cb.get('sample'))[:60]} |")
    L.append("")
    L.append("| BASELINE configuration (same line, `configs`) | ms | of 8 TB/s | kernels | check |")
    L.append("|---|---|---|---|---|")
    for name, c in b.get("configs", {}).items():
        ms = c.get("ms")
        extra = ""
        if ms is None and "images_per_s" in c: ms, extra = None, f"{c['images_per_s']:.0f} images/s, {c.get('bound')}-bound at {c.get('frac')} of {c.get('frac_of')}"
        if ms is None and "wall_ms_per_process" in c: extra = f"{min(c['wall_ms_per_process']):.0f}–{max(c['wall_ms_per_process']):.0f} ms of wall clock per process"
        km2 = ", ".join(f"{k} {v}" for k, v in (c.get("kernel_ms") or {}).items())
        chk = "; ".join(f"{k}: {json.dumps(v) if isinstance(v, dict) else v}" for k, v in (c.get("check") or {}).items())
        L.append(f"| `{name}` | {fmt_ms(ms) if ms is not None else '—'} | {c.get('frac') if ms is not None else '—'} | {km2 or extra or '—'} | {chk[:70]} |")
    return "\n".join(L)


def block_ops(rows, names):
    L = [f"Generated from `{names['ops']}` (`tools/bench_ops.py`: HIP events on the launch stream, one box, device-resident data; 8K = 33.18 Mpx unless the row says otherwise).",
         "", "| operation | ms | GB/s (algorithmic) | of 8 TB/s | B/px | note |", "|---|---|---|---|---|---|"]
    for r in rows:
        if r["op"].startswith("flatten 9 layers, mode"): continue
        L.append(f"| {r['op']} | {fmt_ms(r['ms'])} | {r['achieved_GBs']:,.0f} | {r['hbm_frac']:.3f} | {r['alg_bytes_px']} | {r.get('note', '')} |")
    modes = [r for r in rows if r["op"].startswith("flatten 9 layers, mode")]
    if modes:
        L.append("")
        L.append("Per-blend-mode compositor runs (9 layers at 8K on S2's pixel data, 40 B/px; the spread is the blend function's instruction count, `profiles/r03_blend_isa.md`):")
        L.append("")
        L.append("| mode | ms | of 8 TB/s | mode | ms | of 8 TB/s | mode | ms | of 8 TB/s |")
        L.append("|---|---|---|---|---|---|---|---|---|")
        cells = [f"{r['op'].split('mode ')[1]} | {fmt_ms(r['ms'])} | {r['hbm_frac']:.3f}" for r in modes]
        while len(cells) % 3: cells.append(" | | ")
        for i in range(0, len(cells), 3): L.append("| " + " | ".join(cells[i:i + 3]) + " |")
    return "\n".join(L)


def main():
    check = "--check" in sys.argv
    fb, fp, fo = newest("r*_bench_n1.json"), newest("r*_pmc.json"), newest("r*_ops.txt")
    names = {"bench": os.path.relpath(fb, ROOT), "pmc": os.path.relpath(fp, ROOT), "ops": os.path.relpath(fo, ROOT)}
    b = json.loads(open(fb).read().strip().splitlines()[-1])
    p = json.load(open(fp))
    # the headline block lives in DESIGN.md; the per-operation table (70 rows) in profiles/ops_table.md, which DESIGN.md links to (round 6: DESIGN.md describes the
    # shipped design in <= 25 KB)
    blocks = {"headline": ("DESIGN.md", block_headline(b, p, names)), "ops": (os.path.join("profiles", "ops_table.md"), block_ops(ops_rows(fo), names))}
    stale = []
    for name, (rel, body) in blocks.items():
        path = os.path.join(ROOT, rel)
        text = open(path).read()
        pat = re.compile(rf"(<!-- BEGIN GENERATED:{name} -->\n).*?(<!-- END GENERATED:{name} -->)", re.S)
        if not pat.search(text): raise SystemExit(f"{rel} has no GENERATED:{name} block")
        new = pat.sub(lambda m: m.group(1) + body + "\n" + m.group(2), text)
        if new != text:
            stale.append(rel)
            if not check: open(path, "w").write(new)
    if check:
        if stale:
            print("stale generated blocks in", stale, ": run python tools/design_tables.py"); return 1
        print("generated blocks are current"); return 0
    print("updated", stale or "nothing", "from", names)
    return 0


if __name__ == "__main__":
    sys.exit(main())
