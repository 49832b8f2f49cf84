#!/usr/bin/env python3
"""tools/isa_audit.py <file.hip> <kernel-name-substring> [--flags "..."] [--loop N] — instruction audit of a kernel's innermost loops (no GPU needed).

Compiles the file for gfx950 with the library's flags (device only, -S), finds the kernel whose mangled name contains the substring, and for each innermost
loop ("Inner Loop Header" in hipcc's asm) prints the instruction count by class and the VALU issue cycles at the MEASURED in-mix costs of this chip
(profiles/r05_valu_rates.txt: v_fma_f32 = 2 cycles per wave-instruction; conversions, min / max, integer multiplies, v_perm, three-operand integer forms 3.2-3.5).
Every path through the loop body is counted (both sides of a branch), so the figure is an upper bound for one trip; --path lists basic blocks to include."""
import argparse, os, re, subprocess, sys, tempfile, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "paintfe_amd", "csrc")
BASE = ["-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-gpu-flush-denormals-to-zero",
        "-Wno-unused-function", "--offload-arch=gfx950", "--cuda-device-only", "-S"]


def rates():
    t = {}
    p = os.path.join(ROOT, "profiles", "r05_valu_rates.txt")
    for line in open(p):
        m = re.match(r"(v_\w+)\s+[\d.]+ ms\s+([\d.]+) cycles", line)
        if m:
            t[m.group(1)] = float(m.group(2))
    t["v_fma_f32"] = 2.0
    return t


def cost(mn, table):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if base in table:
        return table[base]
    if base.startswith("v_cvt_f32_ubyte"):
        return table.get("v_cvt_f32_ubyte0", 3.2)
    if base.startswith(("v_cvt_", "v_floor", "v_trunc", "v_rndne", "v_ceil", "v_fract")):
        return 3.3
    if base.startswith(("v_min", "v_max", "v_med3", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_mul_u32", "v_mul_i32", "v_perm", "v_bfe", "v_bfi", "v_alignb", "v_lshl_add",
                        "v_lshl_or", "v_and_or", "v_add3", "v_or3", "v_xad", "v_sad", "v_bcnt", "v_add_lshl", "v_pk_", "v_readlane", "v_readfirstlane", "v_writelane", "v_cmp")):
        return 3.3
    if base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return 6.8
    if base.startswith("v_mfma"):
        return 0.0
    return 1.9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file"); ap.add_argument("kernel")
    ap.add_argument("--flags", default=""); ap.add_argument("--loop", type=int, default=-1); ap.add_argument("--whole", action="store_true", help="count the whole kernel body (straight-line kernels)")
    a = ap.parse_args()
    extra = subprocess.run(["make", "-pn", "-C", CSRC], capture_output=True, text=True).stdout
    stem = os.path.splitext(os.path.basename(a.file))[0]
    m = re.search(rf"^FLAGS_{stem} := (.*)$", extra, re.M)
    flags = BASE + (m.group(1).split() if m else []) + a.flags.split()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, os.path.join(CSRC, os.path.basename(a.file)), "-o", out], cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    table = rates()
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and a.kernel in l), None)
    if start is None:
        sys.exit(f"no kernel matching {a.kernel}")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.amdhsa_kernel") or lines[i].strip() == ".end_amdhsa_kernel" or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    print("kernel:", lines[start].split(":")[0])
    for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "codeLenInByte"):
        for l in lines[end:end + 80]:
            if key in l and l.strip().startswith(";"):
                print("  " + l.strip("; ").strip()); break
    # innermost loops: header label ... the last block that names it as its header
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if m:
            name = m.group(1)[2:]
            last = i
            for j in range(i + 1, len(body)):
                if re.match(r"^\.LBB\d+_\d+:", body[j]):
                    if f"Header={name}" in body[j]: last = j
                    else:
                        if j > last: 
                            # blocks of the loop are contiguous: stop at the first label that is not in it
                            break
            # extend to the end of the last block
            k = last + 1
            while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]): k += 1
            loops.append((name, i, k))
    if a.whole or not loops:
        loops = [("whole kernel", 0, len(body))]
    for n, (name, i, k) in enumerate(loops):
        if a.loop >= 0 and n != a.loop: continue
        cls = collections.Counter(); mn_count = collections.Counter(); cyc = 0.0
        for l in body[i:k]:
            t = l.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"): continue
            mn = t.split()[0]
            mn_count[re.sub(r"_(e32|e64)$", "", mn)] += 1
            if mn.startswith("v_mfma"): cls["mfma"] += 1
            elif mn.startswith("v_"): cls["valu"] += 1; cyc += cost(mn, table)
            elif mn.startswith("s_waitcnt") or mn.startswith("s_nop"): cls["wait/nop"] += 1
            elif mn.startswith(("s_cbranch", "s_branch")): cls["branch"] += 1
            elif mn.startswith("s_"): cls["salu"] += 1
            elif mn.startswith("ds_"): cls["lds"] += 1
            elif mn.startswith(("buffer_", "global_", "flat_", "scratch_")): cls["vmem"] += 1
            else: cls["other"] += 1
        print(f"\nloop {n} ({name}, {k - i} asm lines): " + ", ".join(f"{c} {v}" for c, v in sorted(cls.items())) + f"; VALU issue cycles at measured in-mix costs: {cyc:.0f}")
        print("  " + ", ".join(f"{m} x{c}" for m, c in mn_count.most_common(40)))


if __name__ == "__main__":
    main()
