#!/usr/bin/env python3
"""tools/isa_table.py [out.md] — per-blend-mode gfx950 instruction table of the compositor's pixel code.

Compiles one probe kernel per (blend mode, accumulator specialisation) that runs paintfe_amd/csrc/k_blend.h's
blendN_nx<M, 4, OB> on 4 pixels per lane — the pixel code one layer of flatten_stream_kernel executes per lane — and counts
the instructions hipcc emits (same flags as the library).  The harness (4 accumulator pixels and the layer's pixel
quad loaded from / stored to global memory) is measured by a probe that blends nothing and subtracted.
No GPU needed (cross-compile only)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ["normal", "multiply", "screen", "additive", "reflect", "glow", "color_burn", "color_dodge", "overlay", "difference",
         "negation", "lighten", "darken", "xor", "overwrite", "hard_light", "soft_light", "exclusion", "subtract", "divide",
         "linear_burn", "vivid_light", "linear_light", "pin_light", "hard_mix"]
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "--offload-arch=gfx950", "--cuda-device-only", "-S"]

SRC = r'''
#include "k_blend.h"
using namespace pfxk;
// one layer of flatten_stream_kernel on 4 pixels per lane: the layer pixel arrives as four normalised f32 (typed buffer load),
// the accumulator is held as RN(k / 255) (k_blend.h: blend_nx)
template <int M, int OB>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ top, float4* __restrict__ accs, float opacity)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float acc[4][4], t[4][4];
    for (int p = 0; p < 4; ++p) {
        const float4 a = accs[i * 4 + p]; acc[p][0] = a.x; acc[p][1] = a.y; acc[p][2] = a.z; acc[p][3] = a.w;
        const float4 v = top[i * 4 + p]; t[p][0] = v.x; t[p][1] = v.y; t[p][2] = v.z; t[p][3] = v.w;
    }
    if constexpr (M >= 0) blendN_nx<(uint32_t)M, 4, OB>(acc, t, opacity, rs_clamp(opacity, 0.0f, 1.0f));
    else { acc[0][0] += t[0][0] + t[1][1] + t[2][2] + t[3][3]; }
    for (int p = 0; p < 4; ++p) accs[i * 4 + p] = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
}
template __global__ void probe<-1, 0>(const float4*, float4*, float);
'''


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    src = SRC
    for m in range(25):
        for ob in (0, 1, 2):
            src += f"template __global__ void probe<{m}, {ob}>(const float4*, float4*, float);\n"
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "probe.hip")
        open(f, "w").write(src)
        asm = os.path.join(td, "probe.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I", os.path.join(ROOT, "paintfe_amd", "csrc"), "-I",
                               os.path.join(ROOT, "include"), "-o", asm, f], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    counts = {}
    for mm in re.finditer(r"^_Z5probeILi(n?\d+)ELi(\d)EEvPK15HIP_vector_typeIfLj4EEPS1_f:(.*?)s_endpgm", text, re.S | re.M):
        m, ob, body = int(mm.group(1).replace("n", "-")), int(mm.group(2)), mm.group(3)
        ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((";", "."))]
        valu = sum(1 for i in ins if i.startswith("v_"))
        salu = sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop", "s_load")))
        trans = sum(1 for i in ins if i.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")))
        # issue cost per class, measured on gfx950 (tools/lab/valu_tput.hip -> profiles/r03_valu_rates.txt): FMA / MUL / ADD / SUB f32, moves, logic, shifts
        # and integer adds 2 cycles per wave64 instruction; transcendentals 8; everything else (min / max / med3, trunc, conversions, compares, selects,
        # bfi / perm / and_or, packed ops) 4
        full = ("v_fma_f32", "v_fmac_f32", "v_fmaak", "v_fmamk", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
                "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_not_b32", "v_bitop3_b32")
        cyc = 0
        half = 0
        for i in ins:
            if not i.startswith("v_"):
                continue
            if i.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")):
                cyc += 8
            elif i.startswith(full):
                cyc += 2
            else:
                cyc += 4
                half += 1
        counts[(m, ob)] = (valu, salu, trans, cyc, half)
    base = counts[(-1, 0)][0] - 4  # the empty probe's own four adds
    lines = ["# Compositor pixel code: gfx950 instructions per layer-pixel, by blend mode",
             "",
             "`tools/isa_table.py`: static count of what hipcc emits for `blendN_nx<M, 4, OB>` (k_blend.h) on 4 pixels per lane — the",
             "pixel code one layer of `flatten_stream_kernel` executes — harness subtracted, divided by 4.  OB 0 = general accumulator,",
             "OB 1 = wave-uniform opaque accumulator (out_a == 1), OB 2 = additionally an opaque layer at 100 % opacity.",
             "VALU/px is what bounds the kernel (one wave64 VALU instruction = 2 issue cycles on a SIMD);",
             "`trans` = quarter-rate instructions among them (v_rcp / v_sqrt).", "",
             "`cycles` = the same instructions weighted by their measured issue cost (2 / 4 / 8 cycles per wave64 instruction, profiles/r03_valu_rates.txt);",
             "`half` = how many of them are 4-cycle instructions (compare, select, min / max, trunc, convert).", "",
             "| mode | OB0 VALU/px | OB0 cycles/px | OB0 half-rate/px | OB0 trans/px | OB0 SALU/quad | OB1 VALU/px | OB1 cycles/px | OB2 VALU/px | OB2 cycles/px |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    tot = [0.0] * 6
    base_c = counts[(-1, 0)][3] - 8
    for m in range(25):
        v0, s0, t0, c0, h0 = counts[(m, 0)]
        v1, _, _, c1, _ = counts[(m, 1)]
        v2, _, _, c2, _ = counts[(m, 2)]
        a, b, c = (v0 - base) / 4, (v1 - base) / 4, (v2 - base) / 4
        ca, cb, cc = (c0 - base_c) / 4, (c1 - base_c) / 4, (c2 - base_c) / 4
        for k, v in enumerate((a, b, c, ca, cb, cc)):
            tot[k] += v
        lines.append(f"| {m} {MODES[m]} | {a:.2f} | {ca:.1f} | {(h0 - counts[(-1, 0)][4]) / 4:.1f} | {t0 / 4:.2f} | {s0} | {b:.2f} | {cb:.1f} | {c:.2f} | {cc:.1f} |")
    lines.append(f"| **mean of 25** | **{tot[0] / 25:.2f}** | **{tot[3] / 25:.1f}** | | | | **{tot[1] / 25:.2f}** | **{tot[4] / 25:.1f}** | **{tot[2] / 25:.2f}** | **{tot[5] / 25:.1f}** |")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
