#!/usr/bin/env python3
"""tools/lds_bank_sim.py — LDS-array cycles and bank conflicts of gauss_strip64_kernel's accesses (k_gauss.hip), per step and CU, by the lane-group / banking model
of /opt/skills/guides/MI355X_MICROARCH.md (LDS section): a wave64 access is served in fixed lane groups, one LDS cycle per group when conflict-free, one more
per extra distinct address on a busy bank.  No GPU needed; the totals are checked against rocprofv3's SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT in
profiles/r06_tuning.md.  Usage: python tools/lds_bank_sim.py [--nkb 8] [--variant current|swz]"""
import argparse
import collections

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
HALVES = [list(range(32)), list(range(32, 64))]
CONTIG16 = [list(range(k, k + 16)) for k in range(0, 64, 16)]
CONTIG8 = [list(range(k, k + 8)) for k in range(0, 64, 8)]
# instruction -> (lane groups, bytes, bank modulus)
KINDS = {"read_b32": (HALVES, 4, 32), "read_b64": (HALVES, 8, 64), "read_b128": (G128, 16, 64), "write_b32": (HALVES, 4, 32), "write_b64": (CONTIG16, 8, 32),
         "write_b128": (CONTIG8, 16, 32)}


def cycles(kind, addr_of_lane, active=lambda l: True):
    """(LDS-array cycles, conflict cycles) of one wave instruction; addr_of_lane(l) = byte address"""
    groups, nbytes, mod = KINDS[kind]
    total = extra = 0
    for g in groups:
        per_bank = collections.defaultdict(set)
        for l in g:
            if not active(l):
                continue
            a = addr_of_lane(l)
            for d in range(nbytes // 4):
                per_bank[(a // 4 + d) % mod].add((a // 4 + d))
        worst = max((len(v) for v in per_bank.values()), default=1)
        total += worst
        extra += worst - 1
    return total, extra


def strip64(nkb, variant):
    nkp = nkb + 2
    xrow = 16 * nkp + 16
    YP = 72
    PLANE = 64 * YP + 32           # halfs
    OUTP = 68                      # dwords
    qn = 5 if nkp == 10 else 4
    rows = []

    def add(name, kind, fn, count, active=lambda l: True):
        c, e = cycles(kind, fn, active)
        rows.append((name, kind, count, c, e))

    swz = variant == "swz"
    # ---- producers (4 waves) ----
    for nb in (0,):  # the two column blocks behave alike
        def hr_store(l, nb=nb, g=0, part=0, ro=0):
            i, hh = l & 31, l >> 5
            x = 32 * nb + i
            y = ro + 4 * hh
            if swz and (x >> 3) & 1:
                y ^= 4
            return 2 * ((part * 4 + g) * PLANE + x * YP + y)
        add("producer: H result -> ring (f16x4)", "write_b64", hr_store, 4 * 16)
    def xp_store(l, c=0):
        frow, fs = l >> 3, l & 7
        return c * 8 * xrow + frow * xrow + 16 * fs
    add("producer: de-interleaved samples (16 B)", "write_b128", xp_store, 4 * 4)
    if qn == 5:
        def xp_store4(l, c=0):
            frow, fs = l >> 3, l & 7
            off = 4 * fs
            if swz and c >= 2:
                off ^= 8
            return c * 8 * xrow + frow * xrow + 128 + off
        add("producer: the window's last quad (4 B)", "write_b32", xp_store4, 4 * 4)
    def frag_read(l, kb=0):
        i, hh = l & 31, l >> 5
        h2 = hh ^ ((i >> 4) & 1) if swz else hh
        return (i >> 3) * 8 * xrow + (i & 7) * xrow + 16 * kb + 8 * h2
    add("producer: A fragments of the window (8 B)", "read_b64", frag_read, 4 * nkp)
    add("producer: Toeplitz fragments (16 B)", "read_b128", lambda l: 16 * l, 4 * nkb)
    # ---- consumers (8 waves) ----
    def a_read(l, xb=0, ro=0):
        i, hh = l & 31, l >> 5
        xl, c = i >> 2, i & 3
        return 2 * (c * PLANE + (8 * xb + xl) * YP + 8 * hh + ro)
    add("consumer: ring -> A fragments (16 B)", "read_b128", a_read, 8 * 4)
    def out_write(l, xb=0, g=0):
        i, hh = l & 31, l >> 5
        col = 8 * xb + hh + 2 * g
        if swz:
            return 4 * (i * 65 + col)
        return 4 * (i * OUTP + col)
    add("consumer: packed pixels -> staging (4 B)", "write_b32", out_write, 8 * 4)
    if swz:
        def out_read(l, k=0, wave=0):
            t = 64 * wave + l
            return 4 * ((t & 31) * 65 + 4 * (t >> 5) + k)
        add("consumer: staging -> registers (4 x 4 B)", "read_b32", out_read, 8 * 4)
    else:
        def out_read(l, wave=0):
            t = 64 * wave + l
            return 4 * ((t >> 4) * OUTP + 4 * (t & 15))
        add("consumer: staging -> registers (16 B)", "read_b128", out_read, 8 * 1)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nkb", type=int, default=8)
    ap.add_argument("--variant", default="current")
    a = ap.parse_args()
    rows = strip64(a.nkb, a.variant)
    tot = ext = 0
    print(f"gauss_strip64_kernel<{a.nkb}>, variant {a.variant}: LDS-array cycles per step and CU")
    for name, kind, count, c, e in rows:
        print(f"  {name:46s} {kind:10s} x{count:3d}  {c:2d} cycles each ({e} of them conflict)  -> {count * c:4d} ({count * e})")
        tot += count * c
        ext += count * e
    print(f"  total {tot} cycles, {ext} of them bank conflicts ({ext / tot:.0%})")


if __name__ == "__main__":
    main()
