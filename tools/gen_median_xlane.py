#!/usr/bin/env python3
"""tools/gen_median_xlane.py — median selection for FOUR horizontally adjacent (2r+1)^2 windows per lane with the sorted columns SHARED ACROSS LANES, r = 2 and 3,
writing paintfe_amd/csrc/k_median_xlane_net.h (round 6).

tools/gen_median_shared.py's lane sorts all 4 + 2r columns its four windows touch; 2r of them belong to its neighbours' windows too and are sorted there as well.
Here a lane sorts only ITS OWN four columns and takes the others from its neighbours with a wave shift (one DPP move per register: full rate, against 3.4 cycles for
a packed min / max on gfx950), and the merged run that straddles two lanes — (own last columns, right neighbour's first columns) — is computed by the left lane of the
pair and handed to the right one:
    r = 2: columns c0 c1 | c2 c3 c4 c5 | c6 c7  (left lane's last two | own | right lane's first two);  pairs P12 (received), P34, P56 (sent on);
           X = P12 U P34 (windows 0, 1), Y = P34 U P56 (windows 2, 3);  window = run U one more sorted column, median = k-th of the union (gen_median_shared.py).
    r = 3: columns c0 c1 c2 | c3 c4 c5 c6 | c7 c8 c9;  core = c3..c6 (own), X = (c1 c2) U core, Y = core U (c7 c8): the pairs (c1 c2) / (c7 c8) are the neighbours'
           own pairs (their (c5 c6) / (c3 c4)), received merged.
Graph nodes are min / max or a wave shift (SHR: the value of the lane to the left, SHL: of the lane to the right); dead nodes are pruned from the outputs backwards
THROUGH the shifts (a shifted value is needed in the sending lane), to a fixed point.  Verified like gen_median_shared.py: every window on every combination of
ones-counts of its 2r+1 columns (laid out over three simulated lanes), and on random bytes with heavy ties over a 66-lane row; the merge / k-th-of-union pieces are
verified exhaustively by gen_median_shared.verify_pieces()."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_median_shared import SORTERS, Graph, verify_pieces  # noqa: E402


class XGraph(Graph):
    """Graph + unary lane shifts.  Values are arrays whose FIRST axis is the lane."""

    def shr(self, a):   # value held by lane L - 1
        return self.op("shr", a, a)

    def shl(self, a):   # value held by lane L + 1
        return self.op("shl", a, a)

    def evaluate(self, inputs, keep):
        v = list(inputs) + [None] * len(self.ops)
        for k, (kind, a, b) in enumerate(self.ops):
            if not keep[k]:
                continue
            if kind == "min":
                v[self.n_inputs + k] = np.minimum(v[a], v[b])
            elif kind == "max":
                v[self.n_inputs + k] = np.maximum(v[a], v[b])
            else:
                x = v[a]
                y = np.zeros_like(x)   # lanes at the ends receive nothing meaningful (they are halo lanes in the kernel)
                if kind == "shr":
                    y[1:] = x[:-1]
                else:
                    y[:-1] = x[1:]
                v[self.n_inputs + k] = y
        return v


SORT4 = [(0, 1), (2, 3), (0, 2), (1, 3), (1, 2)]


def build_r2_two_rows():
    """r = 2, TWO vertically adjacent rows of four windows per lane: own columns of six pixels; rows 1 .. 4 are common to both window rows and sorted once (5
    comparators), row 0 / row 5 is merged in for the upper / lower window row (4 each): 13 comparators per column and two rows instead of 2 x 9.
    Input id = own column * 6 + row; outputs 0 .. 3 = upper row's windows, 4 .. 7 = lower row's."""
    g = XGraph(4 * 6)
    mids = [g.sort([c * 6 + k for k in (1, 2, 3, 4)], SORT4) for c in range(4)]
    outs = []
    for half in range(2):
        own = [g.merge([c * 6 + 0], mids[c]) if half == 0 else g.merge(mids[c], [c * 6 + 5]) for c in range(4)]
        c = [[g.shr(x) for x in own[2]], [g.shr(x) for x in own[3]], own[0], own[1], own[2], own[3], [g.shl(x) for x in own[0]], [g.shl(x) for x in own[1]]]
        P34 = g.merge(c[3], c[4])
        P56 = g.merge(c[5], c[6])
        P12 = [g.shr(x) for x in P56]
        X, Y = g.merge(P12, P34), g.merge(P34, P56)
        outs += [g.kth_of_union(X, c[0], 13), g.kth_of_union(X, c[5], 13), g.kth_of_union(Y, c[2], 13), g.kth_of_union(Y, c[7], 13)]
    return g, outs, g.prune(outs)


def verify_two_rows(g, outs, keep):
    s, r = 5, 2
    need_ones = 13
    levels = np.arange(s + 1, dtype=np.int8)
    grids = np.meshgrid(*[levels] * s, indexing="ij")
    counts = [x.reshape(-1) for x in grids]
    zero = np.zeros(len(counts[0]), np.uint8)
    for half in range(2):
        for j in range(4):
            cols = [[[zero] * 6 for _ in range(4)] for _ in range(3)]
            for t in range(s):
                gc = j - r + t + 4
                lane, oc = gc // 4, gc % 4
                col6 = [zero] * 6
                ones = [(counts[t] > (s - 1 - k)).astype(np.uint8) for k in range(s)]
                for k in range(s): col6[k + half] = ones[k]          # the window's five rows are rows half .. half + 4 of the six; the sixth row stays zero
                cols[lane][oc] = col6
            ins = [np.stack([cols[lane][c][k] for lane in range(3)]) for c in range(4) for k in range(6)]
            got = g.evaluate(ins, keep)[outs[4 * half + j]][1]
            want = (sum(c.astype(np.int32) for c in counts) >= need_ones).astype(np.uint8)
            if not np.array_equal(got, want):
                return f"two rows: half {half} window {j} fails the 0/1 test"
            # the sixth row must not matter: all ones there
            for lane in range(3):
                for c in range(4):
                    cols[lane][c] = list(cols[lane][c]); cols[lane][c][5 if half == 0 else 0] = np.ones(len(zero), np.uint8)
            ins = [np.stack([cols[lane][c][k] for lane in range(3)]) for c in range(4) for k in range(6)]
            got = g.evaluate(ins, keep)[outs[4 * half + j]][1]
            if not np.array_equal(got, want):
                return f"two rows: half {half} window {j} depends on the other row"
    rng = np.random.default_rng(202)
    nl = 66
    for nlev in (256, 5, 2):
        px = rng.integers(0, nlev, (6, 4 * nl, 1 << 11), dtype=np.uint8)
        ins = [np.stack([px[k, 4 * lane + c] for lane in range(nl)]) for c in range(4) for k in range(6)]
        v = g.evaluate(ins, keep)
        for half in range(2):
            for j in range(4):
                got = v[outs[4 * half + j]]
                for lane in range(1, nl - 1):
                    x0 = 4 * lane + j - r
                    want = np.sort(px[half:half + 5, x0:x0 + s].reshape(s * s, -1), axis=0)[12]
                    if not np.array_equal(got[lane], want):
                        return f"two rows: half {half} window {j} lane {lane} fails on random bytes ({nlev} levels)"
    return None


def build(r):
    s = 2 * r + 1
    g = XGraph(4 * s)   # input id = own column * s + row
    own = [g.sort([c * s + k for k in range(s)], SORTERS[r]) for c in range(4)]
    K = (s * s) // 2 + 1
    if r == 2:
        c = [[g.shr(x) for x in own[2]], [g.shr(x) for x in own[3]], own[0], own[1], own[2], own[3], [g.shl(x) for x in own[0]], [g.shl(x) for x in own[1]]]
        P34 = g.merge(c[3], c[4])
        P56 = g.merge(c[5], c[6])
        P12 = [g.shr(x) for x in P56]                    # the left lane's (its c5, its c6) = (c1, c2) here
        X, Y = g.merge(P12, P34), g.merge(P34, P56)
        outs = [g.kth_of_union(X, c[0], K), g.kth_of_union(X, c[5], K), g.kth_of_union(Y, c[2], K), g.kth_of_union(Y, c[7], K)]
    elif r == 3:
        A, B = g.merge(own[0], own[1]), g.merge(own[2], own[3])      # own pairs (c3 c4), (c5 c6)
        core = g.merge(A, B)
        c0 = [g.shr(x) for x in own[1]]
        L12 = [g.shr(x) for x in B]                                    # left lane's (c5 c6) = (c1, c2) here
        R78 = [g.shl(x) for x in A]                                    # right lane's (c3 c4) = (c7, c8) here
        c9 = [g.shl(x) for x in own[2]]
        c2 = [g.shr(x) for x in own[3]]
        c7 = [g.shl(x) for x in own[0]]
        X, Y = g.merge(L12, core), g.merge(core, R78)
        # windows: W0 = c0..c6 = c0 U X, W1 = c1..c7 = X U c7, W2 = c2..c8 = c2 U Y, W3 = c3..c9 = Y U c9
        outs = [g.kth_of_union(X, c0, K), g.kth_of_union(X, c7, K), g.kth_of_union(Y, c2, K), g.kth_of_union(Y, c9, K)]
    else:
        raise ValueError(r)
    # prune: needed = outputs; a needed shift makes its source needed (in the sending lane = the same node id, every lane runs the same code)
    keep = g.prune(outs)
    return g, outs, keep


def lanes_input(r, cols):
    """cols[lane][own column][row] -> the graph's input list (arrays with the lane axis first)"""
    s = 2 * r + 1
    return [np.stack([cols[lane][c][k] for lane in range(len(cols))]) for c in range(4) for k in range(s)]


def verify(r, g, outs, keep):
    s = 2 * r + 1
    need_ones = s * s - (s * s) // 2
    # every window of the middle lane of three on every combination of ones-counts of its columns; every other column all zeros
    levels = np.arange(s + 1, dtype=np.int8)
    grids = np.meshgrid(*[levels] * s, indexing="ij")
    counts = [x.reshape(-1) for x in grids]
    n = len(counts[0])
    zero = np.zeros(n, np.uint8)
    for j in range(4):
        # global column index of window j's first column, relative to the middle lane's own column 0: j - r
        cols = [[[zero] * s for _ in range(4)] for _ in range(3)]
        for t in range(s):
            gc = j - r + t + 4            # column index counted from lane 0's own column 0
            lane, oc = gc // 4, gc % 4
            cols[lane][oc] = [(counts[t] > (s - 1 - k)).astype(np.uint8) for k in range(s)]   # unsorted on purpose
        got = g.evaluate(lanes_input(r, cols), keep)[outs[j]][1]
        want = (sum(c.astype(np.int32) for c in counts) >= need_ones).astype(np.uint8)
        if not np.array_equal(got, want):
            return f"window {j} fails the 0/1 test"
    # random bytes with ties over a row of 66 lanes: lanes 1 .. 64 are checked
    rng = np.random.default_rng(100 + r)
    nl = 66
    for nlev in (256, 5, 2):
        px = rng.integers(0, nlev, (s, 4 * nl, 1 << 11), dtype=np.uint8)   # [row][column][sample]
        cols = [[[px[k, 4 * lane + c] for k in range(s)] for c in range(4)] for lane in range(nl)]
        v = g.evaluate(lanes_input(r, cols), keep)
        for j in range(4):
            got = v[outs[j]]
            for lane in range(1, nl - 1):
                x0 = 4 * lane + j - r
                want = np.sort(px[:, x0:x0 + s].reshape(s * s, -1), axis=0)[(s * s) // 2]
                if not np.array_equal(got[lane], want):
                    return f"window {j}, lane {lane} fails on random bytes ({nlev} levels)"
    return None


def emit(r, g, outs, keep, rows=1):
    s = 2 * r + 1 + (rows - 1)
    n_mm = sum(1 for k, o in enumerate(g.ops) if keep[k] and o[0] in ("min", "max"))
    n_sh = sum(1 for k, o in enumerate(g.ops) if keep[k] and o[0] in ("shr", "shl"))
    lines = [f"// median of four adjacent {2 * r + 1}x{2 * r + 1} windows per lane, sorted columns shared across lanes: {n_mm} min / max operations + {n_sh} wave shifts per lane\n"
             f"// ({n_mm / 4:.1f} + {n_sh / 4:.1f} per window; k_median_shared_net.h: {({2: 334, 3: 798}[r]) / 4:.1f} min / max).  IN(c, k) = row k of OWN column c (columns x0 .. x0+3);\n"
             f"// SHR(v) / SHL(v) = v as held by the lane to the left / right; OUT(j, v) receives the median of the window centred on own column j\n"
             f"#define PFX_MEDIAN_XLANE_R{r}{'_ROWS2' if rows == 2 else ''}(T, IN, MIN, MAX, SHR, SHL, OUT) \\\n"]
    if rows == 2:
        lines[0] = lines[0].replace("median of four adjacent", "TWO ROWS (IN(c, 0 .. 5) = six rows; OUT 0 .. 3 upper, 4 .. 7 lower window row) of four adjacent")
    name = lambda i: f"IN({i // s}, {i % s})" if i < g.n_inputs else f"n{i - g.n_inputs}"
    for k, (kind, a, b) in enumerate(g.ops):
        if not keep[k]:
            continue
        if kind in ("min", "max"):
            lines.append(f"    const T n{k} = {'MIN' if kind == 'min' else 'MAX'}({name(a)}, {name(b)}); \\\n")
        else:
            lines.append(f"    const T n{k} = {'SHR' if kind == 'shr' else 'SHL'}({name(a)}); \\\n")
    lines.append("    " + " ".join(f"OUT({j}, {name(o)});" for j, o in enumerate(outs)) + "\n")
    return "".join(lines), n_mm, n_sh


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "paintfe_amd", "csrc", "k_median_xlane_net.h")
    text = ["// k_median_xlane_net.h — generated by tools/gen_median_xlane.py (do not edit): median selection with sorted columns shared across the lanes of a wave;\n"
            "// verified on 0/1 inputs (every window, every combination of ones-counts of its columns, over three simulated lanes) and on random bytes with ties.\n#pragma once\n"]
    if "--no-verify" not in sys.argv:
        err = verify_pieces()
        if err:
            print("VERIFICATION FAILED:", err, file=sys.stderr)
            return 1
    for r in (2, 3):
        g, outs, keep = build(r)
        if "--no-verify" not in sys.argv:
            err = verify(r, g, outs, keep)
            if err:
                print(f"VERIFICATION FAILED (r = {r}):", err, file=sys.stderr)
                return 1
        t, n_mm, n_sh = emit(r, g, outs, keep)
        print(f"r={r}: {n_mm} min / max + {n_sh} shifts per lane = {n_mm / 4:.1f} + {n_sh / 4:.1f} per window", file=sys.stderr)
        text.append(t)
    g, outs, keep = build_r2_two_rows()
    if "--no-verify" not in sys.argv:
        err = verify_two_rows(g, outs, keep)
        if err:
            print("VERIFICATION FAILED:", err, file=sys.stderr)
            return 1
    t, n_mm, n_sh = emit(2, g, outs, keep, rows=2)
    print(f"r=2, two rows: {n_mm} min / max + {n_sh} shifts per lane = {n_mm / 8:.1f} + {n_sh / 8:.1f} per window", file=sys.stderr)
    text.append(t)
    open(out, "w").write("".join(text))
    print(out, file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
