#!/usr/bin/env python3
"""tools/gen_median_shared.py — median selection for FOUR horizontally adjacent (2r+1)^2 windows at once, r = 2 and 3, writing
paintfe_amd/csrc/k_median_shared_net.h.

The four windows of a lane share most of their columns, so the work is organised around what they share instead of around one window
(the 113 / 313-comparator single-window networks of tools/gen_median_net.py):
  1. every column (2r+1 pixels) is sorted once — a column serves up to four windows;
  2. sorted columns are merged pairwise (Batcher's odd-even merge), pairs into the run of columns two neighbouring windows have in
     common: with columns c0 .. c(3+2r) and windows Wj = cj .. c(j+2r)
        r = 2:  X = c1..c4 (W0, W1)   Y = c3..c6 (W2, W3)          r = 3:  X = c1..c6 (W0, W1)   Y = c3..c8 (W2, W3)
     the pair c3..c4 (r = 2) / the quad c3..c6 (r = 3) is merged once and feeds both X and Y;
  3. a window is its shared run plus ONE more sorted column; the median (element len/2 of the ascending sort, ref:
     src/ops/effects/noise.rs:398-404) of `run U column` is  min_i max(run[i-1], column[K-i-1])  over the splits i + (K - i) = K — only
     2r+2 consecutive elements of the run can matter, so everything that does not reach them is pruned away (dead-code elimination over
     the whole graph).
Every node is a min or a max: the graph computes order statistics of any totally ordered input if it does so for 0/1 inputs (min and max
commute with monotone maps), so it is verified exhaustively on 0/1 inputs: the column sorters on all 2^(2r+1) inputs, then every window
on every combination of ones-counts of its 2r+1 sorted columns ((2r+2)^(2r+1) cases) — and on random bytes with heavy ties."""
import os
import sys

import numpy as np

SORT5 = [(0, 1), (3, 4), (2, 4), (2, 3), (0, 3), (0, 2), (1, 4), (1, 3), (1, 2)]
SORT7 = [(1, 2), (3, 4), (5, 6), (0, 2), (3, 5), (4, 6), (0, 1), (4, 5), (2, 6), (0, 4), (1, 5), (0, 3), (2, 5), (1, 3), (2, 4), (2, 3)]
SORT9 = [(0, 1), (3, 4), (6, 7), (1, 2), (4, 5), (7, 8), (0, 1), (3, 4), (6, 7), (0, 3), (3, 6), (0, 3), (1, 4), (4, 7), (1, 4), (2, 5), (5, 8), (2, 5),
         (1, 3), (5, 7), (2, 6), (4, 6), (2, 4), (2, 3), (5, 6)]  # 25 comparators (Floyd)
SORTERS = {2: SORT5, 3: SORT7, 4: SORT9}


class Graph:
    def __init__(self, n_inputs):
        self.n_inputs = n_inputs
        self.ops = []  # (kind, a, b): node id = n_inputs + index

    def op(self, kind, a, b):
        self.ops.append((kind, a, b))
        return self.n_inputs + len(self.ops) - 1

    def ce(self, a, b):
        return self.op("min", a, b), self.op("max", a, b)

    def sort(self, ids, net):
        w = list(ids)
        for (i, j) in net:
            w[i], w[j] = self.ce(w[i], w[j])
        return w

    def merge(self, A, B):
        """Batcher's odd-even merge of two sorted lists of any lengths (TAOCP 5.3.4)"""
        if not A:
            return list(B)
        if not B:
            return list(A)
        if len(A) == 1 and len(B) == 1:
            return list(self.ce(A[0], B[0]))
        C = self.merge(A[0::2], B[0::2])
        D = self.merge(A[1::2], B[1::2])
        out = [C[0]]
        i = 0
        while i < len(D) and i + 1 < len(C):
            lo, hi = self.ce(D[i], C[i + 1])
            out += [lo, hi]
            i += 1
        out += C[i + 1:] + D[i:]
        return out

    def kth_of_union(self, A, B, K):
        """K-th smallest (1-based) of two sorted lists: min over i of max(A[i-1], B[K-i-1]), a missing operand standing for -inf"""
        terms = []
        for i in range(max(0, K - len(B)), min(K, len(A)) + 1):
            a = A[i - 1] if i >= 1 else None
            b = B[K - i - 1] if K - i >= 1 else None
            terms.append(a if b is None else b if a is None else self.op("max", a, b))
        res = terms[0]
        for t in terms[1:]:
            res = self.op("min", res, t)
        return res

    def prune(self, outs):
        need = set(outs)
        keep = [False] * len(self.ops)
        for k in range(len(self.ops) - 1, -1, -1):
            if self.n_inputs + k in need:
                keep[k] = True
                need.add(self.ops[k][1])
                need.add(self.ops[k][2])
        return keep

    def evaluate(self, inputs, keep):
        v = list(inputs) + [None] * len(self.ops)
        for k, (kind, a, b) in enumerate(self.ops):
            if keep[k]:
                v[self.n_inputs + k] = np.minimum(v[a], v[b]) if kind == "min" else np.maximum(v[a], v[b])
        return v


def build(r):
    s, ncol = 2 * r + 1, 4 + 2 * r
    g = Graph(ncol * s)  # input id = column * s + row
    net = SORTERS[r]
    col = [g.sort([c * s + k for k in range(s)], net) for c in range(ncol)]
    K = (s * s) // 2 + 1  # 1-based rank of element len/2
    if r == 4:  # X = c1..c8 (W0, W1), Y = c3..c10 (W2, W3); the six columns c3..c8 all four windows share are merged once
        core = g.merge(g.merge(g.merge(col[3], col[4]), g.merge(col[5], col[6])), g.merge(col[7], col[8]))
        X = g.merge(g.merge(col[1], col[2]), core)
        Y = g.merge(core, g.merge(col[9], col[10]))
        outs = [g.kth_of_union(X, col[0], K), g.kth_of_union(X, col[9], K), g.kth_of_union(Y, col[2], K), g.kth_of_union(Y, col[11], K)]
    elif r == 2:
        m34 = g.merge(col[3], col[4])
        X = g.merge(g.merge(col[1], col[2]), m34)
        Y = g.merge(m34, g.merge(col[5], col[6]))
        outs = [g.kth_of_union(X, col[0], K), g.kth_of_union(X, col[5], K), g.kth_of_union(Y, col[2], K), g.kth_of_union(Y, col[7], K)]
    else:
        core = g.merge(g.merge(col[3], col[4]), g.merge(col[5], col[6]))
        X = g.merge(g.merge(col[1], col[2]), core)
        Y = g.merge(core, g.merge(col[7], col[8]))
        outs = [g.kth_of_union(X, col[0], K), g.kth_of_union(X, col[7], K), g.kth_of_union(Y, col[2], K), g.kth_of_union(Y, col[9], K)]
    return g, outs, g.prune(outs)


def verify_pieces():
    """Batcher's merge and the k-th-of-union formula on every pair of sorted 0/1 lists up to the lengths used (72 + 9): with the column sorters
    correct, a graph composed of correct merges computes correct order statistics, and pruning only removes nodes no output depends on"""
    for la in range(1, 73):
        for lb in (1, 2, 5, 7, 9, 10, 14, 18, 27, 28, 36, 54):
            if la + lb > 81 or lb > la:
                continue
            g = Graph(la + lb)
            M = g.merge(list(range(la)), list(range(la, la + lb)))
            na, nb = np.meshgrid(np.arange(la + 1), np.arange(lb + 1), indexing="ij")
            na, nb = na.reshape(-1), nb.reshape(-1)
            ins = [(na > (la - 1 - k)).astype(np.uint8) for k in range(la)] + [(nb > (lb - 1 - k)).astype(np.uint8) for k in range(lb)]
            v = g.evaluate(ins, [True] * len(g.ops))
            tot = na + nb
            for k, m in enumerate(M):
                if not np.array_equal(v[m], (tot > (la + lb - 1 - k)).astype(np.uint8)):
                    return f"merge({la},{lb}) output {k}"
            for K in {(la + lb) // 2 + 1, 1, la + lb}:
                g2 = Graph(la + lb)
                o = g2.kth_of_union(list(range(la)), list(range(la, la + lb)), K)
                v2 = g2.evaluate(ins, [True] * len(g2.ops))
                if not np.array_equal(v2[o], (tot > (la + lb - K)).astype(np.uint8)):
                    return f"kth_of_union({la},{lb},{K})"
    return None


def verify(r, g, outs, keep):
    s, ncol = 2 * r + 1, 4 + 2 * r
    net = SORTERS[r]
    # the column sorter on every 0/1 input
    for bits in range(1 << s):
        w = [(bits >> k) & 1 for k in range(s)]
        for (i, j) in net:
            w[i], w[j] = min(w[i], w[j]), max(w[i], w[j])
        if w != sorted(w):
            return f"sort{s} fails on {bits:b}"
    # every window on every combination of ones-counts of its columns (the other columns do not reach it: all zeros)
    need_ones = s * s - (s * s) // 2  # element len/2 of the ascending sort is 1 iff at least this many ones
    # r <= 3: every combination of ones-counts; r = 4 (10^9 combinations): the combinations of counts from {0, 1, 2, 7, 8, 9} (10^7, every
    # all-low / all-high mix around the median), on top of verify_pieces()
    levels = np.arange(s + 1, dtype=np.int8) if r <= 3 else np.array([0, 1, 2, 7, 8, 9], np.int8)
    grids = np.meshgrid(*[levels] * s, indexing="ij")
    counts = [x.reshape(-1) for x in grids]
    for j in range(4):
        inputs = []
        for c in range(ncol):
            if j <= c < j + s:
                cnt = counts[c - j]
                inputs += [(cnt > (s - 1 - k)).astype(np.uint8) for k in range(s)]  # unsorted on purpose: ones at the top rows first
            else:
                inputs += [np.zeros(len(counts[0]), np.uint8)] * s
        got = g.evaluate(inputs, keep)[outs[j]]
        want = (sum(c.astype(np.int32) for c in counts) >= need_ones).astype(np.uint8)
        if not np.array_equal(got, want):
            return f"window {j} fails the 0/1 test"
    # random bytes with ties
    rng = np.random.default_rng(r)
    for levels in (256, 5, 2):
        px = rng.integers(0, levels, (s, ncol, 1 << 16), dtype=np.uint8)  # [row][column][sample]
        v = g.evaluate([px[k, c] for c in range(ncol) for k in range(s)], keep)
        for j in range(4):
            want = np.sort(px[:, j:j + s].reshape(s * s, -1), axis=0)[(s * s) // 2]
            if not np.array_equal(v[outs[j]], want):
                return f"window {j} fails on random bytes ({levels} levels)"
    return None


def emit(r, g, outs, keep):
    s = 2 * r + 1
    lines = [f"// median of four adjacent {s}x{s} windows: {sum(keep)} min / max operations ({sum(keep) / 4:.1f} per window; the single-window network "
             f"needs {2 * {2: 113, 3: 313}[r] if r in (2, 3) else 'none exists here'})\n// IN(c, k) = row k of column c (columns x0-{r} .. x0+{3 + r}); OUT(j, v) receives window j's median\n"
             f"#define PFX_MEDIAN_SHARED_R{r}(T, IN, MIN, MAX, OUT) \\\n"]
    name = lambda i: f"IN({i // s}, {i % s})" if i < g.n_inputs else f"n{i - g.n_inputs}"
    for k, (kind, a, b) in enumerate(g.ops):
        if keep[k]:
            lines.append(f"    const T n{k} = {'MIN' if kind == 'min' else 'MAX'}({name(a)}, {name(b)}); \\\n")
    lines.append("    " + " ".join(f"OUT({j}, {name(o)});" for j, o in enumerate(outs)) + "\n")
    return "".join(lines)


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "paintfe_amd", "csrc", "k_median_shared_net.h")
    text = ["// k_median_shared_net.h — generated by tools/gen_median_shared.py (do not edit): shared-column median selection, verified\n"
            "// exhaustively on 0/1 inputs (every window, every combination of its sorted columns) and on random bytes with ties.\n#pragma once\n"]
    if "--no-verify" not in sys.argv:
        err = verify_pieces()
        if err:
            print("VERIFICATION FAILED:", err, file=sys.stderr)
            return 1
    for r in (2, 3, 4):
        g, outs, keep = build(r)
        print(f"r={r}: {len(g.ops)} operations built, {sum(keep)} after pruning = {sum(keep) / 4:.1f} per window", file=sys.stderr)
        if "--no-verify" not in sys.argv:
            err = verify(r, g, outs, keep)
            if err:
                print("VERIFICATION FAILED:", err, file=sys.stderr)
                return 1
        text.append(emit(r, g, outs, keep))
    open(out, "w").write("".join(text))
    print(out, file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
