"""CPU study (no GPU, no product code): how many steps of the serial search chain would a k-wide SPECULATIVE step save?

A speculative step takes the k best open nodes (key order, first index on ties) at once, expands all of them in parallel lane groups and
commits the longest prefix that the sequential algorithm (reference differentiable_astar.py:203-252, one selection per iteration) would have
selected in exactly that order: node i commits iff it is still the minimum after nodes 0..i-1 were expanded AND none of them touched it
(its g / parent would otherwise differ from what the parallel expansion used).  The committed results are then identical to the sequential
run by construction; this script only counts how long the chain of such steps is on the bench workloads.

Usage: python tools/sim_speculative.py [workload] [B] [ks]      (default maze32 4096 2,3,4,8)
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neural-astar_amd")]
import bench  # noqa: E402


DIST = int(os.environ.get("SIM_DIST", "0"))


def heuristic(H, W, gr, gc):
    r = np.arange(H, dtype=np.float32)[:, None] * np.ones((1, W), np.float32)
    c = np.arange(W, dtype=np.float32)[None, :] * np.ones((H, 1), np.float32)
    dr, dc = np.abs(r - np.float32(gr)), np.abs(c - np.float32(gc))
    h = (dr + dc) - np.minimum(dr, dc)
    a, b = r - np.float32(gr), c - np.float32(gc)
    euc = np.sqrt(a * a + b * b).astype(np.float32)
    return (h + np.float32(0.001) * euc).astype(np.float32)


def simulate(cost, passable, s_idx, g_idx, ks, g_ratio=0.5, chunk=0, equal_only=False):
    """chunk > 0: candidates are the minima of `chunk`-cell chunks (what the kernel's open list holds), one per chunk;
    equal_only: only candidates whose key EQUALS the best key (a ballot instead of a sort)"""
    H, W = cost.shape
    HW = H * W
    gr, omg, sq = np.float32(g_ratio), np.float32(1.0 - g_ratio), np.float32(math.sqrt(W))
    cb, pb = cost.reshape(-1), passable.reshape(-1)
    hh = (omg * (heuristic(H, W, g_idx // W, g_idx % W).reshape(-1) + cb)).astype(np.float32)
    g = np.zeros(HW, np.float32)
    key = np.full(HW, np.inf, np.float32)
    closed = np.zeros(HW, bool)
    opened = np.zeros(HW, bool)
    opened[s_idx] = True
    key[s_idx] = (gr * np.float32(0) + hh[s_idx]) / sq
    # per k: (steps so far, remaining speculative list, set of touched nodes since the boundary)
    state = {k: [0, [], set()] for k in ks}
    iters = 0
    while iters < HW:
        mk = np.where(opened, key, np.float32(np.inf))
        s = int(np.argmin(mk))
        if not opened[s]:
            return None
        order = None
        for k in ks:
            st = state[k]
            if st[1] and st[1][0] == s and s not in st[2]:
                st[1].pop(0)
            else:  # boundary: a new speculative step starts here with the k best open nodes
                if order is None:
                    if chunk < 0:  # -1: 2x2 checkerboard subsets (row parity, column parity); -2: chunk (16 cells) index mod 4
                        r_, c_ = np.divmod(np.arange(HW), W)
                        sub = (r_ & 1) * 2 + (c_ & 1) if chunk == -1 else ((np.arange(HW) >> 4) & 3)
                        ci = np.array([np.where(sub == q, mk, np.float32(np.inf)).argmin() for q in range(4)])
                        cand = ci[np.lexsort((ci, mk[ci]))][:max(ks)]
                    elif chunk:
                        cm = mk.reshape(-1, chunk)
                        ci = cm.argmin(axis=1) + np.arange(cm.shape[0]) * chunk
                        cand = ci[np.argsort(mk[ci], kind="stable")][:max(ks)]
                    else:
                        cand = np.argsort(mk, kind="stable")[:max(ks)]
                    order = [int(o) for o in cand if opened[o] and (not equal_only or mk[o] == mk[s])]
                st[0] += 1
                st[1] = list(order[1:k])
                if DIST:  # no speculation on a candidate whose neighbourhood overlaps an earlier one's (Chebyshev distance <= DIST)
                    keep = []
                    for o in st[1]:
                        if any(max(abs(o // W - q // W), abs(o % W - q % W)) <= DIST for q in [s] + keep):
                            break
                        keep.append(o)
                    st[1] = keep
                st[2] = set()
        iters += 1
        closed[s] = True
        if s == g_idx:
            break
        opened[s] = False
        g2 = g[s] + cb[s]
        r, c = divmod(s, W)
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                if not dr and not dc:
                    continue
                rr, cc = r + dr, c + dc
                if rr < 0 or rr >= H or cc < 0 or cc >= W:
                    continue
                n = rr * W + cc
                if pb[n] == 0 or closed[n]:
                    continue
                if opened[n] and not (g[n] > g2):
                    continue
                g[n] = g2
                key[n] = (gr * g2 + hh[n]) / sq
                opened[n] = True
                for k in ks:
                    state[k][2].add(n)
    return iters, {k: state[k][0] for k in ks}


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "maze32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    ks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,3,4,8").split(",")]
    pr = bench.make_problem(kind, B, 1234)
    chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    eq = len(sys.argv) > 5 and sys.argv[5] == "eq"
    m, s, g = (np.asarray(x).reshape(B, x.shape[-2], x.shape[-1]) for x in (pr.map_designs, pr.start_maps, pr.goal_maps))
    res = []
    for b in range(B):
        out = simulate(m[b], m[b], int(s[b].reshape(-1).argmax()), int(g[b].reshape(-1).argmax()), ks, chunk=chunk, equal_only=eq)
        if out is not None:
            res.append((out[0],) + tuple(out[1][k] for k in ks))
    a = np.array(res)
    print(f"{kind} B={B} chunk={chunk} equal_only={eq}: serial steps max {a[:, 0].max()} mean {a[:, 0].mean():.1f}")
    top = np.argsort(-a[:, 0])[:16]
    for i, k in enumerate(ks):
        col = a[:, 1 + i]
        print(f" k={k}: speculative steps max {col.max()} mean {col.mean():.1f}; ratio of maxima {a[:, 0].max() / col.max():.2f}; "
              f"mean ratio {a[:, 0].mean() / col.mean():.2f}; longest 16 maps: {(a[top, 0] / a[top, 1 + i]).mean():.2f}")


if __name__ == "__main__":
    main()
