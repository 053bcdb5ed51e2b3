"""Dev probe (CPU only): would length-aware placement of the maps on the SIMDs shorten the forward launch?  

Per-SIMD model of the launch fitted in round 2: every SIMD holds 4 chains from t = 0, a chain's step takes
max(lone, per * active_waves) cycles, the launch ends with the slowest SIMD.  Chain lengths = the oracle's step counts of the bench
batches.  Compared: natural order, random order, snake / LPT placement driven by the TRUE lengths, and by the Chebyshev start-goal
distance (the only predictor available before the search).  Output of round 3 (2.1 GHz):

    maze32  corr(chebyshev, steps) = 0.13    natural 180 us | true lengths 153-155 | chebyshev 175-184 | longest chain alone 146
    rand32  corr = 0.54                      natural  90 us | true lengths  88-89  | chebyshev  89-90  | longest chain alone  88
    fewer VALU per step instead (lone 580, per 190):  maze32 natural 145 us, rand32 80 us            <- what round 3 built

Usage: python tools/sim_placement.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neural-astar_amd")]


def simd_time(chains, lone, per):
    c = sorted(chains)
    t, prev, n = 0.0, 0, len(c)
    for i, L in enumerate(c):
        t += (L - prev) * max(lone, per * (n - i))
        prev = L
    return t


def launch_us(assign, iters, nsimd=1024, lone=650, per=240, ghz=2.1):
    return max(simd_time(list(iters[assign == s]), lone, per) for s in range(nsimd)) / (ghz * 1e3)


def snake(order, nsimd=1024):
    a = np.empty(len(order), int)
    for k, idx in enumerate(order):
        r, p = divmod(k, nsimd)
        a[idx] = p if r % 2 == 0 else nsimd - 1 - p
    return a


def lpt(order, pred, nsimd=1024, cap=4):
    load, cnt, a = np.zeros(nsimd), np.zeros(nsimd, int), np.empty(len(order), int)
    for idx in order:
        ok = np.where(cnt < cap)[0]
        s = ok[np.argmin(load[ok])]
        a[idx] = s
        load[s] += pred[idx]
        cnt[s] += 1
    return a


def main():
    import bench
    from oracle import oracle as O
    for kind in ("maze32", "rand32"):
        pr = bench.make_problem(kind, 4096, 1234)
        it = O.forward(pr.map_designs, pr.start_maps, pr.goal_maps, pr.map_designs, 0.5, 1024, mode="sm").iters.astype(float)
        s = pr.start_maps[:, 0].reshape(4096, -1).argmax(1)
        g = pr.goal_maps[:, 0].reshape(4096, -1).argmax(1)
        cheb = np.maximum(abs(s // 32 - g // 32), abs(s % 32 - g % 32)).astype(float)
        n = len(it)
        print(f"{kind}: steps mean {it.mean():.1f} max {it.max():.0f}; corr(chebyshev, steps) = {np.corrcoef(cheb, it)[0, 1]:.2f}")
        for lone, per in ((650, 240), (580, 190)):
            kw = dict(lone=lone, per=per)
            o_true, o_cheb = np.argsort(-it), np.argsort(-cheb, kind="stable")
            print(f"  lone {lone} / {per} per active wave: natural {launch_us(np.arange(n) % 1024, it, **kw):.0f} us, "
                  f"random {launch_us(np.random.default_rng(0).permutation(n) % 1024, it, **kw):.0f}, "
                  f"true snake {launch_us(snake(o_true), it, **kw):.0f}, true LPT {launch_us(lpt(o_true, it), it, **kw):.0f}, "
                  f"chebyshev snake {launch_us(snake(o_cheb), it, **kw):.0f}, chebyshev LPT {launch_us(lpt(o_cheb, cheb), it, **kw):.0f}, "
                  f"longest chain alone {it.max() * lone / 2.1e3:.0f}")


if __name__ == "__main__":
    main()
