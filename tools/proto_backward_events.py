#!/usr/bin/env python3
"""CPU prototype (dev tool, pure Python, small cases only) of the round-2 backward algorithm: replay the search from the forward's
selection log and account the softmax gradient per EVENT (open / re-key / close) instead of per step over the whole open list.

    dL/dcost[i] = kfac * sum over the intervals [t0, t1] during which i sat on the open list with a fixed v_i = exp(-q_i):
                  v_i * (G_i * (A(t1) - A(t0)) - (B(t1) - B(t0)))
    A(t) = sum_{tau <= t} w_tau / S_tau,   B(t) = sum_{tau <= t} w_tau * D_tau / S_tau^2,
    S_t = sum_open v,  D_t = sum_open v * G      (maintained incrementally, in double: <= 9 cells change per step)

It mirrors what nastar_backward_replay_kernel does on the device and is checked here against the golden gradients the reference's
autograd produced (tests/golden/grad_*.npz).  Usage: python tools/proto_backward_events.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32


def h0(r, c, gr, gc):
    a = f32(r - gr)
    b = f32(c - gc)
    dr, dc = abs(a), abs(b)
    cheb = f32(f32(dr + dc) - min(dr, dc))
    euc = np.sqrt(f32(f32(a * a) + f32(b * b)))
    return f32(cheb + f32(f32(0.001) * euc))


def backward_map(Gup, cost, passable, s_idx, g_idx, sel, iters, extra, g_ratio, H, W):
    HW = H * W
    gr = f32(g_ratio)
    omg = f32(1.0 - g_ratio)
    sqrtW = f32(np.sqrt(float(W)))
    kfac = f32(omg * f32(f32(-1.0) / sqrtW))
    Gv = Gup.reshape(-1).astype(f32).copy()
    cost = cost.reshape(-1)
    passable = passable.reshape(-1)
    grr, gcc = divmod(g_idx, W)
    if extra > 0:
        Gv[g_idx] = 0.0
    g = np.where(passable != 0, np.inf, -np.inf).astype(f32)
    vcur = np.zeros(HW, f32)  # exp(-q) of open cells
    stampA = np.zeros(HW, np.float64)
    stampB = np.zeros(HW, np.float64)
    acc = np.zeros(HW, f32)
    S = D = A = B = 0.0

    def vkey(i, gi):
        r, c = divmod(i, W)
        hh = f32(omg * f32(h0(r, c, grr, gcc) + cost[i]))
        f = f32(f32(gr * gi) + hh)
        q = f32(f / sqrtW)
        return f32(np.exp(f32(-q)))

    def flush(i):
        nonlocal S, D
        dA = f32(A - stampA[i])
        dB = f32(B - stampB[i])
        acc[i] += f32(f32(kfac * vcur[i]) * f32(f32(Gv[i] * dA) - dB))
        S -= float(vcur[i])
        D -= float(f32(Gv[i] * vcur[i]))

    def open_cell(i, gi):
        nonlocal S, D
        g[i] = gi
        vcur[i] = vkey(i, gi)
        S += float(vcur[i])
        D += float(f32(Gv[i] * vcur[i]))
        stampA[i], stampB[i] = A, B

    open_cell(s_idx, f32(0.0))
    for t in range(iters):
        rS = 1.0 / S
        A += rS
        B += D * rS * rS
        s = int(sel[t])
        goal_step = s == g_idx
        if goal_step and extra <= 0:
            break
        g2 = f32(g[s] + cost[s])
        if not goal_step:
            flush(s)
            g[s] = -np.inf
        r, c = divmod(s, W)
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                if dr == 0 and dc == 0:
                    continue
                rr, cc = r + dr, c + dc
                if not (0 <= rr < H and 0 <= cc < W):
                    continue
                n = rr * W + cc
                if n == s or not (g[n] > g2):
                    continue
                if np.isfinite(g[n]):
                    flush(n)
                open_cell(n, g2)
        if goal_step:
            rS = 1.0 / S
            A += extra * rS
            B += extra * D * rS * rS
            break
    for i in range(HW):
        if np.isfinite(g[i]):
            flush(i)
    return acc.reshape(H, W)


def main():
    worst = 0.0
    for name in G.names():
        if not name.startswith("grad_"):
            continue
        gd = G.load(name)
        if gd.H * gd.W > 4200 and "fixture64_eval" not in name and gd.H * gd.W > 10000:
            pass
        o = O.forward(gd.cost_maps, gd.start_maps, gd.goal_maps, gd.passable, gd.g_ratio, gd.max_iters, mode="sm", want_log=True)
        t_batch = int(o.iters.max()) - 1
        got = np.zeros((gd.B, gd.H, gd.W), f32)
        for b in range(gd.B):
            s_idx = int(gd.start_maps[b].reshape(-1).argmax())
            g_idx = int(gd.goal_maps[b].reshape(-1).argmax())
            extra = t_batch - (int(o.iters[b]) - 1)
            got[b] = backward_map(gd.grad_up[b, 0], gd.cost_maps[b, 0], gd.passable[b, 0], s_idx, g_idx, o.sel_log[b],
                                  int(o.iters[b]), extra, gd.g_ratio, gd.H, gd.W)
        err = float(np.abs(got - gd.grad_cost[:, 0]).max())
        scale = max(1.0, float(np.abs(gd.grad_cost).max()))
        worst = max(worst, err / scale)
        print(f"{name:34s} max|err| {err:.3e}  (max|grad| {np.abs(gd.grad_cost).max():.3e})  {'ok' if err <= 1e-5 * scale else 'FAIL'}")
    print("worst relative-to-scale error:", worst)


if __name__ == "__main__":
    main()
