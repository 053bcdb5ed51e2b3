#!/usr/bin/env python3
"""tests/golden/coupled_*.npz (round 6) -- more members of the batch-coupled class, with the reference's own
selection logs and GRADIENTS (authoring container only: runs the reference by path).  Candidates are found with the oracle's literal
restatement (fast), the stored outputs are the reference's:

  coupled_grad_g020          4 maps 20x24, U(0,10) costs, g_ratio 0.2, eval mode: histories, paths, per-step selections, dL/dcost
  coupled_grad_train_g010    6 maps 16x16, g_ratio 0.1, training mode Tmax 0.25: the budget ends the loop while a finished map wanders
  coupled_signed_g050        4 maps 18x22, costs in [-2, 1), g_ratio 0.5 (the default!): costs below -1.001 reach the class at any g_ratio
                             (f(n) - f(goal) = 0.5 (h0(n) + c_n) < 0 needs c_n < -h0(n) <= -1.001)
  coupled_large140x150_g020  2 maps 140x150 (state larger than LDS: the hybrid kernel and the HBM-state replay), g_ratio 0.2; costs and the
                             upstream gradient are regenerated from seeds (stored: seeds + outputs)
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
from neural_astar.utils import synthetic as syn  # noqa: E402
from oracle import gen_golden as GG  # noqa: E402
from oracle import oracle as O  # noqa: E402


def n_coupled_cells(cost, pr, gr, T):
    d = O.forward(cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T, mode="dense", want_log=True)
    if d.status:
        return 0, d
    s = O.forward(cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T, mode="sm")
    return int((d.histories != s.histories).sum()), d


def emit(ref, name, pr, cost, gr, Tmax, training, up, extra=None, store_cost=True):
    out, grad = GG.run_ref(ref, cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, Tmax, training, want_grad=up, store=True)
    sel = GG.sel_from_intermediate(out)
    GG.save(name, pr, cost if store_cost else None, out, gr, Tmax, training, grad_up=up, grad=grad, sel=sel, extra=extra)


def main():
    ref = GG.load_reference()
    rng = np.random.Generator(np.random.PCG64(606))
    # 1. eval mode, g_ratio 0.2
    for seed in range(2000):
        pr = syn.random_obstacle_maps(4, 20, 24, 0.1, seed=seed)
        cost = syn.random_costs(4, 20, 24, seed=seed + 7, hi=10.0)
        n, d = n_coupled_cells(cost, pr, 0.2, 24 * 24)
        if n >= 3:
            emit(ref, "coupled_grad_g020", pr, cost, 0.2, 1.0, False, rng.standard_normal((4, 1, 20, 24)).astype(np.float32))
            break
    # 2. training mode: the budget ends the batch loop (some map never reaches its goal) while a finished map wanders
    for seed in range(4000):
        pr = syn.random_obstacle_maps(6, 16, 16, 0.2, seed=seed)
        cost = syn.random_costs(6, 16, 16, seed=seed + 7, hi=10.0)
        T = int(0.25 * 16 * 16)
        n, d = n_coupled_cells(cost, pr, 0.1, T)
        if n >= 2 and d.t_batch == T - 1:
            gi = pr.goal_maps.reshape(6, -1).argmax(1)
            lg = d.sel_log[:, :T]
            wandering = [(lg[b] == gi[b]).any() and lg[b, -1] != gi[b] for b in range(6)]  # reached its goal, not on it when the budget ends
            if any(wandering):
                emit(ref, "coupled_grad_train_g010", pr, cost, 0.1, 0.25, True, rng.standard_normal((6, 1, 16, 16)).astype(np.float32))
                break
    # 3. negative costs at the default g_ratio
    for seed in range(4000):
        pr = syn.random_obstacle_maps(4, 18, 22, 0.1, seed=seed)
        cost = syn.random_costs(4, 18, 22, seed=seed + 7, lo=-2.0, hi=1.0)
        n, d = n_coupled_cells(cost, pr, 0.5, 22 * 22)
        if n >= 2:
            emit(ref, "coupled_signed_g050", pr, cost, 0.5, 1.0, False, np.random.Generator(np.random.PCG64(607)).standard_normal((4, 1, 18, 22)).astype(np.float32))
            break
    # 4. a size whose state does not fit LDS
    H, W = 140, 150
    for seed in range(200):
        pr = syn.random_obstacle_maps(2, H, W, 0.1, seed=seed)
        cost = syn.random_costs(2, H, W, seed=seed + 7, hi=10.0)
        n, d = n_coupled_cells(cost, pr, 0.2, W * W)
        if n >= 2:
            up = np.random.Generator(np.random.PCG64(seed + 99)).standard_normal((2, 1, H, W)).astype(np.float32)
            out, grad = GG.run_ref(ref, cost, pr.start_maps, pr.goal_maps, pr.map_designs, 0.2, 1.0, False, want_grad=up)
            GG.save("coupled_large140x150_g020", pr, None, out, 0.2, extra={"cost_seed": np.int64(seed + 7), "cost_hi": np.float32(10.0),
                                                                            "up_seed": np.int64(seed + 99), "grad_cost_ref": grad.astype(np.float32)})
            break


if __name__ == "__main__":
    main()
