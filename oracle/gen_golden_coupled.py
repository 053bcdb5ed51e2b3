#!/usr/bin/env python3
"""tests/golden/coupled_forward_g020.npz -- a member of the class in which the reference's FORWARD outputs of a map depend on the rest of its
batch (authoring container only: runs the reference).

The reference steps every map of a batch until ALL of them select their goal in the same step (differentiable_astar.py:219-225, :251); a map
that reached its goal earlier keeps being stepped, with its goal still on the open list.  When the goal's own expansion opens a neighbour whose
priority BEATS the goal's -- possible for g_ratio < 0.5 with an expensive goal cell (f(n) - f(goal) = (2 g_ratio - 1) c_goal + (1 - g_ratio)
(h0(n) + c_n)), for g_ratio = 1 with a zero-cost goal cell, or with negative costs; impossible for g_ratio in [0.5, 1) with costs >= 0 -- the
finished map goes on closing cells until the slowest map is done.  The kernels stop each map at its own goal = what the reference returns
for the map searched ALONE.  This file holds both: the batch outputs and the per-map-alone outputs of the reference on one small batch found
by scanning seeds (U(0,10) costs -- `const`-scaled encoder outputs --, g_ratio = 0.2)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
from neural_astar.utils import synthetic as syn  # noqa: E402
from oracle import gen_golden as GG  # noqa: E402

H, W, B, GR = 20, 24, 3, 0.2


def main():
    ref = GG.load_reference()
    for seed in range(1000):
        pr = syn.random_obstacle_maps(B, H, W, 0.1, seed=seed)
        cost = syn.random_costs(B, H, W, seed=seed + 7, hi=10.0)
        out, _ = GG.run_ref(ref, cost, pr.start_maps, pr.goal_maps, pr.map_designs, GR)
        alone_h, alone_p = [], []
        for b in range(B):
            o1, _ = GG.run_ref(ref, cost[b:b + 1], pr.start_maps[b:b + 1], pr.goal_maps[b:b + 1], pr.map_designs[b:b + 1], GR)
            alone_h.append(o1.histories[0].numpy())
            alone_p.append(o1.paths[0].numpy())
        alone_h, alone_p = np.stack(alone_h), np.stack(alone_p)
        hb = out.histories.numpy()
        extra = int((hb != alone_h).sum())
        if extra > 0 and np.array_equal(out.paths.numpy(), alone_p):
            print(f"seed {seed}: the batch run closes {extra} cell(s) more than the maps searched alone")
            GG.save("coupled_forward_g020", pr, cost, out, GR, extra={"hist_alone_bits": GG.pack(alone_h), "path_alone_bits": GG.pack(alone_p)})
            return
    raise SystemExit("no coupled case found")


if __name__ == "__main__":
    main()
