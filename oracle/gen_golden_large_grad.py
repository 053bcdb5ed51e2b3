#!/usr/bin/env python3
"""tests/golden/wide_grad_260x270.npz (round 6) -- the reference's own forward outputs and autograd gradient on maps ABOVE 65,519 cells, the
size from which the replay backward needs 32-bit history stamps (authoring container only: runs the reference by path).  2 maps 260x270, 10 %
obstacles, U(0,1) costs, g_ratio 0.5, eval mode; costs and the upstream gradient are regenerated from seeds (stored: seeds + outputs)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
from neural_astar.utils import synthetic as syn  # noqa: E402
from oracle import gen_golden as GG  # noqa: E402


def main():
    ref = GG.load_reference()
    H, W, seed = 260, 270, 11
    pr = syn.random_obstacle_maps(2, H, W, 0.1, seed=seed)
    cost = syn.random_costs(2, H, W, seed=seed + 7, hi=1.0)
    up = np.random.Generator(np.random.PCG64(seed + 99)).standard_normal((2, 1, H, W)).astype(np.float32)
    out, grad = GG.run_ref(ref, cost, pr.start_maps, pr.goal_maps, pr.map_designs, 0.5, 1.0, False, want_grad=up)
    GG.save("wide_grad_260x270", pr, None, out, 0.5, extra={"cost_seed": np.int64(seed + 7), "cost_hi": np.float32(1.0), "up_seed": np.int64(seed + 99),
                                                              "grad_cost_ref": grad.astype(np.float32)})
    print("histories", int(out.histories.sum()), "grad max", float(np.abs(grad).max()))


if __name__ == "__main__":
    main()
