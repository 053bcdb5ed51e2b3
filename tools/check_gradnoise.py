#!/usr/bin/env python3
"""The kernels' dL/dcost on tests/golden/gradnoise_u10_45x47.npz (run on the GPU box): distance to the reference's fp32 autograd gradient and to
the reference's own graph evaluated in float64.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as G  # noqa: E402
from neural_astar.planner.differentiable_astar import DifferentiableAstar  # noqa: E402

dev = torch.device("cuda:0")
g = G.load("gradnoise_u10_45x47")
g64 = np.load(os.path.join(G.GOLDEN_DIR, "gradnoise_u10_45x47.npz"))["grad_f64"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
m = DifferentiableAstar(g_ratio=g.g_ratio, Tmax=g.Tmax).to(dev).eval()
cost = t(g.cost_maps).requires_grad_(True)
out = m(cost, t(g.start_maps), t(g.goal_maps), t(g.passable))
(out.histories * t(g.grad_up)).sum().backward()
got = cost.grad.cpu().numpy()
scale = max(1.0, float(np.abs(g.grad_cost).max()))
print(json.dumps({"forward_equal": bool(np.array_equal(out.histories.detach().cpu().numpy(), g.histories) and np.array_equal(out.paths.cpu().numpy(), g.paths)),
                  "kernel_vs_reference_fp64": float(np.abs(got - g64).max()) / scale, "kernel_vs_reference_fp32": float(np.abs(got - g.grad_cost).max()) / scale,
                  "reference_fp32_vs_fp64": float(np.abs(g.grad_cost - g64).max()) / scale, "scale": scale}))
