#!/usr/bin/env python3
"""ms of DifferentiableAstar forward (with selection log) and backward under autograd on maps whose replay state lives in the HBM workspace
(run on the GPU box): costs = 0.8 map + 0.2 U(0,1) -- long searches --, eval mode.  DESIGN.md section 4.2 quotes it
(profiles/r06/probe_backward_large.jsonl; the single-launch, agent-scope form of rounds 2-5 on the same inputs: NOTES.md round 6)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neural-astar_amd")]
import torch  # noqa: E402

from neural_astar import ops  # noqa: E402
from neural_astar.planner.differentiable_astar import DifferentiableAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
for (H, B) in ((128, 256), (256, 64), (512, 64), (1024, 16)):
    pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=7)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    cost = (torch.from_numpy(syn.random_costs(B, H, H, seed=3)).to(dev) * 0.2 + m * 0.8).requires_grad_(True)
    da = DifferentiableAstar(0.5, 1.0).to(dev).eval()
    out = da(cost, s, g, m)
    out.histories.sum().backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    n = 5
    for _ in range(n):
        cost.grad = None
        e[0].record()
        out = da(cost, s, g, m)
        e[1].record()
        out.histories.sum().backward()
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
    it = da.last_iters
    print(json.dumps({"H": H, "W": H, "B": B, "cells": H * H, "state_in_lds": bool(ops.in_lds(H, H)), "forward_ms": tf / n, "backward_ms": tb / n,
                      "longest_search_steps": int(it.max()), "all_steps": int(it.sum())}), flush=True)
