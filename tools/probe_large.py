#!/usr/bin/env python3
"""ns per search step of the large-map kernel (run on the GPU box): one map = the latency of the serial chain; 256 maps = what the chip sustains.
Round 6: `python tools/probe_large.py mid` compares, on sides 72 .. 128 (LDS-resident sizes whose compiled loop scans several chunk entries per
lane), the LDS kernel with the hybrid kernel (NASTAR_HYBRID_FROM_CELLS=4097 in a child process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neural_astar import ops  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")


def run(H, B, cost_kind, reps=5, tag=""):
    pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=7)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    c = m if cost_kind == "map" else torch.from_numpy(syn.random_costs(B, H, H, seed=5)).to(dev)
    out = ops.search_nograd(c, s, g, m, 0.5, H * H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ops.search_nograd(c, s, g, m, 0.5, H * H)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    it = out[2].cpu().numpy()
    return {"H": H, "B": B, "kernel": tag or ("lds" if ops.in_lds(H, H) else "hybrid"), "cost": cost_kind, "launch_ms": ms, "max_iters": int(it.max()),
            "sum_iters": int(it.sum()), "ns_per_step_of_longest": ms * 1e6 / it.max(), "steps_per_s": it.sum() / (ms * 1e-3),
            "hist_sum": int(out[0].sum().item())}


if len(sys.argv) > 1 and sys.argv[1] == "mid":
    if os.environ.get("NASTAR_HYBRID_FROM_CELLS"):
        for H in (72, 80, 96, 112, 128):
            for B in (1, 64, 1024):
                print(json.dumps(run(H, B, "map")), flush=True)
                print(json.dumps(run(H, B, "uniform")), flush=True)
    else:
        for env in ({}, {"NASTAR_HYBRID_FROM_CELLS": "4097"}):
            subprocess.run([sys.executable, __file__, "mid"], env=dict(os.environ, NASTAR_HYBRID_FROM_CELLS=env.get("NASTAR_HYBRID_FROM_CELLS", str(1 << 40))))
    sys.exit(0)

for H in (128, 256, 512, 1024):
    for B in (1, 64 if H == 1024 else 256):
        for ck in ("map", "uniform"):
            print(json.dumps(run(H, B, ck)), flush=True)
