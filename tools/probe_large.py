#!/usr/bin/env python3
"""ns per search step of the large-map kernels (run on the GPU box): hybrid (open list in LDS) vs round 4's all-HBM kernel (flags 512), and the
LDS kernel's step on the largest LDS-resident size for scale.  One map = the latency of the serial chain; 256 maps = what the chip sustains."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neural_astar import ops  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")


def run(H, B, flags, cost_kind, reps=5):
    pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=7)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    c = m if cost_kind == "map" else torch.from_numpy(syn.random_costs(B, H, H, seed=5)).to(dev)
    out = ops.search_nograd(c, s, g, m, 0.5, H * H, flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ops.search_nograd(c, s, g, m, 0.5, H * H, flags=flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    it = out[2].cpu().numpy()
    return {"H": H, "B": B, "flags": flags, "cost": cost_kind, "launch_ms": ms, "max_iters": int(it.max()), "sum_iters": int(it.sum()),
            "ns_per_step_of_longest": ms * 1e6 / it.max(), "steps_per_s": it.sum() / (ms * 1e-3)}


def variants():
    """A/B of the hybrid kernel's variants (include/nastar.h NASTAR_FLAG_HYBRID_*): step time and equality of every output with the default"""
    SC1, NOFENCE, SCALAR, BALLOT = 2048, 4096, 8192, 16384
    combos = [0, SC1, NOFENCE, SCALAR, SCALAR | NOFENCE, BALLOT, BALLOT | NOFENCE]
    for H, B, ck in ((256, 256, "map"), (512, 24, "map"), (256, 256, "uniform")):  # (generating 256 maps of 512x512 takes minutes of host time)
        pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=7)
        m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
        c = m if ck == "map" else torch.from_numpy(syn.random_costs(B, H, H, seed=5)).to(dev)
        ref = ops.search_nograd(c, s, g, m, 0.5, H * H, want_log=True)
        for flags in combos:
            r = run(H, B, flags, ck, reps=3)
            out = ops.search_nograd(c, s, g, m, 0.5, H * H, want_log=True, flags=flags)
            it = ref[2].long()
            mask = torch.arange(H * H, device=dev)[None, :] < it[:, None]
            r["equal_to_default"] = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
                                         and torch.equal(out[3], ref[3]) and torch.equal(out[4][mask], ref[4][mask]))
            print(json.dumps(r), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "variants":
    variants()
    sys.exit(0)

for H in (128, 256, 512):
    for B in (1, 256):
        for flags in ((0,) if H == 128 else (0, 512)):
            for ck in ("map", "uniform"):
                print(json.dumps(run(H, B, flags, ck)), flush=True)
