#!/usr/bin/env python3
"""How often does the reference leave the quotient order?  (authoring container only: RUNS THE REFERENCE; test infrastructure)

The kernels select the first-index arg-min over the open list of q = fl(f / fl32(sqrt(W))).  The reference
selects the first arg-max of y = exp(-q) * open / sum (differentiable_astar.py:55-74, :206-209).  The two agree unless `exp` or the
division by the row sum maps two DISTINCT quotients to one float -- possible only where exp(-q) has fewer distinct values than q, i.e.
q < ~2 (f < 2 sqrt(W)) -- AND the cell with the larger quotient has the lower flat index.  This script measures it on the live reference:

  * a TorchDispatchMode records the argument of every aten.exp (= -q, the reference's own tensor),
  * `_st_softmax_noexp` is wrapped: per map and step it compares the reference's own pick with the quotient rule and counts
      steps            active selection steps (maps still searching)
      low_f            ... whose best quotient is < 2 (the regime where a merge is possible at all)
      merged           ... where some open cell with a DIFFERENT quotient than the best shares the maximal y (two quotients, one y)
      divergent        ... where the reference's pick is not the first-index arg-min of q (the outputs may differ from here on)

Usage: python oracle/tie_census.py [--out profiles/r05/tie_census.json]
       python oracle/tie_census.py --encoder-costs [--out profiles/r06/tie_census_encoder_costs.json]
           round 6: the regime DESIGN.md 2.5 names as the class's exposure -- cost maps of the reference's SHIPPED checkpoint through the
           reference's own CNN encoder (oracle/gen_golden.py: cnn_cost_maps), scaled by `const` in {1, 2, 5, 10} as NeuralAstar(const=...) scales
           them (astar.py:150-152, encoder.py:32-34), at g_ratio 0.5 / 0.2 / 0.8, plus U(0, 10) costs
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neural-astar_amd"))
REF = "/root/reference/src/neural_astar/planner/differentiable_astar.py"

from neural_astar.utils import synthetic as syn  # noqa: E402


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_differentiable_astar", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class ExpTap(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.last = None

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if func is torch.ops.aten.exp.default:
            self.last = args[0]
        return func(*args, **(kwargs or {}))


def census(ref, cost, start, goal, passable, g_ratio=0.5):
    counts = dict(steps=0, low_f=0, merged=0, divergent=0)
    tap = ExpTap()
    orig = ref._st_softmax_noexp
    done = None
    goal_idx = torch.from_numpy(goal.reshape(goal.shape[0], -1).argmax(1))

    def wrapped(val):
        nonlocal done
        y_st = orig(val)
        B = val.shape[0]
        v = val.reshape(B, -1)
        negq = tap.last.reshape(B, -1)               # -q, the argument of the reference's own exp
        openm = v > 0
        q = torch.where(openm, -negq, torch.full_like(negq, float("inf")))
        qmin, qarg = q.min(1)                        # torch.min: first index among equal values
        y = v / v.sum(-1, keepdim=True)
        ymax, yarg = y.max(1)
        active = torch.ones(B, dtype=torch.bool) if done is None else ~done
        tie = (y == ymax[:, None]) & openm & (q != qmin[:, None])
        counts["steps"] += int(active.sum())
        counts["low_f"] += int((active & (qmin < 2.0)).sum())
        counts["merged"] += int((active & tie.any(1)).sum())
        counts["divergent"] += int((active & (yarg != qarg)).sum())
        reached = yarg == goal_idx
        done = reached if done is None else (done | reached)
        return y_st

    ref._st_softmax_noexp = wrapped
    try:
        m = ref.DifferentiableAstar(g_ratio=g_ratio, Tmax=1.0).eval()
        with torch.no_grad(), tap:
            m(*(torch.from_numpy(x) for x in (cost, start, goal, passable)))
    finally:
        ref._st_softmax_noexp = orig
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05", "tie_census.json"))
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--encoder-costs", action="store_true")
    args = ap.parse_args()
    ref = load_reference()
    torch.set_num_threads(os.cpu_count() or 1)
    res = {}
    B = args.batch
    if args.encoder_costs:
        sys.path[:] = [q for q in sys.path if os.path.abspath(q or ".") != os.path.join(ROOT, "oracle")]  # (the script's own directory shadows the package)
        sys.path.insert(0, ROOT)
        from oracle import gen_golden as GG
        if args.out.endswith(os.path.join("r05", "tie_census.json")):
            args.out = os.path.join(ROOT, "profiles", "r06", "tie_census_encoder_costs.json")
        total = dict(steps=0, low_f=0, merged=0, divergent=0)
        for wname, pr in (("maze32", syn.maze_maps(B, 32, seed=1234)), ("rand32", syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1234))):
            enc = GG.cnn_cost_maps((pr.map_designs, pr.start_maps, pr.goal_maps))  # sigmoid outputs in (0, 1): the shipped mazes_032_moore_c8 model
            for const in (1.0, 2.0, 5.0, 10.0):
                for gr in (0.5, 0.2, 0.8):
                    c = census(ref, (enc * np.float32(const)).astype(np.float32), pr.start_maps, pr.goal_maps, pr.map_designs, g_ratio=gr)
                    name = f"{wname}, shipped-checkpoint CNN cost x const {const:g}, g_ratio {gr}"
                    res[name] = c
                    for k in total:
                        total[k] += c[k]
                    print(name, c, flush=True)
            for gr in (0.5, 0.2, 0.8):
                c = census(ref, syn.random_costs(B, 32, 32, seed=4321, hi=10.0), pr.start_maps, pr.goal_maps, pr.map_designs, g_ratio=gr)
                name = f"{wname}, cost ~ U(0,10), g_ratio {gr}"
                res[name] = c
                for k in total:
                    total[k] += c[k]
                print(name, c, flush=True)
        res["_total"] = total
        res["_meta"] = {"torch": torch.__version__, "threads": torch.get_num_threads(), "batch": B,
                        "cpu_flags": [f for f in ("avx2", "avx512f") if f in open("/proc/cpuinfo").read()]}
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
        print("TOTAL", total)
        return
    cases = []
    for name, pr in (("maze32 (bench headline batch, seed 1234)", syn.maze_maps(B, 32, seed=1234)),
                     ("rand32 (seed 1234)", syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1234)),
                     ("rand64 (seed 1234, 1024 maps)", syn.random_obstacle_maps(min(B, 1024), 64, 64, 0.20, seed=1234))):
        cases.append((name + ", cost = map (VanillaAstar)", pr, pr.map_designs))
    pr = syn.maze_maps(B, 32, seed=1234)
    cases.append(("maze32, cost ~ U(0,1) (encoder-like)", pr, syn.random_costs(B, 32, 32, seed=4321)))
    pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1234)
    cases.append(("rand32, cost ~ U(0,1) (encoder-like)", pr, syn.random_costs(B, 32, 32, seed=4321)))
    cases.append(("rand32, cost ~ U(0,0.05) (a confident encoder: small costs, f < 2 sqrt(W) for longer)", pr, syn.random_costs(B, 32, 32, seed=99, hi=0.05)))
    for name, pr, cost in cases:
        c = census(ref, cost, pr.start_maps, pr.goal_maps, pr.map_designs)
        res[name] = c
        print(name, c, flush=True)
    res["_meta"] = {"torch": torch.__version__, "threads": torch.get_num_threads(),
                    "cpu_flags": [f for f in ("avx2", "avx512f") if f in open("/proc/cpuinfo").read()],
                    "definition": __doc__.split("Usage")[0].strip()}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
