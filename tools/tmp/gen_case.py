"""regenerate the inputs of one run_module fuzz case on the CPU (oracle only) and save them"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np
from neural_astar import ops
from neural_astar.utils import synthetic as syn
from oracle import oracle as O

def gen(seed, target, grad_frac=0.35, large_frac=0.1):
    rng = np.random.default_rng(seed)
    for case in range(target + 1):
        large = rng.random() < large_frac
        if large:
            H, W = int(rng.integers(112, 151)), int(rng.integers(112, 201))
        else:
            H, W = (int(rng.integers(4, 40)), int(rng.integers(4, 40))) if rng.random() < 0.8 else (int(rng.choice([16, 32])),) * 2
        B = int(rng.integers(2, 4 if large else 6))
        pr = syn.random_obstacle_maps(B, H, W, float(rng.choice([0.0, 0.1, 0.25])), seed=int(rng.integers(1 << 30)))
        kind = str(rng.choice(["map", "u01", "u10", "u10", "zeros", "signed", "signed2"]))
        if kind == "map":
            cost = pr.map_designs
        elif kind == "zeros":
            cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30))) * (rng.random((B, 1, H, W)) < 0.5).astype(np.float32)
        elif kind == "signed":
            cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)), lo=-0.5, hi=1.0)
        elif kind == "signed2":
            cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)), lo=-2.0, hi=1.0)
        else:
            cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)), hi=1.0 if kind == "u01" else 10.0)
        gr = float(rng.choice([0.5, 0.2, 0.0, 0.8, 1.0, 0.3]))
        train = bool(rng.random() < 0.3)
        Tmax = float(rng.choice([0.25, 0.5])) if train else 1.0
        T = int(Tmax * W * W) if train else W * W
        if T < 1:
            continue
        o = O.forward(cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T, mode="dense")
        if o.status:
            continue
        with_grad = kind != "map" and rng.random() < grad_frac and H * W <= 32000
        mode = [True, "deferred", False][int(rng.integers(0, 3))]
        if with_grad:
            up = rng.standard_normal((B, 1, H, W)).astype(np.float32)
        print(case, H, W, B, kind, gr, train, Tmax, mode, with_grad, flush=True)
        if case == target:
            osm = O.forward(cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T, mode="sm")
            np.savez_compressed(os.path.join(ROOT, "tools", "tmp", f"case_{seed}_{target}.npz"), cost=cost, start=pr.start_maps, goal=pr.goal_maps,
                                passable=pr.map_designs, g_ratio=gr, Tmax=Tmax, T=T, train=train, up=up if with_grad else 0,
                                histories=o.histories, paths=o.paths, iters=o.iters, sm_histories=osm.histories, sm_iters=osm.iters)
            print("saved; dense iters", o.iters, "sm iters", osm.iters, "equal hist", np.array_equal(o.histories, osm.histories))

gen(int(sys.argv[1]), int(sys.argv[2]))
