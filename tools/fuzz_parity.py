#!/usr/bin/env python3
"""Randomised parity sweep (run on the GPU box): the search through the C ABI against the oracle's state-machine restatement on random map
sizes (LDS-resident, compiled and hand-scheduled instantiations, and the hybrid large-map kernel just above the LDS limit), obstacle
densities, cost kinds (map / U(0,1) / U(0,10)), g_ratio and budgets, with and without a selection log and a random placement.  Prints one
JSON line per failing case and a summary; exit code 1 on any mismatch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from neural_astar import ops  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")
sizes = [(16, 16), (32, 32), (64, 64), (12, 12), (7, 5), (20, 45), (64, 128), (100, 100), (128, 128), (130, 131), (135, 150), (140, 140), (150, 200), (33, 31),
         (96, 96), (48, 80), (5, 200), (200, 5), (1, 40), (40, 1), (129, 129), (131, 127)]


def run(seed=20260926, N=160, big_frac=0.15, verbose=True):
  """-> (cases per kernel family, list of failing case descriptions)"""
  rng = np.random.default_rng(seed)
  bad = []
  stats = {}
  for case in range(N):
    if case < 2 * len(sizes):
        H, W = sizes[case % len(sizes)]
    elif rng.random() < big_frac:  # above the LDS limit: the hybrid kernel (fill / search / store launches)
        H, W = int(rng.integers(120, 260)), int(rng.integers(120, 260))
    elif rng.random() < 0.3:       # the hand-scheduled instruction streams (general and unit-cost layouts)
        H = W = int(rng.choice([16, 32, 64]))
    else:
        H, W = int(rng.integers(3, 160)), int(rng.integers(3, 160))
    B = int(rng.integers(1, 9))
    p = float(rng.choice([0.0, 0.1, 0.2, 0.3]))
    try:
        pr = syn.random_obstacle_maps(B, H, W, p, seed=int(rng.integers(1 << 30)))
    except Exception:
        continue
    kind = str(rng.choice(["map", "u01", "u10", "signed", "zeros"]))
    if kind == "map":
        cost = pr.map_designs
    elif kind == "signed":  # negative costs: the round-2 instruction stream with its order-preserving key transform (16 / 32 / 64), generic paths elsewhere
        cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)), lo=-0.5, hi=1.0)
    elif kind == "zeros":   # many exactly-zero costs: ties in g as well as in h
        cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30))) * (rng.random((B, 1, H, W)) < 0.5).astype(np.float32)
    else:
        cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)), hi=1.0 if kind == "u01" else 10.0)
    maps = pr.map_designs.copy()
    starts = pr.start_maps
    variant = str(rng.choice(["plain", "plain", "plain", "start_on_obstacle", "start_is_goal"]))
    if variant == "start_on_obstacle":  # the start is expanded even on an obstacle (reference :187: open = start)
        maps.reshape(B, -1)[np.arange(B), pr.start_maps.reshape(B, -1).argmax(1)] = 0.0
        if kind == "map":
            cost = maps
    elif variant == "start_is_goal":
        starts = pr.goal_maps.copy()
    pr = syn.Problems(maps, starts, pr.goal_maps)
    unit = kind == "map" and H == W and W in (32, 64) and rng.random() < 0.5  # the unit-cost LDS layout (NASTAR_FLAG_UNIT_COST)
    gr = float(rng.choice([0.5, 0.5, 0.5, 0.2, 0.8, 0.0, 1.0]))
    T = W * W if rng.random() < 0.7 else max(1, int(rng.choice([0.05, 0.25, 0.5]) * W * W))
    log = bool(rng.random() < 0.5) and not unit
    in_lds = ops.in_lds(H, W)
    order = None
    if in_lds and rng.random() < 0.4:
        order = torch.from_numpy(rng.permutation(B).astype(np.int32)).to(dev)
    c, s, g, m = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (cost, pr.start_maps, pr.goal_maps, pr.map_designs))
    try:
        hist, paths, iters, status, sel = ops.search_nograd(c, s, g, c if kind == "map" else m, gr, T, want_log=log, order=order, check_order=False,
                                                            flags=ops.FLAG_UNIT_COST if unit else 0)
        torch.cuda.synchronize()
        o = O.forward(cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T, mode="sm", want_log=log)
        ok = (np.array_equal(hist.cpu().numpy(), o.histories) and np.array_equal(paths.cpu().numpy(), o.paths) and np.array_equal(iters.cpu().numpy(), o.iters)
              and bool((status == 0).all()))
        if ok and log:
            lg, it = sel.cpu().numpy(), o.iters
            ok = all(np.array_equal(lg[b, :it[b]], o.sel_log[b, :it[b]]) for b in range(B))
    except Exception as e:  # noqa: BLE001
        ok = False
        if verbose:
            print(json.dumps({"case": case, "error": f"{type(e).__name__}: {e}"[:300]}), flush=True)
    key = ("unit" if unit else "lds") if in_lds else "hybrid"
    stats[key] = stats.get(key, 0) + 1
    if not ok:
        d = {"case": case, "H": H, "W": W, "B": B, "p": p, "cost": kind, "variant": variant, "unit": bool(unit), "g_ratio": gr, "max_iters": T, "log": log,
             "placed": order is not None, "in_lds": in_lds}
        bad.append(d)
        if verbose:
            print(json.dumps(d), flush=True)
  return stats, bad


def run_backward(seed=11, N=40, verbose=True, large=False):
    """dL/dcost of the replay backward (through DifferentiableAstar under autograd) against the oracle's literal reverse-mode restatement on
    random small maps (``large``: 64 .. 140 cells per side -- the generic LDS replay and the one with its state in the HBM workspace; the
    oracle needs seconds per map there), training and eval budgets, random upstream gradients; tolerance 1e-5 (north_star).
    -> (cases, failures)"""
    from neural_astar.planner.differentiable_astar import DifferentiableAstar
    rng = np.random.default_rng(seed)
    small = [(16, 16), (32, 32), (12, 12), (7, 5), (20, 45), (33, 31), (24, 40), (48, 48), (9, 30)]
    if large:
        small = [(64, 64), (96, 96), (100, 100), (110, 130), (140, 90), (128, 128)]
    bad, n = [], 0
    for case in range(N):
        H, W = small[case % len(small)] if (large or case < 2 * len(small)) else (int(rng.integers(4, 50)), int(rng.integers(4, 50)))
        B = int(rng.integers(1, 3 if large else 5))
        pr = syn.random_obstacle_maps(B, H, W, float(rng.choice([0.0, 0.15, 0.3])), seed=int(rng.integers(1 << 30)))
        cost_np = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)))
        gr = float(rng.choice([0.5, 0.5, 0.2, 0.8]))
        train = bool(rng.random() < 0.5)
        Tmax = float(rng.choice([0.05, 0.1] if large else [0.25, 0.5, 1.0])) if train else 1.0
        T = int((Tmax if train else 1.0) * W * W)
        if T < 1:
            continue
        up = rng.standard_normal((B, 1, H, W)).astype(np.float32)
        da = DifferentiableAstar(gr, Tmax).to(dev).train(train)
        cost = torch.from_numpy(cost_np).to(dev).requires_grad_(True)
        s, g, m = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pr.start_maps, pr.goal_maps, pr.map_designs))
        out = da(cost, s, g, m)
        (out.histories * torch.from_numpy(up).to(dev)).sum().backward()
        ref = O.backward(up, cost_np, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T)
        err = float(np.abs(cost.grad[:, 0].cpu().numpy() - ref).max())
        n += 1
        if not err <= 1e-5 * max(1.0, float(np.abs(ref).max())):
            d = {"case": case, "H": H, "W": W, "B": B, "g_ratio": gr, "train": train, "Tmax": Tmax, "err": err}
            bad.append(d)
            if verbose:
                print(json.dumps(d), flush=True)
    return n, bad


def run_module(seed=31, N=60, verbose=True, grad_frac=0.35, large_frac=0.1, large_hw=((112, 151), (112, 201))):
    """DifferentiableAstar.forward() against the oracle's LITERAL restatement of the reference's batch loop, on random batches incl. the cost
    kinds / g_ratio values of the batch-coupled class.  Round 6: every mode -- same-call verdict, deferred (verdict
    collected before the outputs are read), unchecked (where the class is reachable with costs >= 0 the exact pipeline runs anyway) --, a share
    of the cases UNDER AUTOGRAD (dL/dcost against the oracle's literal reverse mode, 1e-5) and a share on maps whose state does not fit LDS
    (the hybrid kernel's lock-step modes; up to 150x200 -- the oracle's dense restatement needs tens of seconds for one of those).
    -> (cases, batches in the coupled class, gradient cases, failures)"""
    import warnings
    from neural_astar.planner.differentiable_astar import DifferentiableAstar
    rng = np.random.default_rng(seed)
    bad, n, reruns, ngrad = [], 0, 0, 0
    for case in range(N):
        large = rng.random() < large_frac
        if large:
            H, W = int(rng.integers(*large_hw[0])), int(rng.integers(*large_hw[1]))
        else:
            H, W = (int(rng.integers(4, 40)), int(rng.integers(4, 40))) if rng.random() < 0.8 else (int(rng.choice([16, 32])),) * 2
        B = int(rng.integers(2, 4 if large else 6))
        pr = syn.random_obstacle_maps(B, H, W, float(rng.choice([0.0, 0.1, 0.25])), seed=int(rng.integers(1 << 30)))
        kind = str(rng.choice(["map", "u01", "u10", "u10", "zeros", "signed", "signed2"]))
        if kind == "map":
            cost = pr.map_designs
        elif kind == "zeros":
            cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30))) * (rng.random((B, 1, H, W)) < 0.5).astype(np.float32)
        elif kind == "signed":  # negative costs (the 16 / 32 / 64 streams take the key transform)
            cost = syn.random_costs(B, H, W, seed=int(rng.integers(1 << 30)), lo=-0.5, hi=1.0)
        elif kind == "signed2":  # costs below -1: the batch-coupled class at ANY g_ratio (found through the status summary for g_ratio in [0.5, 1))
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
            continue  # (negative costs can empty an open list: the reference crashes there)
        osm = O.forward(cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T, mode="sm")
        reruns += int(not np.array_equal(o.histories, osm.histories))
        with_grad = kind != "map" and rng.random() < grad_frac and H * W <= 32000
        mode = [True, "deferred", False][int(rng.integers(0, 3))]
        if mode is False and not ops.coupling_possible(gr):
            mode = True  # (the documented gap: unchecked calls read nothing back, and only costs below -1 reach the class at this g_ratio)
        if mode == "deferred" and with_grad and not ops.coupling_possible(gr):
            mode = True  # (deferred + autograd + costs below -1: refused loudly by design)
        da = DifferentiableAstar(gr, Tmax, check_solvable=mode).to(dev).train(train)
        c, s, g, m = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (cost, pr.start_maps, pr.goal_maps, pr.map_designs))
        err = None
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                if with_grad:
                    up = rng.standard_normal((B, 1, H, W)).astype(np.float32)
                    cg = c.clone().requires_grad_(True)
                    out = da(cg, s, g, m)
                    (out.histories * torch.from_numpy(up).to(dev)).sum().backward()
                    da.raise_if_unsolvable()
                    ref = O.backward(up, cost, pr.start_maps, pr.goal_maps, pr.map_designs, gr, T)
                    gerr = float(np.abs(cg.grad[:, 0].cpu().numpy() - ref).max())
                    ngrad += 1
                    if not gerr <= 1e-5 * max(1.0, float(np.abs(ref).max())):
                        err = f"grad err {gerr:.3e} (scale {float(np.abs(ref).max()):.3e})"
                else:
                    with torch.no_grad():
                        out = da(c, s, g, c if kind == "map" else m)
                        da.raise_if_unsolvable()
            ok = np.array_equal(out.histories[:, 0].detach().cpu().numpy(), o.histories) and np.array_equal(out.paths[:, 0].cpu().numpy(), o.paths)
            if not ok:
                err = "histories / paths differ from the literal batch loop"
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        n += 1
        if err:
            d = {"case": case, "H": H, "W": W, "B": B, "cost": kind, "g_ratio": gr, "train": train, "Tmax": Tmax, "mode": str(mode), "grad": bool(with_grad), "error": err}
            bad.append(d)
            if verbose:
                print(json.dumps(d), flush=True)
        if verbose and case % 20 == 19:
            print(json.dumps({"module_cases_so_far": n, "of": case + 1, "coupled": reruns, "with_grad": ngrad, "failures": len(bad)}), flush=True)
    run_module.ngrad = ngrad
    return n, reruns, bad


def run_encoder(seed=21, N=30, verbose=True):
    """cost maps of the MI355X inference encoders (f16x3: the default backend) against the SAME module on torch.nn fp32, random depths / map
    sizes / inputs / const, BatchNorm statistics and weights randomised; tolerance 1e-5 (north_star) on the cost map.  -> (cases, failures)"""
    from neural_astar.planner import NeuralAstar
    rng = np.random.default_rng(seed)
    bad, n = [], 0
    routes = {}
    for case in range(N):
        arch = "CNNDownSize" if rng.random() < 0.3 else "CNN"
        depth = int(rng.integers(1, 5))
        if arch == "CNN":
            H, W = (32, 32) if rng.random() < 0.25 else (int(rng.integers(4, 100)), int(rng.integers(4, 120)))
            gh, gw = H, W
        else:
            f = 1 << depth
            gh, gw = int(rng.integers(2, 9)), int(rng.integers(2, 9))
            H, W = gh * f, gw * f
        inp = str(rng.choice(["m+", "m"])) if arch == "CNN" else "m+"
        const = None if rng.random() < 0.5 else float(rng.choice([2.0, 10.0]))
        B = int(rng.integers(1, 6))
        torch.manual_seed(int(rng.integers(1 << 30)))
        na = NeuralAstar(encoder_input=inp, encoder_arch=arch, encoder_depth=depth, const=const, learn_obstacles=(arch == "CNNDownSize")).to(dev).eval()
        with torch.no_grad():
            for mod in na.encoder.modules():  # non-trivial BatchNorm statistics
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.normal_(0, 0.3)
                    mod.running_var.uniform_(0.5, 2.0)
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.normal_(0, 0.2)
            m = (torch.rand(B, 1, H, W, device=dev) > 0.2).float()
            s = torch.zeros(B, 1, gh, gw, device=dev)
            g = torch.zeros(B, 1, gh, gw, device=dev)
            s[:, 0, 0, 0] = 1
            g[:, 0, -1, -1] = 1
            na.encoder_backend = "auto"
            try:
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    c_hip = na.encode(m, s, g)
                route = na.last_encoder_route
                na.encoder_backend = "torch"
                c_ref = na.encode(m, s, g)
                scale = float(const or 1.0)
                err = float((c_hip - c_ref).abs().max()) / scale
                ok = err <= 1e-5 or not route.startswith("hip")
            except Exception as e:  # noqa: BLE001
                ok, err, route = False, -1.0, f"{type(e).__name__}: {e}"[:200]
        n += 1
        routes[route.split(" ")[0]] = routes.get(route.split(" ")[0], 0) + 1
        if not ok:
            d = {"case": case, "arch": arch, "depth": depth, "H": H, "W": W, "B": B, "input": inp, "const": const, "err": err, "route": route}
            bad.append(d)
            if verbose:
                print(json.dumps(d), flush=True)
    run_encoder.routes = routes
    return n, bad


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "module":
    # python tools/fuzz_parity.py module <seed> <N>: only the module sweep (every mode, autograd, large maps) against the literal batch loop
    # (a 4th argument "small": large maps only to 125x130 -- the oracle's dense restatement needs tens of seconds for one 150x200 batch)
    small = len(sys.argv) > 4 and sys.argv[4] == "small"
    nm, rr, bad_m = run_module(seed=int(sys.argv[2]) if len(sys.argv) > 2 else 31, N=int(sys.argv[3]) if len(sys.argv) > 3 else 300,
                               **({"large_frac": 0.06, "large_hw": ((112, 126), (112, 131))} if small else {}))
    print(json.dumps({"module_vs_literal_batch_loop_cases": nm, "batches_in_the_coupled_class": rr, "gradient_cases": run_module.ngrad, "module_failures": len(bad_m)}))
    sys.exit(1 if bad_m else 0)

if __name__ == "__main__":
    st, bad_cases = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20260926, int(sys.argv[2]) if len(sys.argv) > 2 else 160,
                        float(sys.argv[3]) if len(sys.argv) > 3 else 0.15)
    print(json.dumps({"cases": sum(st.values()), "by_kernel": st, "mismatches": len(bad_cases)}))
    nb, bad_b = run_backward(N=max(20, (int(sys.argv[2]) if len(sys.argv) > 2 else 160) // 10))
    print(json.dumps({"backward_cases": nb, "backward_failures": len(bad_b)}))
    bad_l = []
    if len(sys.argv) > 4:  # a few LARGE backward cases (seconds of oracle time each)
        nl, bad_l = run_backward(seed=5, N=int(sys.argv[4]), large=True)
        print(json.dumps({"backward_large_cases": nl, "backward_large_failures": len(bad_l)}))
    nm, rr, bad_m = run_module(N=int(sys.argv[6]) if len(sys.argv) > 6 else 60)
    print(json.dumps({"module_vs_literal_batch_loop_cases": nm, "batches_in_the_coupled_class": rr, "gradient_cases": run_module.ngrad, "module_failures": len(bad_m)}))
    if bad_m:
        sys.exit(1)
    ne, bad_e = run_encoder(N=int(sys.argv[5]) if len(sys.argv) > 5 else 40)
    print(json.dumps({"encoder_cases": ne, "encoder_failures": len(bad_e), "routes": run_encoder.routes}))
    sys.exit(1 if (bad_cases or bad_b or bad_l or bad_e) else 0)
