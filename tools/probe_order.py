"""Dev probe (GPU box): what does a placement that knows the step counts buy?  (`order[B]`)
nastar_forward_ordered runs map order[i] in workgroup i.  The step counts of a first pass give the orders tried here.
Usage: python tools/probe_order.py [workloads] [flags]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import bench  # noqa: E402


def time_order(run, order, dev, steps=300, chain=False):
    """chain: every launch emits the placement of the next one (order_out -> order), starting from `order`"""
    lib = run.lib
    st = torch.cuda.current_stream(dev).cuda_stream
    bufs = [torch.zeros((run.B + 1,), dtype=torch.int32, device=dev) for _ in range(2)]
    state = {"cur": order, "k": 0}

    def one():
        cur = state["cur"]
        out = bufs[state["k"]] if chain else None
        rc = lib.nastar_forward_ordered(run.m.data_ptr(), run.s.data_ptr(), run.g.data_ptr(), run.m.data_ptr(), run.B, run.H, run.W,
                                        run.g_ratio, run.max_iters, run.hist.data_ptr(), run.paths.data_ptr(), None,
                                        run.iters.data_ptr(), run.status.data_ptr(), None, None, 0, run.flags,
                                        cur.data_ptr() if cur is not None else None, out.data_ptr() if out is not None else None, st)
        assert rc == 0, rc
        if chain:
            state["cur"] = out
            state["k"] ^= 1
    for _ in range(30):
        one()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps * 1e3


def bfs_features(pr):
    m = pr.map_designs[:, 0] > 0
    s = pr.start_maps[:, 0] > 0
    g = pr.goal_maps[:, 0] > 0
    B = m.shape[0]
    vis = s.copy()
    L = np.zeros(B, np.int64)
    V = np.zeros(B, np.int64)
    done = np.zeros(B, bool)
    for lvl in range(1, m.shape[1] * m.shape[2]):
        p = np.pad(vis, ((0, 0), (1, 1), (1, 1)))
        nb = np.zeros_like(vis)
        for dr in (0, 1, 2):
            for dc in (0, 1, 2):
                nb |= p[:, dr:dr + vis.shape[1], dc:dc + vis.shape[2]]
        new = (nb & m) | vis
        reached = (new & g).any((1, 2)) & ~done
        L[reached] = lvl
        V[reached] = new[reached].sum((1, 2))
        done |= reached
        if done.all() or (new == vis).all():
            break
        vis = new
    return L, V


def main():
    dev = torch.device("cuda:0")
    workloads = (sys.argv[1] if len(sys.argv) > 1 else "maze32,rand32,rand64").split(",")
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for w in workloads:
        pr = bench.make_problem(w, 4096, 1234)
        run = bench.Runner(pr, dev, flags=flags)
        bench.prewarm(run, dev, 0.5)
        run.step()
        torch.cuda.synchronize(dev)
        it = run.iters.cpu().numpy().astype(np.int64)
        ref_hist, ref_paths = run.hist.clone(), run.paths.clone()
        B = run.B
        desc = np.argsort(-it, kind="stable")
        orders = {"identity": None, "identity_explicit": np.arange(B), "longest_first": desc, "shortest_first": desc[::-1].copy()}
        # longest maps spread with stride: position i gets the (i % 256)-th group's next map
        for grp in (256, 512, 1024):
            o = np.empty(B, np.int64)
            o[:] = desc.reshape(B // grp, grp).T.reshape(-1) if False else desc.reshape(-1, grp).reshape(-1)
            orders[f"desc_blocks{grp}"] = o
        # snake: ranks 0..255 forward, 256..511 backward, ... (balances the per-CU sums if the dispatcher deals round-robin)
        sn = desc.reshape(-1, 256).copy()
        sn[1::2] = sn[1::2, ::-1]
        orders["snake256"] = sn.reshape(-1)
        # interleave: longest, shortest, 2nd longest, 2nd shortest ...
        il = np.empty(B, np.int64)
        il[0::2] = desc[:B // 2]
        il[1::2] = desc[::-1][:B // 2]
        orders["long_short_alternating"] = il
        rng = np.random.default_rng(0)
        orders["random"] = rng.permutation(B)
        # what a cheap PRE-PASS could know about a map never searched before: unit-cost BFS from the start (Moore-8) until the goal is
        # reached -- L = its level (the optimal path length), V = cells reached by then (an upper bound of the expansions of a unit-cost A*)
        L, V = bfs_features(pr)
        corr = {"corr_iters_bfs_level": float(np.corrcoef(it, L)[0, 1]), "corr_iters_bfs_visited": float(np.corrcoef(it, V)[0, 1])}
        si = pr.start_maps[:, 0].reshape(B, -1).argmax(1)
        gi = pr.goal_maps[:, 0].reshape(B, -1).argmax(1)
        Wd = pr.map_designs.shape[-1]
        cheb = np.maximum(np.abs(si // Wd - gi // Wd), np.abs(si % Wd - gi % Wd))
        corr["corr_iters_chebyshev"] = float(np.corrcoef(it, cheb)[0, 1])
        orders["chebyshev_desc"] = np.argsort(-cheb, kind="stable")
        orders["bfs_level_desc_8buckets"] = np.argsort(-(L * 8 // (L.max() + 1)), kind="stable")
        orders["bfs_level_desc"] = np.argsort(-L, kind="stable")
        orders["bfs_visited_desc"] = np.argsort(-V, kind="stable")
        orders["bfs_visited_desc_64buckets"] = np.argsort(-(V * 64 // (V.max() + 1)), kind="stable")
        res = {"workload": w, "flags": flags, "iters_max": int(it.max()), "iters_mean": float(it.mean())}
        for name, o in orders.items():
            ot = torch.from_numpy(o.astype(np.int32)).to(dev) if o is not None else None
            us = [time_order(run, ot, dev) for _ in range(2)]
            ok = bool(torch.equal(run.hist, ref_hist) and torch.equal(run.paths, ref_paths))
            res[name] = {"us": [round(u, 1) for u in us], "same_outputs": ok}
        res.update(corr)
        # a FRESH batch end to end: predictor (2 small launches) + search with its order, all on one stream
        from neural_astar import ops
        mt, stt, gt = run.m, run.s, run.g

        def fresh():
            o = ops.placement_predict(mt, stt, gt)
            rc = run.lib.nastar_forward_ordered(run.m.data_ptr(), run.s.data_ptr(), run.g.data_ptr(), run.m.data_ptr(), run.B, run.H, run.W,
                                                run.g_ratio, run.max_iters, run.hist.data_ptr(), run.paths.data_ptr(), None,
                                                run.iters.data_ptr(), run.status.data_ptr(), None, None, 0, run.flags, o.data_ptr(), None,
                                                torch.cuda.current_stream(dev).cuda_stream)
            assert rc == 0
        for _ in range(20):
            fresh()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fresh()
        e1.record()
        torch.cuda.synchronize(dev)
        res["fresh_batch_predict_plus_search_us"] = round(e0.elapsed_time(e1) / 200 * 1e3, 1)
        e0.record()
        for _ in range(200):
            ops.placement_predict(mt, stt, gt)
        e1.record()
        torch.cuda.synchronize(dev)
        res["predictor_alone_us"] = round(e0.elapsed_time(e1) / 200 * 1e3, 1)
        o, lv = ops.placement_predict(mt, stt, gt, return_levels=True)
        res["predictor_levels_equal_numpy_bfs"] = bool(np.array_equal(lv.cpu().numpy(), L))
        res["same_outputs_fresh"] = bool(torch.equal(run.hist, ref_hist) and torch.equal(run.paths, ref_paths))
        us = [time_order(run, None, dev, chain=True) for _ in range(2)]
        ok = bool(torch.equal(run.hist, ref_hist) and torch.equal(run.paths, ref_paths))
        res["chained_completion_order"] = {"us": [round(u, 1) for u in us], "same_outputs": ok}
        run.flags = flags
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
