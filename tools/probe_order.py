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
        res = {"workload": w, "flags": flags, "iters_max": int(it.max()), "iters_mean": float(it.mean())}
        for name, o in orders.items():
            ot = torch.from_numpy(o.astype(np.int32)).to(dev) if o is not None else None
            us = [time_order(run, ot, dev) for _ in range(2)]
            ok = bool(torch.equal(run.hist, ref_hist) and torch.equal(run.paths, ref_paths))
            res[name] = {"us": [round(u, 1) for u in us], "same_outputs": ok}
        us = [time_order(run, None, dev, chain=True) for _ in range(2)]
        ok = bool(torch.equal(run.hist, ref_hist) and torch.equal(run.paths, ref_paths))
        res["chained_completion_order"] = {"us": [round(u, 1) for u in us], "same_outputs": ok}
        run.flags = flags
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
