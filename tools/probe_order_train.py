"""Dev probe (GPU box): the training pair (forward with selection log at Tmax = 0.25 + replay backward) with placements.
The forward's own completion order (order_out) is the backward's placement: both are known inside ONE step, no previous visit needed."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import bench  # noqa: E402
from neural_astar import _native  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _native.load()
    B, H = 4096, 32
    pr = bench.make_problem("maze32", B, 1234)
    m, s, g = (torch.from_numpy(x[:, 0]).to(dev).contiguous() for x in pr)
    cost = torch.from_numpy(syn.random_costs(B, H, H, seed=3)[:, 0]).to(dev).contiguous()
    gh = torch.randn((B, H, H), device=dev)
    T = 256
    hist = torch.empty((B, H, H), device=dev); paths = torch.empty((B, H, H), dtype=torch.int64, device=dev)
    log = torch.empty((B, T), dtype=torch.int32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); status = torch.empty_like(iters)
    gc = torch.empty((B, H, H), device=dev)
    ws_n = int(lib.nastar_backward_workspace_bytes(B, H, H, T)); ws = torch.empty((ws_n,), dtype=torch.uint8, device=dev)
    bufs = [torch.zeros((B + 1,), dtype=torch.int32, device=dev) for _ in range(2)]
    st = torch.cuda.current_stream(dev).cuda_stream

    def fwd(order, out):
        rc = lib.nastar_forward_ordered(cost.data_ptr(), s.data_ptr(), g.data_ptr(), m.data_ptr(), B, H, H, 0.5, T, hist.data_ptr(), paths.data_ptr(),
                                        log.data_ptr(), iters.data_ptr(), status.data_ptr(), None, None, 0, 0,
                                        order.data_ptr() if order is not None else None, out.data_ptr() if out is not None else None, st)
        assert rc == 0, rc

    def bwd(order):
        rc = lib.nastar_backward_replay_ordered(gh.data_ptr(), None, None, None, cost.data_ptr(), s.data_ptr(), g.data_ptr(), m.data_ptr(), log.data_ptr(),
                                                B, H, H, 0.5, T, iters.data_ptr(), None, gc.data_ptr(), ws.data_ptr(), ws_n, 0,
                                                order.data_ptr() if order is not None else None, st)
        assert rc == 0, rc

    def timeit(f, n=100):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / n * 1e3, 1)

    fwd(None, bufs[0]); bwd(None); torch.cuda.synchronize()
    ref = gc.clone()
    it = iters.cpu().numpy()
    res = {"iters_mean": float(it.mean()), "at_budget": float((it >= T).mean())}
    res["fwd_natural_us"] = timeit(lambda: fwd(None, None))
    res["bwd_natural_us"] = timeit(lambda: bwd(None))
    fwd(None, bufs[0]); torch.cuda.synchronize()
    comp = bufs[0][:B].clone()
    res["bwd_by_forward_completion_order_us"] = timeit(lambda: bwd(comp))
    res["bwd_same_gradients"] = bool(torch.equal(gc, ref))
    desc = torch.from_numpy(np.argsort(-it, kind="stable").astype(np.int32)).to(dev)
    res["bwd_by_iters_desc_us"] = timeit(lambda: bwd(desc))
    res["fwd_hinted_by_previous_visit_us"] = timeit(lambda: fwd(comp, None))
    k = {"k": 0}

    def pair():
        k["k"] ^= 1
        fwd(None, bufs[k["k"]])
        bwd(bufs[k["k"]])
    res["pair_natural_us"] = timeit(lambda: (fwd(None, None), bwd(None)))
    res["pair_fwd_natural_bwd_by_its_completion_order_us"] = timeit(pair)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
