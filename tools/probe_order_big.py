"""Dev probe (GPU box): ONE launch over 32768 maps (8 rounds of resident maps) -- which placement?  natural / longest first /
longest and shortest alternating / long ones spread evenly.  Usage: python tools/probe_order_big.py [workloads] [flags]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import bench  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    workloads = (sys.argv[1] if len(sys.argv) > 1 else "maze32,rand32,rand64").split(",")
    for flags in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,64").split(",")]:
        for w in workloads:
            pr = bench.make_problem(w, 4096, 1234)
            big = syn.Problems(*(np.concatenate([x] * 8) for x in pr))
            run = bench.Runner(big, dev, flags=flags, placement="natural")
            B = run.B
            st = torch.cuda.current_stream(dev).cuda_stream

            def launch(order):
                rc = run.lib.nastar_forward_ordered(run.m.data_ptr(), run.s.data_ptr(), run.g.data_ptr(), run.m.data_ptr(), B, run.H, run.W,
                                                    run.g_ratio, run.max_iters, run.hist.data_ptr(), run.paths.data_ptr(), None,
                                                    run.iters.data_ptr(), run.status.data_ptr(), None, None, 0, flags,
                                                    order.data_ptr() if order is not None else None, None, st)
                assert rc == 0, rc

            def timeit(order, n=40):
                for _ in range(5):
                    launch(order)
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    launch(order)
                e1.record()
                torch.cuda.synchronize(dev)
                return e0.elapsed_time(e1) / n

            launch(None)
            torch.cuda.synchronize(dev)
            it = run.iters.cpu().numpy().astype(np.int64)
            ref = run.hist.clone()
            desc = np.argsort(-it, kind="stable")
            alt = np.empty(B, np.int64)
            alt[0::2] = desc[:B // 2]
            alt[1::2] = desc[::-1][:B // 2]
            # the longest eighth first (one per resident slot of the first round), the rest in natural order
            k = B // 8
            head = desc[:k]
            rest = np.setdiff1d(np.arange(B), head, assume_unique=False)
            spread = np.concatenate([head, rest])
            res = {"workload": w, "flags": flags, "maps": B}
            nbytes = 28 * run.H * run.W * B
            for name, o in (("natural", None), ("longest_first", desc), ("alternating", alt), ("longest_eighth_first_then_natural", spread)):
                ot = torch.from_numpy(o.astype(np.int32)).to(dev) if o is not None else None
                ms = timeit(ot)
                res[name] = {"ms": round(ms, 4), "M_maps_per_s": round(B / ms / 1e3, 1), "hbm_frac": round(nbytes / (ms * 1e-3) / 8e12, 3),
                             "same": bool(torch.equal(run.hist, ref))}
            print(json.dumps(res), flush=True)
            del run


if __name__ == "__main__":
    main()
