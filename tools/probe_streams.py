#!/usr/bin/env python3
"""Throughput regime of the forward search (dev probe): the same 4096-map batches issued round-robin over S HIP streams, per
workload and per NASTAR_FORWARD_FLAGS value, plus ONE launch of a k-times larger batch on one stream (the hardware's own
workgroup refill instead of overlapping launches).

    python tools/probe_streams.py --workloads maze32,rand32,rand64 --flags 0,8 --streams 1,2,4,6 --bigb 4 [--steps 240]

Prints one JSON object per (workload, flags).  With NASTAR_LIB=.../libnastar_hip_dev.so also flags 4 (two maps per wavefront).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "neural-astar_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import bench_extras  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402


def big_batch_throughput(prs, steps, dev):
    """ONE launch per step over the concatenation of `prs` (k x 4096 maps), one stream"""
    cat = syn.Problems(*(np.concatenate([p[i] for p in prs]) for i in range(3)))
    run = bench.Runner(cat, dev)
    for _ in range(3):
        run.step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return run.B * steps / dt, dt / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="maze32,rand32,rand64")
    ap.add_argument("--flags", default="0")
    ap.add_argument("--streams", default="1,2,3,4,6")
    ap.add_argument("--bigb", default="4", help="comma list of batch multipliers for the one-launch variant (0 = skip)")
    ap.add_argument("--steps", type=int, default=240)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for w in a.workloads.split(","):
        prs = [bench.make_problem(w, bench.B_PER_GPU, seed=1234)] * 8  # k copies of ONE batch: same work per copy, distinct memory
        for f in a.flags.split(","):
            os.environ["NASTAR_FORWARD_FLAGS"] = f
            out = {"workload": w, "flags": int(f), "lib": os.environ.get("NASTAR_LIB", "product"), "streams": {}, "one_launch": {}}
            try:
                r0 = bench.Runner(prs[0], dev)
                bench.prewarm(r0, dev, 0.2)
                for s in (int(x) for x in a.streams.split(",")):
                    out["streams"][str(s)] = round(bench_extras.multi_stream_throughput(prs[0], a.steps, dev, s) / 1e6, 2)
                for k in (int(x) for x in a.bigb.split(",")):
                    if k > 0:
                        v, ms = big_batch_throughput(prs[:k], max(10, a.steps // k), dev)
                        out["one_launch"][str(k * bench.B_PER_GPU)] = {"M_maps_per_s": round(v / 1e6, 2), "ms": round(ms, 4)}
            except Exception as e:  # noqa: BLE001
                out["error"] = f"{type(e).__name__}: {e}"
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
