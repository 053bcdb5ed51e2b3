#!/usr/bin/env python3
"""Where the host time of the drop-in boundary goes (run on the GPU box): VanillaAstar.forward() against the raw C-ABI launch, and
parallel.InFlightPlanner against round-robin C-ABI launches on preallocated buffers.  JSON lines on stdout."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bench_extras  # noqa: E402
from neural_astar import _native, ops  # noqa: E402
from neural_astar.parallel import InFlightPlanner  # noqa: E402
from neural_astar.planner import VanillaAstar  # noqa: E402

dev = torch.device("cuda:0")
lib = _native.load()


def levels_of(m, s, g):
    d = bench._device_distances(m[:, 0], g[:, 0]).reshape(m.shape[0], -1)
    return (d * (s.reshape(m.shape[0], -1) > 0)).sum(1).to(torch.int32).contiguous()


def raw_sync_floor(m, s, g, order, flags, reps=100):
    """the same launch through ctypes on preallocated outputs + a stream wait per call: what no Python shim can beat in sync mode"""
    B, _, H, W = m.shape
    hist = torch.empty((B, H, W), device=dev)
    paths = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def one():
        lib.nastar_forward_ex(m.data_ptr(), s.data_ptr(), g.data_ptr(), m.data_ptr(), B, H, W, 0.5, W * W, hist.data_ptr(), paths.data_ptr(), None,
                              it.data_ptr(), st.data_ptr(), None, None, 0, flags, order.data_ptr() if order is not None else None, None, None, None,
                              stream.cuda_stream)
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
        stream.synchronize()
    t_sync = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    t_async = (time.perf_counter() - t0) / reps
    return t_sync * 1e3, t_async * 1e3


def module_times(m, s, g, reps=100):
    out = {}
    for label, chk in (("sync", True), ("deferred", "deferred"), ("off", False)):
        va = VanillaAstar().to(dev).eval()
        va.astar.check_solvable = chk
        with torch.no_grad():
            for _ in range(10):
                va(m, s, g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                va(m, s, g)
            torch.cuda.synchronize()
            out[label + "_ms"] = (time.perf_counter() - t0) / reps * 1e3
            if chk != True:  # noqa: E712 -- host time to ISSUE a call (the GPU is behind: fewer than 64 calls)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(48):
                    va(m, s, g)
                out[label + "_issue_us"] = (time.perf_counter() - t0) / 48 * 1e6
                torch.cuda.synchronize()
            va.astar.raise_if_unsolvable()
    return out


def main():
    for w in ("maze32", "rand64"):
        prs = [bench.make_problem(w, 4096, seed=1234 + 1000 * k) for k in range(3)]
        batches = [tuple(torch.from_numpy(x).to(dev) for x in (pr.map_designs, pr.start_maps, pr.goal_maps)) for pr in prs]
        m, s, g = batches[0]
        order = ops.order_from_levels(levels_of(m, s, g))
        rec = {"workload": w}
        for name, o, fl in (("general_natural", None, 0), ("general_dataset_order", order, 0), ("unit_natural", None, 64), ("unit_dataset_order", order, 64)):
            a, b = raw_sync_floor(m, s, g, o, fl)
            rec["raw_" + name] = {"launch_plus_stream_wait_ms": a, "back_to_back_ms": b}
        rec["module_no_hint"] = module_times(m, s, g)
        ops.attach_order(s, levels_of(m, s, g))
        rec["module_with_dataset_order"] = module_times(m, s, g)
        del s.placement_order
        print(json.dumps(rec), flush=True)
        # batches in flight: round-robin raw launches on preallocated buffers vs the Python object
        n = 48
        va = VanillaAstar().to(dev).eval()
        for unit in (True, False):
            for k in (3, 4, 6):
                runs = [bench.Runner(prs[i % 3], dev, flags=64 if unit else 0, placement="natural") for i in range(k)]
                raw = bench_extras.multi_stream_throughput(prs[0], n, dev, k, runs=runs)
                del runs
                fly = InFlightPlanner(va, streams=k, unit_cost="auto" if unit else False)
                fly.plan_many(batches[i % 3] for i in range(n))  # warm: the allocator now holds n output sets
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    fly.submit(*batches[i % 3])
                t_sub = time.perf_counter() - t0
                outs = fly.collect()
                dt = time.perf_counter() - t0
                del outs
                t0 = time.perf_counter()
                for i in range(n):
                    fly.submit(*batches[i % 3], inputs_ready=True)
                outs = fly.collect()
                dt_ready = time.perf_counter() - t0
                del outs
                print(json.dumps({"workload": w, "unit": unit, "streams": k, "raw_round_robin_maps_per_s": raw, "in_flight_planner_maps_per_s": n * 4096 / dt,
                                  "submit_host_us_per_batch": t_sub / n * 1e6, "in_flight_inputs_ready_maps_per_s": n * 4096 / dt_ready}), flush=True)


if __name__ == "__main__":
    main()
