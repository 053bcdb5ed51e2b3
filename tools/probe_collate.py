"""Where does the collated step (search launch + 1 RCCL all-gather, bench.py's N > 1 path) lose time against the plain search step?

1-rank RCCL group on one GPU (the collective's launch path, stream hand-over and a device-to-device copy; no xGMI traffic).
Prints per-step wall time for: plain search, search emitting the packed masks, + all-gather (bench.py's scheme), the same with the
gathered buffer preallocated, and the host time spent SUBMITTING each variant (loop time before the final synchronize)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "neural-astar_amd"))
import bench  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    wl = sys.argv[1] if len(sys.argv) > 1 else "maze32"
    prs = [bench.make_problem(wl, 4096, seed=1234 + 1000 * k) for k in range(3)]
    run = bench.Runner(prs, dev)
    bench.prewarm(run, dev, 0.5)

    def measure(name, body, steps=200, fin=None):
        for _ in range(20):
            body()
        if fin:
            fin()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            body()
        t_submit = time.perf_counter() - t0
        if fin:
            fin()
        torch.cuda.synchronize(dev)
        t = time.perf_counter() - t0
        print(f"{name:58s} {t / steps * 1e6:8.1f} us/step   host submit {t_submit / steps * 1e6:8.1f} us/step", flush=True)

    measure("plain nastar_forward", run.step)
    run.enable_packed()
    measure("nastar_forward_packed", run.step)

    from neural_astar import parallel
    state = {"pending": None}

    def step_gather():
        run.step()
        _, fin = parallel.all_gather_packed(run.packed[run._pk], async_op=True)
        if state["pending"] is not None:
            state["pending"]()
        state["pending"] = fin

    def fin_all():
        if state["pending"] is not None:
            state["pending"]()
            state["pending"] = None

    measure("packed + all_gather (bench.py scheme)", step_gather, fin=fin_all)

    outs = [torch.empty_like(run.packed[0]) for _ in range(2)]
    works = [None, None]

    def step_gather_prealloc():
        run.step()
        k = run._pk
        if works[k] is not None:
            works[k].wait()
        works[k] = dist.all_gather_into_tensor(outs[k], run.packed[k], async_op=True)

    def fin_pre():
        for k in (0, 1):
            if works[k] is not None:
                works[k].wait()
                works[k] = None

    measure("packed + all_gather, preallocated, wait 2 steps later", step_gather_prealloc, fin=fin_pre)

    def step_gather_sync():
        run.step()
        dist.all_gather_into_tensor(outs[0], run.packed[run._pk])

    measure("packed + all_gather on the search stream order (no overlap)", step_gather_sync)

    # the collective alone
    def only_gather():
        w = dist.all_gather_into_tensor(outs[0], run.packed[0], async_op=True)
        w.wait()

    measure("all_gather alone (1 MB, 1 rank)", only_gather)

    # a plain device copy on a side stream instead of the collective: what the overlap itself costs
    side = torch.cuda.Stream(dev)
    evs = [torch.cuda.Event() for _ in range(2)]

    def step_copy_side():
        run.step()
        k = run._pk
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            outs[k].copy_(run.packed[k], non_blocking=True)
            evs[k].record(side)
        main.wait_event(evs[k ^ 1])

    measure("packed + side-stream copy (no RCCL)", step_copy_side)

    # the search on a HIGH-priority stream: its workgroups are dispatched ahead of the collective's when both are ready
    hp = torch.cuda.Stream(dev, priority=-1)
    with torch.cuda.stream(hp):
        measure("HIGH-priority search stream: plain packed", run.step)
        measure("HIGH-priority search stream: packed + all_gather overlapped", step_gather, fin=fin_all)
        measure("HIGH-priority search stream: packed + all_gather prealloc", step_gather_prealloc, fin=fin_pre)
    lowside = torch.cuda.Stream(dev, priority=0)

    def step_copy_side_hp():
        run.step()
        k = run._pk
        main = torch.cuda.current_stream(dev)
        lowside.wait_stream(main)
        with torch.cuda.stream(lowside):
            outs[k].copy_(run.packed[k], non_blocking=True)
            evs[k].record(lowside)
        main.wait_event(evs[k ^ 1])

    with torch.cuda.stream(hp):
        measure("HIGH-priority search stream: packed + side-stream copy", step_copy_side_hp)

    # buffer reuse guarded on the HOST (is_completed() poll; stream wait only if the collective is really still running):
    # no barrier packet in the search queue
    def step_gather_hostguard():
        run.step()
        k = run._pk
        works[k] = dist.all_gather_into_tensor(outs[k], run.packed[k], async_op=True)
        w = works[k ^ 1]
        if w is not None and not w.is_completed():
            w.wait()

    with torch.cuda.stream(hp):
        measure("HIGH-priority search stream: all_gather, host-side reuse guard", step_gather_hostguard, fin=fin_pre)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
