#!/usr/bin/env python3
"""bench.py -- forward DifferentiableAstar throughput on MI355X (the BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is ONE pass of the hot path (one ``nastar_forward`` launch through the C ABI) over one batch of
B = 4096 synthetic 32x32 Moore-8 maps that already sit in HBM.  With N > 1 (launched by
``python -m torch.distributed.run --nproc-per-node N``) every rank owns its own 4096 maps (weak scaling) and each
step also collates the bit-packed ``AstarOutput`` of all ranks with ONE RCCL all-gather, overlapped with the next
step's search.  Rank 0 prints ONE JSON line.

Workload (``config.workload``): ``maze32`` = seeded maze-like stand-in for the absent ``mazes_032_moore_c8.npz``
(SURVEY.md section 8d-ii), VanillaAstar convention cost = map, g_ratio 0.5, eval mode (search runs to the goal).
The easier ``rand32`` (random 25 % obstacles; what BASELINE.md's CPU probes used) is reported as a secondary figure.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "neural-astar_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H = W = 32  # headline workload; `Runner` carries the size of whatever workload it was given
B_PER_GPU = 4096
G_RATIO = 0.5
BYTES_PER_MAP = 28 * H * W  # reads cost+start+goal+passable (4x4 B/cell), writes fp32 hist + int64 paths
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MAX_FRESH_SETS = 512  # never-searched batches the headline loop may hold resident (50 MB of inputs each at 4096 x 32x32)
N_ROTATE = 3  # distinct input/output batch sets cycled by the timed loop: 3 x 100 MB > the 256 MB MALL, so HBM is what is read


_T0 = time.perf_counter()


def _log(msg: str) -> None:
    """progress on stderr (the JSON line on stdout stays alone)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def _synth(kind: str, B: int, seed: int):
    from neural_astar.utils import synthetic as syn
    if kind == "maze32":
        return syn.maze_maps(B, 32, seed=seed)
    if kind == "rand32":
        return syn.random_obstacle_maps(B, 32, 32, 0.25, seed=seed)
    if kind == "rand64":  # 64x64, p = 0.20 (the per-GPU shard of BASELINE config 4)
        return syn.random_obstacle_maps(B, 64, 64, 0.20, seed=seed)
    raise ValueError(kind)


def make_problem(kind: str, B: int, seed: int, wait: bool = True):
    """One seeded synthetic batch, cached under /tmp.  Several ranks of one node may ask for the same batch at once (strong
    scaling: every rank touches chunks of ONE global batch): an exclusive file lock makes exactly one of them synthesise it, the
    others load the finished file.  wait=False: return None instead of waiting when another process is building it."""
    import fcntl
    from neural_astar.utils import synthetic as syn
    cache = os.path.join("/tmp", f"nastar_bench_{kind}_{B}_{seed}.npz")

    def load():
        z = np.load(cache)
        return syn.Problems(z["m"], z["s"], z["g"])
    if os.path.exists(cache):
        return load()
    try:
        lock = open(cache + ".lock", "w")
    except OSError:  # /tmp not writable: no cache, every caller synthesises
        return _synth(kind, B, seed)
    try:
        try:
            fcntl.flock(lock, fcntl.LOCK_EX | (0 if wait else fcntl.LOCK_NB))
        except BlockingIOError:
            return None
        if os.path.exists(cache):  # built while this process waited for the lock
            return load()
        pr = _synth(kind, B, seed)
        try:
            tmp = f"{cache}.{os.getpid()}.tmp.npz"
            np.savez(tmp, m=pr.map_designs, s=pr.start_maps, g=pr.goal_maps)
            os.replace(tmp, cache)  # atomic: a reader never sees a half-written file
        except OSError:
            pass
        return pr
    finally:
        lock.close()


def make_problem_rows(kind: str, total: int, seed: int, rows: np.ndarray, rank: int = 0, world: int = 1):
    """Rows `rows` of a `total`-map global batch that is defined chunk-wise (1024 maps per chunk, chunk c seeded from (seed, c)),
    so that a rank only synthesises the chunks its shard touches and every N sees the same global batch.  With interleaved shards
    every rank touches every chunk: a first pass visits the chunks starting at a rank-dependent offset and skips the ones another
    rank is already building (make_problem's file lock), so the ranks of a node synthesise DIFFERENT chunks in parallel; the second
    pass loads (or waits for) all of them."""
    from neural_astar.utils import synthetic as syn
    CH = 1024
    rows = np.asarray(rows)
    assert (np.diff(rows) > 0).all(), "rows must be ascending"
    chunks = [int(c) for c in np.unique(rows // CH)]

    def chunk_args(c):
        return kind, min(CH, total - c * CH), seed * 100003 + c
    got = {}
    k0 = (rank * len(chunks)) // max(world, 1)
    for c in chunks[k0:] + chunks[:k0]:
        pr = make_problem(*chunk_args(c), wait=False)
        if pr is not None:
            got[c] = pr
    parts = []
    for c in chunks:
        pr = got.get(c) or make_problem(*chunk_args(c))
        sel = rows[(rows // CH) == c] - c * CH
        parts.append(tuple(x[sel] for x in pr))
    return syn.Problems(*(np.concatenate([p[k] for p in parts]) for k in range(3)))


class Runner:
    """Device-resident inputs + preallocated outputs; step() = one nastar_forward_ex launch on torch's current stream.

    `prs` may be a list of problem sets (same shape): step i works on set i % len(prs), each with its own input buffers and one of
    N_ROTATE rotating output buffer sets, so consecutive timed steps do not re-read a cache-resident batch (>256 MB in rotation).
    A set is a host `Problems` tuple or a device dict {"m", "s", "g"[, "order"]} (FreshBatches.batch)."""

    def __init__(self, prs, dev, g_ratio=G_RATIO, max_iters=None, flags=None, placement=None):
        from neural_astar import _native
        self.lib = _native.load()
        self._check = _native.check
        self.dev = dev
        if not isinstance(prs, (list, tuple)) or hasattr(prs, "map_designs"):
            prs = [prs]
        self.sets = []
        outs = []
        for pr in prs:
            if isinstance(pr, dict):
                z = dict(m=pr["m"], s=pr["s"], g=pr["g"], order=pr.get("order"))
            else:
                z = dict(m=torch.from_numpy(pr.map_designs[:, 0]).to(dev).contiguous(), s=torch.from_numpy(pr.start_maps[:, 0]).to(dev).contiguous(),
                         g=torch.from_numpy(pr.goal_maps[:, 0]).to(dev).contiguous(), order=None)
            B, Hh, Ww = z["m"].shape
            if len(outs) < max(N_ROTATE, 1):
                outs.append(dict(hist=torch.empty((B, Hh, Ww), dtype=torch.float32, device=dev),
                                 paths=torch.empty((B, Hh, Ww), dtype=torch.int64, device=dev),
                                 iters=torch.empty((B,), dtype=torch.int32, device=dev),
                                 status=torch.empty((B,), dtype=torch.int32, device=dev)))
            z.update(outs[len(self.sets) % len(outs)])
            self.sets.append(z)
        self.B, self.H, self.W = self.sets[0]["m"].shape
        self.g_ratio = float(g_ratio)
        self.max_iters = int(max_iters) if max_iters is not None else self.W * self.W  # eval mode: search to the goal
        # NASTAR_FLAG_* of include/nastar.h; default: the dev A/B switch NASTAR_FORWARD_FLAGS (0 = the general kernel)
        self.flags = int(os.environ.get("NASTAR_FORWARD_FLAGS", "0")) if flags is None else int(flags)
        self.packed = None  # set by enable_packed(): the step then also emits the bit-packed masks (all-gather payload)
        # placement (which map workgroup i searches; outputs never depend on it):
        #   "dataset": the order the batch was ASSEMBLED with -- counting sort of |opt_dist[start]|, data every sample of the reference's maze
        #              files carries (utils/data.py:127-134,200-221); nothing measured in an earlier search is used (the headline);
        #   "hinted":  every batch set remembers the order its searches finished in at its previous visit and the next visit starts the
        #              longest first: what a validation loop over a fixed set has every epoch after the first;
        #   "natural": workgroup i searches map i.
        self.placement = (PLACEMENT if placement is None else placement)
        if self.placement == "dataset" and any(z["order"] is None for z in self.sets):
            raise ValueError("placement 'dataset' needs sets assembled with an order (FreshBatches.batch)")
        if self.placement == "hinted":
            for z in self.sets:
                z["ord"] = [torch.zeros((self.B + 1,), dtype=torch.int32, device=dev) for _ in range(2)]
                z["k"], z["seen"] = 0, False
        self._i = 0
        self._bind(0)

    def _bind(self, k):
        z = self.sets[k]
        self.m, self.s, self.g = z["m"], z["s"], z["g"]
        self.hist, self.paths, self.iters, self.status = z["hist"], z["paths"], z["iters"], z["status"]

    def enable_packed(self):
        nb = (self.H * self.W + 7) // 8
        self.packed = [torch.empty((self.B, 2 * nb), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self._pk = 0

    def step(self):
        z = self.sets[self._i % len(self.sets)]
        self._bind(self._i % len(self.sets))
        self._i += 1
        order = order_out = None
        if self.placement == "hinted":
            order = z["ord"][z["k"]] if z["seen"] else None
            order_out = z["ord"][z["k"] ^ 1]
        elif self.placement == "dataset":
            order = z["order"]
        pk = None
        if self.packed is not None:
            self._pk ^= 1  # double buffer: the previous step's payload may still be in flight in the all-gather
            pk = self.packed[self._pk].data_ptr()
        rc = self.lib.nastar_forward_ex(
            self.m.data_ptr(), self.s.data_ptr(), self.g.data_ptr(), self.m.data_ptr(), self.B, self.H, self.W,
            self.g_ratio, self.max_iters, self.hist.data_ptr(), self.paths.data_ptr(), None, self.iters.data_ptr(),
            self.status.data_ptr(), pk, None, 0, self.flags, order.data_ptr() if order is not None else None,
            order_out.data_ptr() if order_out is not None else None, None, None, torch.cuda.current_stream(self.dev).cuda_stream)
        self._check(rc, "nastar_forward_ex")
        if self.placement == "hinted":  # only now: a launch that failed must not leave a half-initialised order as the next hint
            z["k"] ^= 1
            z["seen"] = True


def _device_distances(m, g):
    """[P,H,W] fp32 passable maps + goal one-hots on the device -> [P,H,W] int32 8-connected unit-cost distance to the goal (-1: obstacle
    or unreachable): what a planning-datasets file stores as `opt_dists` (reference utils/data.py:127-134).  Bench-side data
    preparation (one dilation per level with torch ops), not the product path."""
    import torch.nn.functional as F
    passable = m > 0
    dist = torch.full(m.shape, -1, dtype=torch.int32, device=m.device)
    front = (g > 0) & passable
    seen = front.clone()
    d = 0
    while True:
        dist[front] = d
        grown = F.max_pool2d(front.float().unsqueeze(1), 3, 1, 1)[:, 0] > 0
        front = grown & passable & ~seen
        if d % 8 == 7 and not bool(front.any()):
            break
        seen |= front
        d += 1
        if d > m.shape[-1] * m.shape[-2]:
            break
    return dist


class FreshBatches:
    """Batches that have NEVER been searched, assembled the way the reference's loader assembles them: a resident pool of problems in
    the layout of a planning-datasets split (map_designs, goal_maps, opt_dists; reference utils/data.py:127-134) and, per batch, a random
    draw of maps with a random START per map -- for the maze workload from a random one of the 55-70 / 70-85 / 85-100 percentile bands
    of the optimal distance (utils/data.py:200-221, `get_random_start_map`), for the random-obstacle workloads uniform over the cells that
    reach the goal.  The sample's own `opt_dists[start]` gives the batch its placement (`order`: counting sort, longest route first;
    ops.order_from_levels) AT ASSEMBLY -- no search of these inputs, and nothing any search has measured, goes into it."""

    def __init__(self, kind, dev, prs, seed):
        self.kind, self.dev = kind, dev
        self.m = torch.cat([torch.from_numpy(pr.map_designs[:, 0]) for pr in prs]).to(dev).contiguous()
        self.g = torch.cat([torch.from_numpy(pr.goal_maps[:, 0]) for pr in prs]).to(dev).contiguous()
        self.s0 = torch.cat([torch.from_numpy(pr.start_maps[:, 0]) for pr in prs]).to(dev).contiguous()  # the generator's own starts (fixed())
        self.P, self.H, self.W = self.m.shape
        self.dist = _device_distances(self.m, self.g).reshape(self.P, -1)
        d = self.dist.cpu().numpy()
        self.th = None
        if kind.startswith("maze"):
            th = np.empty((self.P, 4), np.float64)
            for n in range(self.P):
                v = d[n]
                th[n] = np.percentile(v[v > 0], 100.0 * np.array([0.55, 0.70, 0.85, 1.0]))
            self.th = torch.from_numpy(th).to(dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)

    def fixed(self, k, B):
        """rows [k B, (k + 1) B) of the pool with the starts the generator drew (a recurring, seeded batch: the oracle / reference
        checks and the strong-scaling global batch), placed by ITS starts' distances"""
        from neural_astar import ops
        sl = slice(k * B, (k + 1) * B)
        s = self.s0[sl]
        levels = (self.dist[sl] * (s.reshape(B, -1) > 0)).sum(1).to(torch.int32).contiguous()
        return {"m": self.m[sl], "s": s, "g": self.g[sl], "order": ops.order_from_levels(levels), "levels": levels}

    def batch(self, B):
        from neural_astar import ops
        idx = torch.randperm(self.P, device=self.dev, generator=self.gen)[:B]
        d = self.dist[idx]
        mask = d > 0
        if self.th is not None:
            band = torch.randint(0, 3, (B, 1), device=self.dev, generator=self.gen)
            th = self.th[idx]
            lo, hi = torch.gather(th, 1, band), torch.gather(th, 1, band + 1)
            dd = d.double()
            banded = (dd >= lo) & (dd <= hi) & mask
            mask = torch.where(banded.any(1, keepdim=True), banded, mask)
        sidx = torch.multinomial(mask.float(), 1, generator=self.gen)
        s = torch.zeros((B, self.H * self.W), dtype=torch.float32, device=self.dev)
        s.scatter_(1, sidx, 1.0)
        levels = torch.gather(d, 1, sidx).reshape(-1).contiguous()
        return {"m": self.m[idx].contiguous(), "s": s.reshape(B, self.H, self.W), "g": self.g[idx].contiguous(),
                "order": ops.order_from_levels(levels), "levels": levels}


PLACEMENT = "hinted"  # default of Runner(placement=None) for the secondary figures; the headline runner is built with placement="dataset"
PREWARM_S = 0.3  # untimed launches before the W warm-up steps: the driver times 20 steps (~3 ms) after 5 warm-up steps, which on a GPU fresh out
                 # of problem synthesis measures the clock ramp, not the kernel (same process: 23.6 M maps/s first, 25.9 M a minute later)


def prewarm(run, dev, seconds=PREWARM_S):
    """Bring the GPU to its steady clock state: untimed launches of the workload for `seconds` (not part of the W warm-up or K timed steps)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            run.step()
        torch.cuda.synchronize(dev)


def fresh_batches_pipelined(run, steps, warmup, dev):
    """Batches that have NEVER been searched, in a pipeline: while batch i is searched, a side stream computes the placement of batch
    i + 1 from its maps alone (nastar_placement_predict: length of the shortest route by a bit-parallel wave + counting sort; 17-44 us of
    small launches that fit into the search launch's idle tail).  Nothing measured on an earlier visit of a batch is used.  Returns
    seconds for `steps` steps, or None when the map size has no predictor."""
    lib = run.lib
    if run.H != run.W or run.W not in (32, 64):
        return None
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    nset = len(run.sets)
    orders = [torch.empty((run.B,), dtype=torch.int32, device=dev) for _ in range(nset)]
    wss = [torch.empty((run.B,), dtype=torch.int32, device=dev) for _ in range(nset)]
    ready = [torch.cuda.Event() for _ in range(nset)]
    done = [torch.cuda.Event() for _ in range(nset)]

    def predict(k):
        z = run.sets[k]
        side.wait_event(done[k])  # the previous search of this set no longer reads its order buffer
        rc = lib.nastar_placement_predict(z["m"].data_ptr(), z["s"].data_ptr(), z["g"].data_ptr(), run.B, run.H, run.W, orders[k].data_ptr(),
                                          wss[k].data_ptr(), run.B * 4, side.cuda_stream)
        run._check(rc, "nastar_placement_predict")
        ready[k].record(side)

    def search(k):
        z = run.sets[k]
        main.wait_event(ready[k])
        rc = lib.nastar_forward_ordered(z["m"].data_ptr(), z["s"].data_ptr(), z["g"].data_ptr(), z["m"].data_ptr(), run.B, run.H, run.W,
                                        run.g_ratio, run.max_iters, z["hist"].data_ptr(), z["paths"].data_ptr(), None, z["iters"].data_ptr(),
                                        z["status"].data_ptr(), None, None, 0, run.flags, orders[k].data_ptr(), None, main.cuda_stream)
        run._check(rc, "nastar_forward_ordered")
        done[k].record(main)

    for k in range(nset):
        done[k].record(main)
    predict(0)
    i = 0
    for phase, n in (("warm", warmup), ("timed", steps)):
        if phase == "timed":
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        for _ in range(n):
            k = i % nset
            predict((i + 1) % nset)  # ... of the NEXT batch, beside this batch's search
            search(k)
            i += 1
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def timed_loop(run, steps, warmup, world, dev, collate=None):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (seconds max over ranks, device ms)."""
    import gc
    gc_was = gc.isenabled()
    gc.collect()   # (before the warm-up steps: a collection between warm-up and clock start would idle the GPU for tens of ms)
    gc.disable()
    pending = None
    for _ in range(warmup):
        run.step()
        if collate is not None:
            pending = collate(pending)
    if pending is not None:
        pending()
    # host hygiene of the timed region: the two events exist and have been recorded once BEFORE the clock starts (their first record after
    # a few thousand untimed launches was seen to take 40 ms in a process that had synthesised its problems itself -- 2.1 ms per step
    # instead of 0.12 over the driver's 20 steps, the GPU idle meanwhile), and the cyclic garbage collector does not run inside it
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    e1.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    pending = None
    t0 = time.perf_counter()
    e0.record()
    dbg = [] if os.environ.get("NASTAR_BENCH_DEBUG") else None
    for _ in range(steps):
        ts = time.perf_counter()
        run.step()
        if dbg is not None:
            dbg.append(time.perf_counter() - ts)
        if collate is not None:
            pending = collate(pending)
    e1.record()
    t_sub = time.perf_counter() - t0
    if dbg:
        _log("per-step submit us: " + " ".join(f"{x * 1e6:.0f}" for x in dbg))
    if pending is not None:
        pending()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    if os.environ.get("NASTAR_BENCH_DEBUG"):
        _log(f"timed_loop: submit {t_sub * 1e3:.3f} ms, total {dt * 1e3:.3f} ms, placement {getattr(run, 'placement', None)}")
    dev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, dev_ms


def multi_stream_throughput(pr, steps, dev, nstreams, flags=None, runs=None):
    runs = runs if runs is not None else [Runner(pr, dev, flags=flags, placement="natural") for _ in range(nstreams)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for i in range(2 * nstreams):
        with torch.cuda.stream(streams[i % nstreams]):
            runs[i % nstreams].step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]):
            runs[i % nstreams].step()
    torch.cuda.synchronize(dev)
    return runs[0].B * steps / (time.perf_counter() - t0)


FLAG_UNIT_COST = 64  # include/nastar.h NASTAR_FLAG_UNIT_COST
LONE_STEP_NS = 219.0  # step of a wavefront that has its SIMD to itself, chip at working clocks: slope of the launch time over the length of ONE maze
                      # search among 4095 two-step maps (tools/probe_latency.py, profiles/r04/lat_working_clocks.txt; on an otherwise IDLE chip: 280)
FIXED_US = 8.0  # map load + backtrack + output stores of that wavefront


def throughput_regime(dev, steps, workloads=("maze32", "rand32", "rand64"), ks=(2, 3, 4, 6, 8, 12)):
    """What the search sustains when the GPU always has a next batch (a planning service, an evaluation sweep, config 4's 32768-map
    batch): (a) the bench's 4096-map launches issued round-robin over k HIP streams, each stream with its own input AND output buffers
    -- the tail of one batch (its longest search) overlaps the bulk of the next; (b) ONE launch over 32768 maps (8 distinct-memory copies
    of the batch) on one stream: there `hbm_frac` is the roofline fraction in the bench line's own definition (algorithmic bytes of the
    launch / the launch's duration).  Per workload for the general kernel and for the unit-cost LDS layout (NASTAR_FLAG_UNIT_COST:
    cost and passable are one binary tensor, i.e. VanillaAstar; 29 instead of 16 resident 32x32 maps per CU).  Not the headline."""
    from neural_astar.utils import synthetic as syn
    out = []
    for w in workloads:
        pr = make_problem(w, B_PER_GPU, seed=1234)
        for label, flags in (("general", 0), ("unit_cost", FLAG_UNIT_COST)):
            # natural order: a placement is for ONE batch on an otherwise empty chip (latency); with batches in flight it front-loads every
            # launch's long searches and starves the HBM-bound short ones of overlap (rand32: 151 instead of 188 M maps/s)
            runs = [Runner(pr, dev, flags=flags, placement="natural") for _ in range(max(ks))]
            prewarm(runs[0], dev, 0.1)
            nbytes = 24 * runs[0].H * runs[0].W  # cost == passable (one tensor): the bytes that move (28 B/cell figure = x 7/6)
            sweep = {str(k): multi_stream_throughput(pr, steps, dev, k, runs=runs[:k]) for k in ks}
            best_k = max(sweep, key=sweep.get)
            ok = all(int(r.status.abs().sum().item()) == 0 for r in runs)
            del runs
            big = syn.Problems(*(np.concatenate([x] * 8) for x in pr))
            rb = Runner(big, dev, flags=flags)  # (a multi-round launch: order_out ranks the step counts, PLACEMENT applies)
            nbig = max(10, steps // 8)

            def big_ms():
                for _ in range(3):
                    rb.step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(nbig):
                    rb.step()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / nbig * 1e3
            ms_big = big_ms()
            ms_big_nat = None
            if rb.placement == "hinted":
                rb.placement = "natural"
                ms_big_nat = big_ms()
            ok = ok and int(rb.status.abs().sum().item()) == 0
            del rb, big
            out.append({"workload": f"{w}: {B_PER_GPU} maps per launch", "kernel": label,
                        "streams_sweep_maps_per_s": sweep, "best_streams": int(best_k), "maps_per_s": sweep[best_k],
                        "hbm_frac": sweep[best_k] * nbytes / 1e9 / HBM_PEAK_GBS, "hbm_frac_bytes_per_cell": 24,
                        "one_launch_32768_maps": {"ms": ms_big, "maps_per_s": 8 * B_PER_GPU / (ms_big * 1e-3),
                                                  "hbm_frac": 8 * B_PER_GPU * nbytes / (ms_big * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                  "placement": PLACEMENT + (" (maps sorted by the step counts of the previous visit, longest first)" if ms_big_nat else ""),
                                                  "natural_order_ms": ms_big_nat,
                                                  "natural_order_hbm_frac": (8 * B_PER_GPU * nbytes / (ms_big_nat * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_big_nat else None},
                        "all_status_ok": ok})
        del pr
    return out


# ---- what a launch costs the CU's pipes ("report expansions/s against an issue model") ------------------------------------
# Instruction classes of ONE step of the shipped 32x32 stream (nastar_search_asm4.hip.h, g_ratio 0.5 form; counted in the disassembly)
# x the aggregate rates of one CU measured by tools/ubench/rate.hip (profiles/r04/rate.txt; cycles per wavefront instruction with
# >= 2 wavefronts per SIMD / >= 16 per CU).  Round 3 priced every VALU instruction at 4 cycles; the measured machine issues a plain one
# every 2.3 cycles per SIMD and only DPP / compare / lane-read forms at ~4, and the LDS pipe of the CU takes 2.45 (read) / 4.6 (write) /
# 6.0 (64-bit atomic) cycles per instruction WHATEVER the number of active lanes.
STEP_CLASSES = {"valu_plain": 27, "valu_dpp": 7, "valu_cmp": 4, "valu_readlane": 2, "valu_sqrt": 1, "salu_nop_wait_branch": 22, "lds_read": 3,
                "lds_write": 4, "lds_atomic": 1}  # 71 instructions per step (round 3: 76)
RATE_CYCLES = {"valu_plain": 2.3, "valu_dpp": 4.2, "valu_cmp": 4.0, "valu_readlane": 4.0, "valu_sqrt": 8.0, "lds_read": 2.45, "lds_write": 4.6, "lds_atomic": 6.0}


def pipe_model(steps_per_launch, launch_us, max_iters, lone_step_ns, fixed_us):
    valu_cyc = sum(STEP_CLASSES[k] * RATE_CYCLES[k] for k in STEP_CLASSES if k.startswith("valu"))
    lds_cyc = sum(STEP_CLASSES[k] * RATE_CYCLES[k] for k in STEP_CLASSES if k.startswith("lds"))
    clock = 2.4e3  # MHz
    valu_floor = steps_per_launch * valu_cyc / 1024 / clock  # 1024 SIMDs
    lds_floor = steps_per_launch * lds_cyc / 256 / clock  # one LDS pipe per CU
    serial_floor = max_iters * lone_step_ns * 1e-3 + fixed_us
    return {"bound": "serial chain (one launch) / LDS capacity (batches in flight)",
            "steps_per_launch": steps_per_launch, "instructions_per_step": sum(STEP_CLASSES.values()),
            "valu_cycles_per_step_per_simd": valu_cyc, "lds_cycles_per_step_per_cu": lds_cyc,
            "valu_floor_us": valu_floor, "lds_floor_us": lds_floor, "achieved_us": launch_us,
            "valu_busy_frac": valu_floor / launch_us, "lds_busy_frac": lds_floor / launch_us,
            "serial_floor_us": serial_floor, "frac_of_serial_floor": serial_floor / launch_us,
            "serial_floor_note": f"longest search of the batch ({max_iters} steps) x the step of a wavefront that has its SIMD to itself "
                                 f"({lone_step_ns:.0f} ns at working clocks, tools/probe_latency.py, profiles/r04/lat_working_clocks.txt) + {fixed_us:.0f} us "
                                 "load / backtrack / store: what ONE launch cannot beat; frac_of_serial_floor = floor / achieved",
            "source": "instruction classes: disassembly of nastar_forward_compact_kernel<true,5,5,1,true,false,-1> (asm4, g_ratio 0.5); "
                      "rates: profiles/r04/rate.txt (tools/ubench/rate.hip)"}


def two_stream_throughput(pr, steps, dev):
    """Extra (not the headline): the same steps issued round-robin on TWO HIP streams with their own output buffers,
    so the serial tail of one batch (its longest search) overlaps the bulk of the next -- the throughput a planning
    service that always has a next batch would see.  Per-launch latency gets worse, aggregate maps/s better."""
    runs = [Runner(pr, dev, placement="natural"), Runner(pr, dev, placement="natural")]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for i in range(4):
        with torch.cuda.stream(streams[i & 1]):
            runs[i & 1].step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i & 1]):
            runs[i & 1].step()
    torch.cuda.synchronize(dev)
    return runs[0].B * steps / (time.perf_counter() - t0)


def training_step_ms(pr, dev, reps=10):
    """Extra: forward + straight-through backward (nastar_backward_replay) of one 4096-map batch with U(0,1) costs in
    training mode, Tmax = 0.25 (the reference's scripts/config/train.yaml), through the torch custom ops."""
    from neural_astar import ops  # noqa: F401
    from neural_astar.utils import synthetic as syn
    m = torch.from_numpy(pr.map_designs[:, 0]).to(dev)
    s = torch.from_numpy(pr.start_maps[:, 0]).to(dev)
    g = torch.from_numpy(pr.goal_maps[:, 0]).to(dev)
    cost = torch.from_numpy(syn.random_costs(m.shape[0], H, W, seed=3)[:, 0]).to(dev)
    mi = int(0.25 * W * W)
    hist, _, iters, _, _ = torch.ops.nastar.astar_forward(cost, s, g, m, G_RATIO, mi, False)
    gh = torch.randn_like(hist)
    tb = (iters.amax() - 1).to(torch.int32).reshape(1)

    def replay():  # the forward logs its selections, the backward replays them (nastar_backward_replay)
        h, _, it, _, log = torch.ops.nastar.astar_forward(cost, s, g, m, G_RATIO, mi, True)
        torch.ops.nastar.astar_backward_replay(gh, cost, s, g, m, log, G_RATIO, mi, it, tb)
    out = {}
    variants = [("replay_ms", replay)]
    for name, once in variants:
        for _ in range(2):
            once()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = e0.elapsed_time(e1) / reps
    return out


def l1_training_step_ms(pr, dev, batch, reps=20):
    """Extra: the reference's training step on `batch` maps (utils/training.py:55-61, Tmax = 0.25, cost = leaf tensor):
    L1Loss through autograd vs the fused node (nastar_l1_loss + nastar_backward_l1_replay)."""
    from neural_astar import ops
    from neural_astar.utils import synthetic as syn
    m = torch.from_numpy(pr.map_designs[:batch, 0]).to(dev).contiguous()
    s = torch.from_numpy(pr.start_maps[:batch, 0]).to(dev).contiguous()
    g = torch.from_numpy(pr.goal_maps[:batch, 0]).to(dev).contiguous()
    traj = ((torch.rand_like(m) < 0.2).float() * m).contiguous()
    cost = torch.from_numpy(syn.random_costs(batch, H, W, seed=3)[:, 0]).to(dev).requires_grad_(True)
    mi = int(0.25 * W * W)
    l1 = torch.nn.L1Loss()

    def unfused():
        cost.grad = None
        hist, _, _, _, _ = torch.ops.nastar.astar_forward(cost, s, g, m, G_RATIO, mi, True)  # the selection log is the backward's tape
        l1(hist, traj).backward()

    def fused():
        cost.grad = None
        ops.astar_l1_loss(cost, s, g, m, traj, G_RATIO, mi)[0].backward()
    out = {}
    for name, fn in (("autograd_l1loss_ms", unfused), ("fused_ms", fused)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = e0.elapsed_time(e1) / reps
    return out


def data_path_ms(dev, n_maps=400, batch=100):
    """Extra: time to produce one collated training batch of `batch` maze problems (start sampling + optimal-trajectory roll-out):
    the reference-style per-sample host loop (DataLoader over MazeDataset.__getitem__) vs the device-resident loader."""
    import tempfile
    from neural_astar.utils import synthetic as syn
    from neural_astar.utils.data import create_dataloader, create_device_loader
    path = os.path.join(tempfile.mkdtemp(), "mazes.npz")
    syn.write_maze_npz(path, n_train=n_maps, n_valid=1, n_test=1, size=32, seed=11)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        host = create_dataloader(path, "train", batch, shuffle=True)
        devl = create_device_loader(path, "train", batch, dev, shuffle=True)
    t0 = time.perf_counter()
    n = 0
    for b in host:
        b = [x.to(dev, non_blocking=True) for x in b]
        n += 1
    torch.cuda.synchronize(dev)
    host_ms = (time.perf_counter() - t0) * 1e3 / n
    for _ in devl:
        pass
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 0
    for _ in range(5):
        for b in devl:
            n += 1
    torch.cuda.synchronize(dev)
    dev_ms = (time.perf_counter() - t0) * 1e3 / n
    return {"host_dataloader_ms_per_batch": host_ms, "device_loader_ms_per_batch": dev_ms, "batch": batch, "maps": n_maps}


def neural_astar_f16x3_ms(pr, dev, reps=5):
    """Extra: the same NeuralAstar forward with the fp32-grade encoder (encoder_backend="hip_f16x3": cost maps within 1e-5 of the
    fp32 reference encoder, the north-star tolerance for float outputs)."""
    from neural_astar.planner import NeuralAstar
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
    na.encoder_backend = "hip_f16x3"
    na.astar.check_solvable = False
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        for _ in range(2):
            na(m, s, g)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            c = na.encode(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        enc_ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            na(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        full_ms = e0.elapsed_time(e1) / reps
        na.encoder_backend = "torch"   # fp32 torch encoder on a 16-map slice only (MIOpen autotunes per shape; keep it short)
        ref = na.encode(m[:16], s[:16], g[:16])
        err = float((c[:16] - ref).abs().max())
    flop = 3 * 2.0 * m.shape[0] * H * W * 9 * (32 * 64 + 64 * 128 + 128 * 256 + 256)
    return {"encoder_ms": enc_ms, "forward_ms": full_ms, "maps_per_s": m.shape[0] / full_ms * 1e3,
            "max_abs_diff_vs_torch_fp32_encoder_16_maps": err, "mfma_tflops_incl_split_products": flop / enc_ms / 1e9,
            "dtype": "fp16 hi/lo split operands (3 products) / fp32 accumulate"}


def neural_astar_forward_ms(pr, dev, reps=10):
    """Extra (BASELINE config 3 stand-in): NeuralAstar(CNN encoder, depth 4) forward on the bench batch with the bf16-MFMA
    HIP encoder + the HIP search, eval mode.  (The torch/MIOpen encoder is not timed here: its first call autotunes for
    minutes; DESIGN.md quotes it from tools/probe_encoder.py.)"""
    from neural_astar.planner import NeuralAstar
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
    na.encoder_backend = "hip_bf16"
    na.astar.check_solvable = False
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        for _ in range(2):
            na(m, s, g)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            na.encode(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        enc_ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            na(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        full_ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * m.shape[0] * H * W * 9 * (2 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256)
    tf = flop / enc_ms / 1e9
    return {"encoder_ms": enc_ms, "encoder_useful_tflops": tf, "forward_ms": full_ms,
            "maps_per_s": m.shape[0] / full_ms * 1e3, "dtype": "bf16 operands / fp32 accumulate (encoder), f32 (search)",
            "encoder_roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                                 "note": "useful FLOPs of the 5 conv layers / wall time of the whole encoder; the matrix pipe itself "
                                         "sustains 1660 TFLOP/s on random bf16 operands at the power limit (tools/ubench/mfma_peak.hip)"}}


def neural_astar_unet_ms(pr, dev, precision, reps=5):
    """Extra (BASELINE config 3 literally: NeuralAstar UNet encoder + diff-A* on 32x32 mazes, fp16, batch 4096): Unet(vgg16_bn)
    through the generic fp16 MFMA convolution (csrc/nastar_conv_flat.hip.h) + the HIP search, eval mode, random-init weights with
    calibrated BatchNorm statistics.  ``precision``: "f16" (plain fp16 operands) or "f16x3" (split operands, fp32-grade)."""
    from neural_astar.planner import NeuralAstar
    torch.manual_seed(0)
    na = NeuralAstar(encoder_arch="Unet", encoder_depth=4).to(dev)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():  # BatchNorm running statistics = batch statistics of 64 bench maps (activations stay O(1) through 26 layers)
        for mod in na.encoder.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 1.0
        na.train()
        na.encoder(torch.cat((m[:64], s[:64] + g[:64]), dim=1))
    na.eval()
    na.astar.check_solvable = False
    with torch.no_grad():
        ref = na.encode(m[:64], s[:64], g[:64])
        na.encoder_backend = "hip_" + precision
        err = float((na.encode(m[:64], s[:64], g[:64]) - ref).abs().max())
        for _ in range(2):
            na(m, s, g)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            na.encode(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        enc_ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            na(m, s, g)
        e1.record()
        torch.cuda.synchronize(dev)
        full_ms = e0.elapsed_time(e1) / reps
    tf = na._hip_encoder.flops(H, W) * m.shape[0] / enc_ms / 1e9
    products = 3 if precision == "f16x3" else 1
    return {"encoder_ms": enc_ms, "encoder_useful_tflops": tf, "forward_ms": full_ms, "maps_per_s": m.shape[0] / full_ms * 1e3,
            "max_abs_diff_vs_torch_fp32_encoder_64_maps": err,
            "dtype": ("fp16 hi/lo split operands (3 products)" if products == 3 else "fp16 operands") + " / fp32 accumulate (encoder), f32 (search)",
            "encoder_roofline": {"bound": "mfma", "achieved": tf * products, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf * products / 2500.0,
                                 "note": "FLOPs of the 24 conv layers (real channel counts" + (", x3 split products" if products == 3 else "")
                                         + ") / wall time of the whole encoder incl. pooling and input assembly launches"}}


def encoder_train_step_ms(pr, dev):
    """Extra (training): forward + backward of the CNN encoder alone (loss = sum(cost * R)) through the MI355X training
    kernels (neural_astar/encoder_train.py: fp16-MFMA convolutions, input and weight gradients, batch-statistics BatchNorm), at the
    reference's training batch (100 maps) and at the bench batch.  The fp32 torch.nn encoder on the same box: 4.07 / 132.8 ms per
    100 / 4096 maps (profiles/r02/encoder_train_step_ms.json; not re-timed here, MIOpen's autotuning takes minutes)."""
    from neural_astar.planner import NeuralAstar
    out = {}
    for B in (100, 4096):
        m, s, g = (torch.from_numpy(x[:B]).to(dev) for x in pr)
        R = torch.randn((B, 1, H, W), device=dev) / (B * H * W)
        for backend in ("hip_f16x3", "hip_f16"):
            torch.manual_seed(0)
            na = NeuralAstar(encoder_arch="CNN").to(dev).train()
            na.encoder_backend = backend

            def one():
                for p in na.parameters():
                    p.grad = None
                (na.encode(m, s, g) * R).sum().backward()
            one()
            torch.cuda.synchronize(dev)
            reps = 10 if B == 100 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                one()
            torch.cuda.synchronize(dev)
            out[f"batch_{B}_{backend}_ms"] = (time.perf_counter() - t0) / reps * 1e3
            del na
    # BASELINE config 5's encoder: CNNDownSize, rgb+, depth 3, 96x96 RGB -> 12x12, the reference's batch of 100 (train_warcraft.yaml)
    img = torch.rand((100, 3, 96, 96), device=dev)
    s12 = torch.zeros((100, 1, 12, 12), device=dev)
    g12 = torch.zeros((100, 1, 12, 12), device=dev)
    s12[:, 0, 0, 0] = 1
    g12[:, 0, 11, 11] = 1
    R12 = torch.randn((100, 1, 12, 12), device=dev) / 14400
    for backend in ("hip_f16x3", "hip_f16"):
        torch.manual_seed(0)
        na = NeuralAstar(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0).to(dev).train()
        na.encoder_backend = backend

        def one_w():
            for p in na.parameters():
                p.grad = None
            (na.encode(img, s12, g12) * R12).sum().backward()
        one_w()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            one_w()
        torch.cuda.synchronize(dev)
        out[f"warcraft_batch_100_{backend}_ms"] = (time.perf_counter() - t0) / 10 * 1e3
        del na
    # Unet(vgg16_bn) (BASELINE config 3's encoder), 100 maps: ~330 launches per step, launch-bound at this batch
    m, s, g = (torch.from_numpy(x[:100]).to(dev) for x in pr)
    R = torch.randn((100, 1, H, W), device=dev) / (100 * H * W)
    for backend in ("hip_f16x3", "hip_f16"):
        torch.manual_seed(0)
        na = NeuralAstar(encoder_arch="Unet", encoder_depth=4).to(dev).train()
        na.encoder_backend = backend

        def one_u():
            for p in na.parameters():
                p.grad = None
            (na.encode(m, s, g) * R).sum().backward()
        one_u()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            one_u()
        torch.cuda.synchronize(dev)
        out[f"unet_batch_100_{backend}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        del na
    out["torch_fp32_ms_same_box"] = {"batch_100": 4.07, "batch_4096": 132.75, "warcraft_batch_100": 3.13, "unet_batch_100": 6.65,
                                     "source": "profiles/r02/encoder_train_step_ms.json, encoder_train_step_warcraft_b100_ms.json, "
                                               "encoder_train_step_unet_b100_ms.json"}
    out["unit"] = "ms per encoder forward+backward (wall clock), random-init CNN depth 4, 32x32 maps"
    return out


def kernel_launch_ms(run, steps, dev):
    """Average duration of one launch from HIP events recorded on the stream the kernel is launched on
    (torch's current stream), one event pair per launch."""
    steps = max(steps, 100)  # not the K timed steps of the contract: 20 event pairs right after an idle gap measure the gap
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for _ in range(10):
        run.step()
    evs[0].record()
    for i in range(steps):
        run.step()
        evs[i + 1].record()
    torch.cuda.synchronize(dev)
    d = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return sum(d) / len(d), d[len(d) // 2], d[0]


def oracle_check(pr, gpu_hist, gpu_paths, n, g_ratio=G_RATIO):
    """Parity of the first n maps of a bench batch against the CPU oracle (the checker, never the thing measured)."""
    from oracle import oracle as O
    O.build()
    Ww = pr.map_designs.shape[-1]
    o = O.forward(pr.map_designs[:n], pr.start_maps[:n], pr.goal_maps[:n], pr.map_designs[:n], g_ratio, Ww * Ww)
    return bool(np.array_equal(o.histories, gpu_hist[:n]) and np.array_equal(o.paths, gpu_paths[:n]))


def cpu_baseline_port(pr, gpu_hist, gpu_paths):
    """Time the CPU oracle (literal C port of the reference's tensor program, OpenMP over maps) on a bounded sample
    of the SAME workload, and use its outputs to parity-check the GPU results of those maps."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    O.build()
    Ww = pr.map_designs.shape[-1]
    n0 = min(512, pr.map_designs.shape[0])
    O.forward(pr.map_designs[:8], pr.start_maps[:8], pr.goal_maps[:8], pr.map_designs[:8], G_RATIO, Ww * Ww)  # spin up the OpenMP team
    t0 = time.perf_counter()
    O.forward(pr.map_designs[:n0], pr.start_maps[:n0], pr.goal_maps[:n0], pr.map_designs[:n0], G_RATIO, Ww * Ww)
    rate0 = n0 / max(time.perf_counter() - t0, 1e-6)
    n = int(min(pr.map_designs.shape[0], max(n0, rate0 * 10.0)))  # aim at ~10 s of CPU work (bounded by the batch)
    t0 = time.perf_counter()
    o = O.forward(pr.map_designs[:n], pr.start_maps[:n], pr.goal_maps[:n], pr.map_designs[:n], G_RATIO, Ww * Ww)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(o.histories, gpu_hist[:n]) and np.array_equal(o.paths, gpu_paths[:n]))
    return {"value": n / dt, "unit": "maps/s", "cores": cores, "kind": "port",
            "sample": f"first {n} maps of the bench batch, oracle/nastar_oracle.c dense restatement, OpenMP over maps, {dt:.1f} s",
            "gpu_matches_oracle_on_sample": ok}


REF_STAGED = os.path.join(ROOT, "oracle", "_ref", "differentiable_astar.py")


def _reference_worker(path: str) -> None:
    """Child process of cpu_baseline_reference (fresh process: no OpenMP team of the C port spinning beside torch's threads)."""
    import importlib.util
    z = np.load(path)
    spec = importlib.util.spec_from_file_location("ref_differentiable_astar", REF_STAGED)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cores = int(z["threads"])
    torch.set_num_threads(cores)
    planner = ref.DifferentiableAstar(g_ratio=G_RATIO, Tmax=1.0).eval()
    budget = float(z["budget_s"])
    # the reference pays ~45 ATen dispatches (+ a thread-team hand-off each) per loop iteration whatever the batch size, so its
    # rate grows with the batch: start at 1024 maps, then the whole 4096-map batch if the budget allows
    best, spent, n = None, 0.0, (int(z["n_fixed"]) if "n_fixed" in z.files and int(z["n_fixed"]) > 0 else 1024)
    with torch.no_grad():
        m, s_, g = (torch.from_numpy(z[k][:32]) for k in ("m", "s", "g"))
        planner(m, s_, g, m)  # untimed: thread pool start-up, first-touch of the allocator (BASELINE config 1's size)
        while True:
            n = min(n, z["m"].shape[0])
            m, s_, g = (torch.from_numpy(z[k][:n]) for k in ("m", "s", "g"))
            t0 = time.perf_counter()
            out = planner(m, s_, g, m)
            dt = time.perf_counter() - t0
            spent += dt
            ok = bool(np.array_equal(out.histories[:, 0].numpy(), z["hist"][:n]) and np.array_equal(out.paths[:, 0].numpy(), z["paths"][:n]))
            if best is None or n / dt > best[0]:
                best = (n / dt, n, dt, ok)
            if n >= z["m"].shape[0] or spent + 4.5 * dt > budget:
                break
            n *= 4
    print(json.dumps({"rate": best[0], "n": best[1], "dt": best[2], "ok": best[3], "threads": cores, "torch": torch.__version__}))


def cpu_baseline_spec_start(pr, gpu_hist, gpu_paths):
    """BASELINE.md section 3's exact configuration -- ONE forward() call on all 4096 maps with torch.set_num_threads(os.cpu_count()) --
    in a child process with a hard time limit, run after everything else (cpu_baseline["spec_config"]).  On a 256-core host the
    fork/join of 256 ATen threads around ~45 tiny elementwise kernels per search iteration makes this configuration SLOWER than the
    32-thread sample above; when it does not finish inside the limit that fact (an upper bound on its rate) is what is reported."""
    import subprocess
    import tempfile
    if not os.path.exists(REF_STAGED):
        return None
    cores = os.cpu_count() or 1
    n = min(pr.map_designs.shape[0], 4096)
    path = os.path.join(tempfile.mkdtemp(), "ref_in_spec.npz")
    np.savez(path, m=pr.map_designs[:n], s=pr.start_maps[:n], g=pr.goal_maps[:n], hist=gpu_hist[:n], paths=gpu_paths[:n],
             threads=cores, budget_s=1.0, n_fixed=n)
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores))
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ref-worker", path], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=env)
    return proc, cores, time.perf_counter()


def cpu_baseline_spec_collect(handle, timeout_s=120.0):
    if handle is None:
        return {"available": False, "note": "oracle/_ref not staged"}
    proc, cores, t0 = handle
    try:
        out, err = proc.communicate(timeout=max(1.0, timeout_s - (time.perf_counter() - t0)))
    except Exception:  # noqa: BLE001 - a slow host must not sink the bench line
        proc.kill()
        return {"available": False, "cores": cores,
                "note": f"ONE forward() call on 4096 maps with torch.set_num_threads({cores}) did not finish within {timeout_s:.0f} s "
                        f"(i.e. < {4096 / timeout_s:.0f} maps/s); the 32-thread sample above is the faster configuration on this host"}
    if proc.returncode != 0:
        return {"available": False, "note": err[-300:]}
    j = json.loads(out.strip().splitlines()[-1])
    return {"available": True, "value": j["rate"], "unit": "maps/s", "cores": j["threads"], "kind": "reference",
            "sample": f"BASELINE.md 3 as specified: reference forward() on all {j['n']} maps in ONE call, torch.set_num_threads({j['threads']}) "
                      f"= os.cpu_count(), {j['dt']:.2f} s",
            "gpu_matches_reference_on_sample": j["ok"]}


def cpu_baseline_reference(pr, gpu_hist, gpu_paths):
    """The reference's OWN DifferentiableAstar.forward() (eval mode, no_grad) on this box's host cores: the module file staged
    into the git-ignored oracle/_ref/ by `__graft_entry__.build()` in the authoring container (it is torch-only and travels with
    gpurun like a built .so; /root/reference itself is never read here).  Bounded sample = BASELINE.md section 3's batch: ONE call on
    all 4096 maps of the bench batch (~7 s at 32 threads) after an untimed 32-map call, in a fresh child process with a hard time
    limit.  Threads: 32 of the host's cores -- ATen's elementwise kernels on [B,32,32] maps stop scaling long before 256 threads; the
    all-cores configuration BASELINE.md names is attempted last as `spec_config`."""
    import subprocess
    import tempfile
    cores = os.cpu_count() or 1
    threads = min(cores, 32)  # ATen's elementwise kernels on [B,32,32] maps stop scaling long before 256 threads
    n = min(pr.map_designs.shape[0], 4096)
    path = os.path.join(tempfile.mkdtemp(), "ref_in.npz")
    np.savez(path, m=pr.map_designs[:n], s=pr.start_maps[:n], g=pr.goal_maps[:n], hist=gpu_hist[:n], paths=gpu_paths[:n],
             threads=threads, budget_s=25.0, n_fixed=n)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--ref-worker", path], capture_output=True, text=True,
                       timeout=150, env=env)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-400:])
    j = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": j["rate"], "unit": "maps/s", "cores": j["threads"], "kind": "reference",
            "sample": f"reference DifferentiableAstar.forward (torch {j['torch']} CPU, torch.set_num_threads({j['threads']}) of "
                      f"{cores} host cores, eval, no_grad) on all {j['n']} maps of the bench batch in ONE call, {j['dt']:.2f} s",
            "gpu_matches_reference_on_sample": j["ok"]}


def reference_on_this_gpu(pr, gpu_hist, gpu_paths, dev):
    """Extra: the REAL reference DifferentiableAstar.forward (the staged torch-only module, oracle/_ref/) run through PyTorch-ROCm on
    the SAME MI355X -- what a user of the reference gets on this hardware without this package (~45 ATen launches + one device->host
    sync per loop iteration).  One warm-up call on 256 maps, then the whole bench batch once; masks compared with the HIP kernel's."""
    import importlib.util
    if not os.path.exists(REF_STAGED):
        return {"available": False, "note": "oracle/_ref/differentiable_astar.py not staged (built outside the authoring container)"}
    spec = importlib.util.spec_from_file_location("ref_differentiable_astar_gpu", REF_STAGED)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    planner = ref.DifferentiableAstar(g_ratio=G_RATIO, Tmax=1.0).to(dev).eval()
    m, s_, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        planner(m[:256], s_[:256], g[:256], m[:256])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = planner(m, s_, g, m)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    ok = bool(np.array_equal(out.histories[:, 0].cpu().numpy(), gpu_hist) and np.array_equal(out.paths[:, 0].cpu().numpy(), gpu_paths))
    return {"available": True, "value": m.shape[0] / dt, "unit": "maps/s", "seconds_per_batch": dt, "batch": int(m.shape[0]),
            "torch": torch.__version__, "masks_equal_to_hip_kernel": ok,
            "note": "reference differentiable_astar.py on the same GPU via PyTorch-ROCm, eval mode, no_grad"}


def cpu_baseline(pr, gpu_hist, gpu_paths):
    port = cpu_baseline_port(pr, gpu_hist, gpu_paths)
    if os.path.exists(REF_STAGED):
        try:
            ref = cpu_baseline_reference(pr, gpu_hist, gpu_paths)
            ref["port"] = port
            ref["gpu_matches_oracle_on_sample"] = port["gpu_matches_oracle_on_sample"]
            return ref
        except Exception as e:  # reported, not hidden
            port["reference_error"] = f"{type(e).__name__}: {e}"
    return port


def through_module_ms(pr, dev, reps=60):
    """End to end through the drop-in boundary (north_star: "keeps the forward() API"): ms per VanillaAstar.forward() call on
    the bench batch, wall clock -- output allocation, the launch, and the solvability policy: the default (True = "sync") waits for the
    kernel and raises in the same call (one stream wait + one 64-byte read of the pinned status summary the launch wrote), "deferred"
    (opt-in) hands the verdict to a later call (an event, no host wait), False skips it.  The batch carries the placement its loader
    attached (start_maps.placement_order, by the optimal distance of the start cells); `no_placement_*` = the same calls without it."""
    from neural_astar import ops
    from neural_astar.planner import VanillaAstar
    m, s_, g = (torch.from_numpy(x).to(dev) for x in (pr.map_designs, pr.start_maps, pr.goal_maps))
    dist = _device_distances(m[:, 0], g[:, 0]).reshape(m.shape[0], -1)
    levels = (dist * (s_.reshape(m.shape[0], -1) > 0)).sum(1).to(torch.int32).contiguous()
    out = {}
    for hinted in (True, False):
        if hinted:
            ops.attach_order(s_, levels)
        elif hasattr(s_, "placement_order"):
            del s_.placement_order
        for label, chk in (("check_solvable_default_sync", True), ("check_solvable_deferred", "deferred"), ("check_solvable_false", False)):
            va = VanillaAstar().to(dev).eval()
            va.astar.check_solvable = chk
            with torch.no_grad():
                for _ in range(5):
                    va(m, s_, g)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(reps):
                    va(m, s_, g)
                torch.cuda.synchronize(dev)
            out[("" if hinted else "no_placement_") + label] = (time.perf_counter() - t0) / reps * 1e3
            va.astar.raise_if_unsolvable()
    out["unit"] = "ms per VanillaAstar.forward() call, wall clock, same input batch each call"
    out["note"] = ("all three modes run the general kernel (forward() takes the unit-cost layout only with unit_cost=True); the default waits for "
                   "the launch's completion flag in the same call, deferred / false issue the launches back to back without a host wait, i.e. "
                   "run at the kernel's own duration")
    return out


def in_flight_through_api(dev, steps, workloads=(("maze32", True), ("rand32", True), ("rand64", True), ("maze32", False)), ks=(3, 4, 6, 8, 12)):
    """Batches in flight THROUGH THE PYTHON OBJECT (neural_astar.parallel.InFlightPlanner around a VanillaAstar): whole 4096-map batches
    round-robin over k HIP streams, outputs allocated per batch, status summaries read once at collection, unit_cost="auto" without a
    per-call wait.  maps/s over `n` batches incl. submission, collection and the final host wait; outputs checked equal to sequential
    planner.forward() calls on the first batches."""
    from neural_astar.parallel import InFlightPlanner
    from neural_astar.planner import VanillaAstar
    res = []
    n = max(144, min(steps, 288))  # (not the K timed steps of the contract: 20 batches are a pipeline that never fills)
    for w, unit in workloads:
        prs = [make_problem(w, B_PER_GPU, seed=1234 + 1000 * k) for k in range(N_ROTATE)]
        batches = [tuple(torch.from_numpy(x).to(dev) for x in (pr.map_designs, pr.start_maps, pr.goal_maps)) for pr in prs]
        va = VanillaAstar().to(dev).eval()
        with torch.no_grad():
            seq = [va(*b) for b in batches]
        sweep = {}
        same = True
        for k in ks:
            fly = InFlightPlanner(va, streams=k, unit_cost="auto" if unit else False)
            outs = fly.plan_many(batches[i % N_ROTATE] for i in range(n))  # warm-up (the allocator then holds n output sets) + equality with the sequential calls
            same = same and all(torch.equal(o.histories, seq[i % N_ROTATE].histories) and torch.equal(o.paths, seq[i % N_ROTATE].paths)
                                for i, o in enumerate(outs))
            del outs
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            outs = fly.plan_many(batches[i % N_ROTATE] for i in range(n))
            dt = time.perf_counter() - t0
            del outs
            sweep[str(k)] = n * B_PER_GPU / dt
        best = max(sweep, key=sweep.get)
        Hh, Ww = batches[0][0].shape[-2:]
        res.append({"workload": f"{w}: {B_PER_GPU} maps per batch, {n} batches", "kernel": "unit_cost (auto)" if unit else "general",
                    "streams_sweep_maps_per_s": sweep, "best_streams": int(best), "maps_per_s": sweep[best],
                    "hbm_frac": sweep[best] * 24 * Hh * Ww / 1e9 / HBM_PEAK_GBS, "equal_to_sequential_forward": bool(same)})
        del batches, seq
    return res


# ---- --mode train: BASELINE config 5 (and the maze configuration of scripts/train.py) as a driver-runnable training bench ---------------
TRAIN_CONFIGS = {
    # constructor arguments = the reference's scripts (scripts/train.py:33-39 + config/train.yaml, scripts/train_warcraft.py:33-40 +
    # config/train_warcraft.yaml); batch_size 100 per step in both
    "maze": dict(kw=dict(encoder_input="m+", encoder_arch="CNN", encoder_depth=4, Tmax=0.25), chans=[2, 32, 64, 128, 256, 1], pool=False, hw=(32, 32)),
    "warcraft": dict(kw=dict(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0, learn_obstacles=True, Tmax=0.25),
                     chans=[4, 32, 64, 128, 1], pool=True, hw=(96, 96)),
}


def train_batch(config, B, seed, dev):
    """synthetic training batch in the reference loaders' layout (utils/data.py): (map_designs, start_maps, goal_maps, opt_trajs)"""
    from neural_astar.planner import VanillaAstar
    from neural_astar.utils import synthetic as syn
    if config == "warcraft":
        g_ = torch.Generator().manual_seed(seed)
        tiles = torch.rand((B, 3, 12, 12), generator=g_)
        img = (tiles.repeat_interleave(8, 2).repeat_interleave(8, 3) + 0.08 * torch.randn((B, 3, 96, 96), generator=g_)).clamp_(0, 1).to(dev)
        s = torch.zeros((B, 1, 12, 12), device=dev)
        gl = torch.zeros((B, 1, 12, 12), device=dev)
        s[:, 0, 0, 0] = 1
        gl[:, 0, -1, -1] = 1
        true_cost = (0.1 + 0.9 * torch.rand((B, 1, 12, 12), generator=g_)).to(dev)
        with torch.no_grad():  # the label of a WarCraft sample is the shortest path under hidden per-tile costs
            traj = VanillaAstar().to(dev).eval().astar(true_cost, s, gl, torch.ones_like(s)).paths.float()
        return img, s, gl, traj
    pr = syn.maze_maps(B, 32, seed=seed)
    m, s, gl = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        traj = VanillaAstar().to(dev).eval()(m, s, gl).paths.float()
    return m, s, gl, traj


def train_flops_per_map(config):
    """useful convolution FLOPs of one training step per map: forward + input gradient (not for the first layer) + weight gradient"""
    c = TRAIN_CONFIGS[config]
    h, w = c["hw"]
    fwd, total = 0.0, 0.0
    for l, (ci, co) in enumerate(zip(c["chans"][:-1], c["chans"][1:])):
        f = 2.0 * 9 * h * w * ci * co
        fwd += f
        total += f * (2 if l == 0 else 3)
        if c["pool"] and l < len(c["chans"]) - 2:
            h, w = h // 2, w // 2
    return fwd, total


def train_cpu_baseline(config, B, budget_s=25.0):
    """The reference training step on the host cores: the reference's OWN DifferentiableAstar (staged oracle/_ref module, ~45 ATen ops
    per search iteration under autograd) behind this package's torch.nn encoder (tests/test_reference_modules_cpu.py pins it to the
    reference's encoder classes: identical cost maps and gradients), nn.L1Loss, RMSprop -- utils/training.py:55-61 as the reference
    runs it on a CPU.  A bounded sample: `B` maps, as many steps as fit the budget (at least 1)."""
    import importlib.util
    from neural_astar.planner import NeuralAstar
    path = os.path.join(ROOT, "oracle", "_ref", "differentiable_astar.py")
    if not os.path.exists(path):
        return {"available": False, "note": "oracle/_ref not staged (run __graft_entry__.build() where /root/reference exists)"}
    spec = importlib.util.spec_from_file_location("ref_da_train", path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cores = min(32, os.cpu_count() or 1)  # small per-iteration tensors: more threads only add fork/join overhead (as in cpu_baseline)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    na = NeuralAstar(**TRAIN_CONFIGS[config]["kw"]).train()
    astar = ref.DifferentiableAstar(g_ratio=0.5, Tmax=0.25).train()
    opt = torch.optim.RMSprop(na.parameters(), 1e-3)
    g_ = torch.Generator().manual_seed(5)
    if config == "warcraft":
        m = torch.rand((B, 3, 96, 96), generator=g_)
        s = torch.zeros((B, 1, 12, 12)); gl = torch.zeros((B, 1, 12, 12))
        s[:, 0, 0, 0] = 1; gl[:, 0, -1, -1] = 1
        traj = torch.zeros((B, 1, 12, 12)); traj[:, 0, torch.arange(12), torch.arange(12)] = 1
        passable = torch.ones_like(s)
    else:
        from neural_astar.utils import synthetic as syn
        pr = syn.maze_maps(B, 32, seed=9)
        m, s, gl = (torch.from_numpy(x) for x in pr)
        traj = (torch.rand(m.shape, generator=g_) < 0.1).float() * m
        passable = m

    def step():
        opt.zero_grad(set_to_none=True)
        out = astar(na.encode(m, s, gl), s, gl, passable)
        torch.nn.L1Loss()(out.histories, traj).backward()
        opt.step()
    step()
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < 20):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": B / dt, "unit": "maps/s", "cores": cores, "kind": "reference",
            "sample": f"{n} training step(s) of {B} maps after 1 warm-up: the reference's DifferentiableAstar under autograd (staged module) + "
                      f"torch.nn encoder + L1Loss + RMSprop on {cores} host threads", "ms_per_step": dt * 1e3}


def train_main(args, real_stdout):
    """`bench.py --mode train --config maze|warcraft [--gpus N]`: the reference's training step (planner forward, L1 loss on histories,
    straight-through backward, RMSprop) with encoder AND search on the MI355X kernels; N > 1 = DataParallelTrainer over RCCL (each
    rank its own `--batch-per-gpu` maps: weak scaling; BatchNorm statistics of the global batch, one flat gradient all-reduce)."""
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import distributed as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1 or args.force_collate:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gpus = max(world, 1)
    B = args.batch_per_gpu
    cfg = TRAIN_CONFIGS[args.config]
    torch.manual_seed(1234)
    planner = NeuralAstar(**cfg["kw"]).to(dev)
    planner.astar.check_solvable = "deferred"  # opt-in: no host sync inside the timed steps; the verdicts are collected after the loop
    planner.encoder_backend = args.encoder_backend  # "auto" (the package default) resolves to hip_f16x3 on a HIP device
    if args.encoder_backend == "auto":
        args.encoder_backend = planner.effective_encoder_backend(torch.empty(0, device=dev))
    multi = dist.is_initialized() and (world > 1 or args.force_collate)  # --force-collate: the RCCL path in a 1-rank group
    sync_bn = multi and args.encoder_backend.startswith(("hip", "auto"))
    trainer = D.DataParallelTrainer(planner, lr=1e-3, coupling="global" if multi else "local", sync_bn=sync_bn,
                                    force_collectives=args.force_collate)
    batches = [train_batch(args.config, B, 1234 + 17 * rank + 1000 * k, dev) for k in range(4)]  # a few distinct batches in rotation
    _log(f"train mode: {args.config}, {B} maps/GPU, encoder_backend={args.encoder_backend}, world={world}, sync_bn={sync_bn}")

    def run(n):
        last = None
        for i in range(n):
            last = trainer.train_step(*batches[i % len(batches)])
        return last
    t_pre = time.perf_counter()  # untimed: clocks out of their idle state (see PREWARM_S)
    while time.perf_counter() - t_pre < PREWARM_S:
        run(2)
        torch.cuda.synchronize(dev)
    import gc
    gc.collect()  # (before the warm-up steps; the cyclic collector then stays out of the timed region, as in timed_loop)
    gc.disable()
    run(args.warmup)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    loss = run(args.steps)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gc.enable()
    planner.astar.raise_if_unsolvable()  # the deferred verdicts of every step above
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the same step WITHOUT sync BatchNorm (every rank normalises with its own rows: no per-layer collectives, only the flat gradient
    # all-reduce) so that a scaling curve can separate the number of small collectives from wire time
    dt_nosync = None
    if multi and sync_bn:
        trainer_ns = D.DataParallelTrainer(planner, lr=1e-3, coupling="global", sync_bn=False, force_collectives=args.force_collate)

        def run_ns(n):
            for i in range(n):
                trainer_ns.train_step(*batches[i % len(batches)])
        run_ns(max(2, args.warmup))
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        run_ns(args.steps)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_nosync = float(t.item())
        planner.astar.raise_if_unsolvable()
    if rank == 0:
        fwd_f, step_f = train_flops_per_map(args.config)
        ms = dt / args.steps * 1e3
        split = args.encoder_backend == "hip_f16x3"
        out = {
            "metric": f"map-instances/s (NeuralAstar TRAINING step, {args.config} configuration, Tmax 0.25, batch {B}/GPU)",
            "value": n_gpus * B * args.steps / dt, "unit": "maps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"hip_f16x3": "f16x3 (split fp16 operands, fp32 accumulation: fp32-grade)", "hip_f16": "f16 (fp32 accumulation)",
                      "torch": "f32"}[args.encoder_backend] + " encoder, f32 search",
            "data": "synthetic",
            "config": {"workload": f"train/{args.config}: NeuralAstar({', '.join(f'{k}={v}' for k, v in cfg['kw'].items())}), {B} maps/GPU per step, "
                                   f"encoder forward+backward on {'the MI355X training kernels' if args.encoder_backend.startswith('hip') else 'torch.nn (MIOpen)'}, "
                                   "HIP search forward + replay backward, fused L1 loss, RMSprop(lr 1e-3); random-init weights",
                       "batch_per_gpu": B, "global_batch": B * n_gpus, "encoder_backend": args.encoder_backend,
                       "parallelism": (f"dp{n_gpus}: flat fp32 gradient all-reduce ({'RCCL' if args.dist_backend == 'nccl' else args.dist_backend}) + all-reduced BatchNorm sums (sync_bn={sync_bn}), "
                                       "coupling=global") if multi else "single"},
            "steps_per_s": args.steps / dt, "final_loss": float(loss),
            "sync_bn": {"on_ms_per_step": ms if sync_bn else None, "off_ms_per_step": (dt_nosync / args.steps * 1e3) if dt_nosync else (None if sync_bn else ms),
                        "note": "on = BatchNorm statistics of the GLOBAL batch (one small all-reduce per BatchNorm layer and direction: "
                                "the single-device step on the concatenated batch); off = per-rank statistics, only the flat gradient "
                                "all-reduce; `value` / `ms_per_step` are the sync_bn=on figures when n_gpus > 1"},
            "roofline": {"bound": "mfma", "achieved": B * step_f / (ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": B * step_f / (ms * 1e-3) / 1e12 / 2500.0, "traffic": None,
                         "kernel": "WHOLE STEP, not one kernel: useful convolution FLOPs (forward + input gradient + weight gradient, "
                                   "no split-operand products counted) / wall time of the step; at 100 maps the step is launch-bound",
                         "useful_flops_per_map": step_f, "forward_flops_per_map": fwd_f,
                         "matrix_products_issued_x": 3 if split else 1},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            _log("train: same step with the torch.nn encoder on this GPU")
            try:
                torch.manual_seed(1234)
                p2 = NeuralAstar(**cfg["kw"]).to(dev)
                t2 = D.DataParallelTrainer(p2, lr=1e-3, coupling="local")
                for i in range(3):
                    t2.train_step(*batches[i % len(batches)])
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(max(5, args.steps // 4)):
                    t2.train_step(*batches[i % len(batches)])
                torch.cuda.synchronize(dev)
                out["torch_encoder_on_this_gpu"] = {"ms_per_step": (time.perf_counter() - t0) / max(5, args.steps // 4) * 1e3,
                                                    "note": "HIP search kernels + torch.nn (MIOpen fp32) encoder"}
            except Exception as e:  # noqa: BLE001
                out["torch_encoder_on_this_gpu"] = {"available": False, "note": f"{type(e).__name__}: {e}"}
            _log("train: cpu baseline")
            try:
                out["cpu_baseline"] = train_cpu_baseline(args.config, min(B, 100))
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"available": False, "note": f"{type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--ref-worker":
        _reference_worker(sys.argv[2])
        return
    # stdout carries exactly ONE line (the JSON): anything a library prints there meanwhile (RCCL / c10d notices) goes to stderr
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="maze32", choices=["maze32", "rand32", "rand64"],
                    help="maze32 = BASELINE config 2 (headline); rand64 = the 64x64 maps of config 4")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: this many maps in total, global_batch/N rows per rank (config 4: --workload rand64 "
                         "--global-batch 32768); default 0 = weak scaling, 4096 maps per GPU")
    ap.add_argument("--shard", default="contiguous", choices=["contiguous", "interleaved"],
                    help="strong scaling: which rows a rank owns (parallel.shard_rows)")
    ap.add_argument("--placement", default="dataset", choices=["dataset", "hinted", "natural"],
                    help="dataset (default): every timed step searches a NEVER-SEARCHED batch, placed longest-first by the optimal distance of its "
                         "start cells (data the sample carries; order computed at batch assembly); hinted: three recurring batch sets, each placed by the "
                         "order its searches finished in at its previous visit (the round-4 headline; reported beside the headline anyway); natural: "
                         "workgroup i = map i")
    ap.add_argument("--no-natural", action="store_true", help="skip the natural-order and hinted comparison passes (clean kernel profiles)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the untimed clock pre-warm launches (clean kernel profiles: every launch of the run is then a dataset-placed one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (clean kernel profiles)")
    ap.add_argument("--no-collate", action="store_true", help="N>1: skip the all-gather of AstarOutput")
    ap.add_argument("--no-priority-stream", action="store_true",
                    help="collated steps: launch the search on torch's default stream instead of parallel.search_stream() (A/B)")
    ap.add_argument("--force-collate", action="store_true",
                    help="dev: run the N>1 collation path (pack kernel + all-gather) in a 1-rank RCCL group")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = the headline (BASELINE config 2 / 4); train = one full NeuralAstar training step per bench step "
                         "(BASELINE config 5 with --config warcraft)")
    ap.add_argument("--config", default="warcraft", choices=sorted(TRAIN_CONFIGS), help="--mode train: which reference training configuration")
    ap.add_argument("--batch-per-gpu", type=int, default=100, help="--mode train: maps per rank and step (the reference's batch_size is 100)")
    ap.add_argument("--encoder-backend", default="hip_f16x3", choices=["auto", "hip_f16x3", "hip_f16", "torch"],
                    help="--mode train (auto = the package default = hip_f16x3 on a HIP device)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="dev: gloo lets N ranks share ONE GPU (with --share-gpu) to exercise every world > 1 branch of this script on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="dev: every rank uses cuda:0 (only with --dist-backend gloo)")
    args = ap.parse_args()
    global PLACEMENT
    PLACEMENT = "hinted" if args.placement == "dataset" else args.placement  # (the secondary workloads run on recurring batch sets)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: re-launch one rank per GPU the way the driver does
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env, stdout=real_stdout))
    if args.mode == "train":
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback in the product path)"
        train_main(args, real_stdout)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_collate:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gpus = world if world > 1 else 1
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback in the product path)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    strong = args.global_batch > 0
    if strong:
        # strong scaling: ONE global batch (seeded independently of N), every rank searches its rows of it
        from neural_astar import parallel as _par
        assert args.global_batch % n_gpus == 0, "--global-batch must be a multiple of the number of GPUs"
        b_rank = args.global_batch // n_gpus
        rows = _par.shard_rows(args.global_batch, n_gpus, rank, args.shard).numpy()
        prs = [make_problem_rows(args.workload, args.global_batch, 1234 + 1000 * k, rows, rank, n_gpus) for k in range(N_ROTATE)]
    else:
        b_rank = B_PER_GPU
        prs = [make_problem(args.workload, B_PER_GPU, seed=1234 + rank + 1000 * k) for k in range(N_ROTATE)]
    pr = prs[0]
    # the resident pool (maps, goals, distance-to-goal maps: a planning-datasets split) the loader-style batches are drawn from
    pool = FreshBatches(args.workload, dev, prs, seed=4321 + rank)
    per_pass = args.steps + args.warmup
    if args.placement == "dataset" and not strong:
        # every step of the cold pass AND of the headline pass searches a batch that has never been searched: 2 (W + K) of them
        # (+ 2: the collate trial step), capped at MAX_FRESH_SETS -- beyond the cap the loop revisits batches (same dataset order, nothing learned)
        n_fresh = min(2 * per_pass + 2, MAX_FRESH_SETS)
        sets = [pool.batch(b_rank) for _ in range(n_fresh)]
    else:
        sets = [pool.fixed(k, b_rank) for k in range(N_ROTATE)]
        n_fresh = 0
    torch.cuda.synchronize(dev)
    run = Runner(sets, dev, placement=args.placement)
    warm = Runner([pool.batch(b_rank) for _ in range(N_ROTATE)] if not strong else sets, dev, placement=args.placement if args.placement != "hinted" else "natural")
    Hh, Ww = run.H, run.W
    bytes_per_map = 28 * Hh * Ww  # four input tensors
    bytes_moved_per_map = 24 * Hh * Ww  # ... of which VanillaAstar's call passes ONE as cost and passable (reference astar.py:93-94): what really moves

    collate = None
    collate_note = "n/a (single GPU)"
    if (world > 1 or args.force_collate) and not args.no_collate:
        from neural_astar import parallel

        run.enable_packed()

        def collate(pending):
            # all-gather of step i overlaps the search of step i+1: wait for the previous one only now
            _, fin = parallel.all_gather_packed(run.packed[run._pk], async_op=True)
            if pending is not None:
                pending()
            return fin
        try:
            run.step()
            collate(None)()
            torch.cuda.synchronize(dev)
            collate_note = (f"bit-packed histories+paths emitted by the search launch itself, 1 all-gather per step over "
                            f"{'RCCL' if args.dist_backend == 'nccl' else args.dist_backend} (kept packed), overlapped with the next step's search")
        except Exception as e:  # reported, not hidden: the line then says the collective was not part of the step
            collate = None
            collate_note = f"FAILED ({type(e).__name__}: {e}); steps timed without the all-gather"

    _log("problems resident, pre-warming the clocks, then timing the headline loop")
    import contextlib
    ctx = contextlib.nullcontext()
    if collate is not None and not args.no_priority_stream:
        # the search queue at high priority: the all-gather of step i runs in the slots step i+1's finished maps free instead of
        # displacing its workgroups (parallel.search_stream: 188 -> 171 us per step in a 1-rank RCCL group)
        hp = parallel.search_stream(dev)
        hp.wait_stream(torch.cuda.current_stream(dev))
        ctx = torch.cuda.stream(hp)
        collate_note += ", search launched on a high-priority stream"
    with ctx:
        # the contract's protocol to the letter first -- W warm-up + K timed steps straight after start-up, clocks as the idle GPU left
        # them -- reported as `contract_exact_no_prewarm`; then PREWARM_S of untimed launches and the same W + K again = the headline
        # (`config.prewarm_s`; ADVICE r3: label which is which)
        dt_cold, _ = timed_loop(run, args.steps, args.warmup, world, dev, collate)
        if not args.no_prewarm:
            prewarm(warm, dev)  # (on its OWN batches: the headline's batches stay unsearched)
        first_fresh = run._i
        dt, dev_ms = timed_loop(run, args.steps, args.warmup, world, dev, collate)
        all_first_visits = n_fresh > 0 and run._i <= len(run.sets)
        dt_nat = dt_hint = None
        run_h = None
        if not args.no_natural and run.packed is None:
            # the same W + K steps (a) in the natural order (workgroup i = map i) on the same kind of fresh batches, (b) "hinted": three
            # recurring batch sets, each searched longest-first by the order its searches FINISHED in at its previous visit
            run_n = Runner(sets[:per_pass] if n_fresh else sets, dev, placement="natural")
            dt_nat, _ = timed_loop(run_n, args.steps, args.warmup, world, dev, None)
            run_h = Runner([pool.fixed(k, b_rank) for k in range(N_ROTATE)], dev, placement="hinted")
            prewarm(run_h, dev, 0.1)
            dt_hint, _ = timed_loop(run_h, args.steps, args.warmup, world, dev, None)
    torch.cuda.synchronize(dev)
    _log(f"headline: {dt / args.steps * 1e3:.4f} ms/step")
    total_maps = n_gpus * b_rank * args.steps
    value = total_maps / dt

    if rank == 0:
        # parity sample + iteration statistics on the pool's seeded batch 0 (the generator's own starts), through the SAME dataset placement
        chk = Runner([pool.fixed(0, b_rank)], dev, placement="dataset")
        chk.step()
        torch.cuda.synchronize(dev)
        hist = chk.hist.cpu().numpy()
        paths = chk.paths.cpu().numpy()
        outs3 = run.sets[:N_ROTATE]  # the rotating output sets hold the latest three batches
        iters = torch.cat([z["iters"] for z in outs3]).cpu().numpy()
        assert all(int(z["status"].abs().sum().item()) == 0 for z in outs3) and int(chk.status.abs().sum().item()) == 0, "unsolvable map in the synthetic batch"
        avg_ms, med_ms, min_ms = kernel_launch_ms(run, min(args.steps, 100), dev)
        nat = hinted = None
        dt_pipe = None
        if run_h is not None and n_gpus == 1:
            try:
                dt_pipe = fresh_batches_pipelined(run_h, max(args.steps, 60), max(args.warmup, 9), dev)
            except Exception as e:  # noqa: BLE001 - an extra figure never sinks the line
                _log(f"fresh_batches_pipelined failed: {type(e).__name__}: {e}")
        if dt_nat is not None:
            nat_ms = kernel_launch_ms(run_n, min(args.steps, 100), dev)[0]
            nat = {"value": total_maps / dt_nat, "ms_per_step": dt_nat / args.steps * 1e3, "launch_ms_avg": nat_ms,
                   "roofline_frac": bytes_moved_per_map * b_rank / (nat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "roofline_frac_28B_per_cell": bytes_per_map * b_rank / (nat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "note": "same W + K steps on never-searched batches, workgroup i searches map i: a batch whose assembler passed no placement; identical outputs",
                   "pipelined_with_predictor": ({"value": b_rank * max(args.steps, 60) / dt_pipe, "ms_per_step": dt_pipe / max(args.steps, 60) * 1e3,
                                                 "note": "never-searched batches in a pipeline: a side stream computes the NEXT batch's placement from "
                                                         "its maps alone (nastar_placement_predict) while this batch is searched; nothing from an "
                                                         "earlier visit is used"} if dt_pipe else None)}
        if dt_hint is not None:
            hint_ms = kernel_launch_ms(run_h, min(args.steps, 100), dev)[0]
            hinted = {"value": total_maps / dt_hint, "ms_per_step": dt_hint / args.steps * 1e3, "launch_ms_avg": hint_ms,
                      "roofline_frac": bytes_moved_per_map * b_rank / (hint_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "RECURRING batches (three sets in rotation, each visited many times before the clock): every visit is searched longest-first by the "
                              "order its searches finished in at the previous visit (nastar_forward_ex order_out -> order; planner.Placement) -- what a "
                              "validation loop over a fixed set has from its second epoch on; this was the round-4 headline"}
        achieved = bytes_moved_per_map * b_rank / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            with open(tp) as f:
                tj = json.load(f)
            traffic = tj.get(args.workload, {}).get("bytes_per_launch")
        kprof = None  # the committed rocprofv3 average of the same kernel on the same command: `frac` is reproducible from one file
        kp = os.path.join(ROOT, "profiles", "kernel_profile.json")
        if os.path.exists(kp):
            with open(kp) as f:
                kprof = json.load(f).get(args.workload)
        placement_note = {
            "dataset": ("dataset: workgroup i searches map order[i], order = counting sort (longest first) of |opt_dist[start]| -- the optimal distance of each "
                        "sample's start cell, which the reference's maze files carry and its loader reads to draw that start (utils/data.py:127-134,"
                        "200-221) -- computed when the batch is ASSEMBLED (ops.order_from_levels: one small launch, outside the timed region like the "
                        "assembly itself); nothing any search has measured is used; identical outputs"),
            "hinted": "hinted: every recurring batch set is searched longest-first by the order its searches finished in at its previous visit",
            "natural": "natural: workgroup i searches map i"}[run.placement]
        out = {
            "metric": f"map-instances/s (forward A*) {Hh}x{Ww} Moore-8 @batch {b_rank * n_gpus if strong else B_PER_GPU}",
            "value": value, "unit": "maps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {b_rank} maps/GPU of {Hh}x{Ww} Moore-8, cost=map (VanillaAstar), "
                                   f"g_ratio={G_RATIO}, eval mode (search to goal), "
                                   + (f"{n_fresh} never-searched batches assembled loader-style from a resident pool of {pool.P} problems "
                                      f"(random maps + a random start per map from the optimal-distance bands, reference utils/data.py:200-221), "
                                      f"{n_fresh * 12 * Hh * Ww * b_rank / 1e6:.0f} MB of inputs per GPU, {N_ROTATE} rotating output sets, " if n_fresh else
                                      f"{N_ROTATE} recurring batch sets in rotation, ")
                                   + (f"global batch {args.global_batch} seeded 1234+1000k, {args.shard} shards"
                                      if strong else "pool seeds 1234+rank+1000k"),
                       "batch_per_gpu": b_rank, "global_batch": n_gpus * b_rank, "H": Hh, "W": Ww,
                       "parallelism": f"shard{n_gpus}" if n_gpus > 1 else "single", "collate": collate_note,
                       "prewarm_s": 0.0 if args.no_prewarm else PREWARM_S,
                       "prewarm_note": "untimed launches on batches of their OWN (never the headline's)",
                       "placement": placement_note,
                       "every_timed_step_is_a_first_visit": bool(all_first_visits),
                       "first_visit_note": (f"headline pass = batches #{first_fresh}..#{first_fresh + per_pass - 1} of {n_fresh} assembled; none was searched before its step"
                                            if all_first_visits else "batches recur (see workload)")},
            "natural_order": nat,
            "value_natural_order": nat["value"] if nat else None,  # (the same figure at top level: maps/s when the batch comes with no placement)
            "hinted": hinted,
            "value_hinted": hinted["value"] if hinted else None,  # (recurring batches placed by their previous visit: the round-4 headline)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "bytes_note": "achieved / frac on the bytes this call MUST move: 24 B/cell (VanillaAstar hands ONE tensor over as cost and passable "
                                       "map, reference astar.py:93-94; reads 3 x 4 B, writes fp32 histories + int64 paths); SURVEY 8(d)'s 28 B/cell "
                                       "(DifferentiableAstar boundary, four distinct inputs) beside it",
                         "algorithmic_bytes_per_launch": bytes_moved_per_map * b_rank,
                         "frac_28B_per_cell": bytes_per_map * b_rank / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": "committed constant: profiles/hbm_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of the same command, same kernel and batch shape; see its note), NOT measured in this run",
                         "kernel": "nastar_forward_compact_kernel (hand-scheduled step loop: nastar_search_asm4.hip.h)",
                         "launch_ms_avg": avg_ms, "launch_ms_median": med_ms, "launch_ms_min": min_ms,
                         "launch_ms_note": "HIP events around single launches on the launch stream, measured in THIS run, placement as the headline",
                         "committed_kernel_profile_us": kprof["avg_us"] if kprof else None,
                         "committed_kernel_profile_source": kprof["source"] if kprof else None,
                         "committed_frac_from_kernel_profile": (bytes_moved_per_map * b_rank / (kprof["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if kprof else None},
            # the search is a serial chain of select + update steps with all state on-chip, so next to the HBM fraction the line
            # carries what the launch costs the CU's pipes and what its longest chain alone costs (pipe_model above)
            "issue_model": (pipe_model(float(iters.sum()) / N_ROTATE, avg_ms * 1e3, int(iters.max()), LONE_STEP_NS, FIXED_US)
                            if (Hh, Ww) == (32, 32) and not strong else None),
            "contract_exact_no_prewarm": {"value": total_maps / dt_cold, "ms_per_step": dt_cold / args.steps * 1e3,
                                          "note": "the same W warm-up + K timed steps run FIRST (on never-searched batches of their own), without the untimed "
                                                  "pre-warm launches: the headline `value` is the second pass (clocks out of their idle state)"},
            "expansions_per_s": float(iters.sum()) / N_ROTATE * n_gpus * args.steps / dt,
            "mean_iters_per_map": float(iters.mean()), "max_iters_per_map": int(iters.max()),
            "device_ms_per_step": dev_ms / args.steps,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            _log("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(pr, hist, paths)
            _log("through_module")
            out["through_module"] = through_module_ms(pr, dev)
            _log("reference on this gpu")
            try:
                out["reference_torch_on_this_gpu"] = reference_on_this_gpu(pr, hist, paths, dev)
            except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline
                out["reference_torch_on_this_gpu"] = {"available": False, "note": f"{type(e).__name__}: {e}"}
        if n_gpus == 1 and not args.no_secondary:
            # secondary workloads on the same GPU (not the headline): short searches and the 64x64 shard of config 4
            sec = []
            for other in ("maze32", "rand32", "rand64"):
                if other == args.workload:
                    continue
                _log(f"secondary {other}")
                pr2 = make_problem(other, B_PER_GPU, seed=1234)
                run2 = Runner(pr2, dev)
                prewarm(run2, dev, 0.1)
                dt2, _ = timed_loop(run2, max(50, args.steps // 4), max(2, args.warmup // 4), 1, dev)
                a2, _, _ = kernel_launch_ms(run2, 50, dev)
                nbytes = 24 * run2.H * run2.W * B_PER_GPU  # cost == passable: the bytes that move
                a2n = None
                if run2.placement == "hinted" and not args.no_natural:
                    run2.placement = "natural"
                    a2n = kernel_launch_ms(run2, 50, dev)[0]
                    run2.placement = "hinted"
                sec.append({"workload": f"{other}: {B_PER_GPU} maps of {run2.H}x{run2.W}", "value": B_PER_GPU * max(50, args.steps // 4) / dt2,
                            "unit": "maps/s", "launch_ms_avg": a2, "hbm_frac": nbytes / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "hbm_frac_bytes_per_cell": 24,
                            "placement": run2.placement, "launch_ms_avg_natural_order": a2n,
                            "hbm_frac_natural_order": (nbytes / (a2n * 1e-3) / 1e9 / HBM_PEAK_GBS) if a2n else None,
                            "mean_iters_per_map": float(run2.iters.float().mean().item()),
                            "max_iters_per_map": int(run2.iters.max().item()),
                            "gpu_matches_oracle_on_sample": oracle_check(pr2, run2.hist.cpu().numpy(), run2.paths.cpu().numpy(), 256),
                            "oracle_sample": "first 256 maps"})
                del run2, pr2
            # the two other readings of BASELINE.json's "tau = 0.25" (SURVEY.md section 0.3): training-mode budget Tmax = 0.25
            # (searches truncated after 256 selections) and g_ratio = 0.8, on the headline maze batch
            for label, kw in ((("maze32, training-mode budget Tmax=0.25 (max 256 steps)", {"max_iters": int(0.25 * W * W)}),
                               ("maze32, g_ratio=0.8 (eval mode)", {"g_ratio": 0.8})) if args.workload == "maze32" and not strong else ()):
                run2 = Runner(pr, dev, **kw)
                dt2, _ = timed_loop(run2, max(10, args.steps // 4), max(2, args.warmup // 4), 1, dev)
                a2, _, _ = kernel_launch_ms(run2, max(10, min(args.steps // 4, 50)), dev)
                sec.append({"workload": f"{label}: {B_PER_GPU} maps of 32x32", "value": B_PER_GPU * max(10, args.steps // 4) / dt2,
                            "unit": "maps/s", "launch_ms_avg": a2, "hbm_frac": bytes_moved_per_map * B_PER_GPU / (a2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "hbm_frac_bytes_per_cell": 24, "mean_iters_per_map": float(run2.iters.float().mean().item()),
                            "max_iters_per_map": int(run2.iters.max().item())})
                del run2
            out["secondary"] = sec
        if n_gpus == 1 and not args.no_secondary and Hh == 32 and Ww == 32 and not strong:
            ex = {}
            for name, fn in (("neural_astar_cnn_hip_bf16", lambda: neural_astar_forward_ms(pr, dev)),
                             ("neural_astar_cnn_hip_f16x3", lambda: neural_astar_f16x3_ms(pr, dev)),
                             ("neural_astar_unet_hip_f16", lambda: neural_astar_unet_ms(pr, dev, "f16")),
                             ("neural_astar_unet_hip_f16x3", lambda: neural_astar_unet_ms(pr, dev, "f16x3")),
                             ("encoder_train_step", lambda: encoder_train_step_ms(pr, dev)),
                             ("train_fwd_bwd_ms_per_4096_maps_Tmax025", lambda: training_step_ms(pr, dev)),
                             ("data_path_32x32", lambda: data_path_ms(dev)),
                             ("train_l1_step_Tmax025", lambda: {"batch_100": l1_training_step_ms(pr, dev, 100),
                                                                "batch_4096": l1_training_step_ms(pr, dev, 4096)}),
                             # (extras are not bound to the K timed steps of the contract: the driver's K = 20 is 3 ms, too short for a pipeline to fill;
                             #  the multi-stream sweep moved to out["throughput_regime"], all workloads, both kernels)
                             ):
                _log(f"extra {name}")
                try:
                    ex[name] = fn()
                except Exception as e:  # noqa: BLE001 - an extra never sinks the headline line
                    ex[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            _log("batches in flight through the Python API (InFlightPlanner)")
            try:
                out["in_flight_through_api"] = in_flight_through_api(dev, args.steps)
                out["value_in_flight"] = out["in_flight_through_api"][0]["maps_per_s"]  # maze32, VanillaAstar, through the Python object
            except Exception as e:  # noqa: BLE001 - never sinks the headline line
                out["in_flight_through_api"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            _log("throughput regime (all workloads, general + unit-cost kernels)")
            try:
                out["throughput_regime"] = throughput_regime(dev, max(args.steps, 240))
            except Exception as e:  # noqa: BLE001 - never sinks the headline line
                out["throughput_regime"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            out["extra"] = {**ex,
                            "note": "encoder / training / data-path figures on the headline batch; none of them is the headline value"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            # strictly LAST and alone: 256 ATen threads on [B,32,32] maps starve the GPU launch thread of anything timed beside them
            _log("cpu baseline in BASELINE.md's exact configuration (hard limit 45 s)")
            out["cpu_baseline"]["spec_config"] = cpu_baseline_spec_collect(cpu_baseline_spec_start(pr, hist, paths), timeout_s=45.0)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_collate:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
