#!/usr/bin/env python3
"""bench.py -- forward DifferentiableAstar throughput on MI355X (the BASELINE.json metric), through the reference's own API.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is ONE call of the hot path the way a user of the reference makes it -- ``VanillaAstar.forward(map_designs, start_maps, goal_maps)``
(reference planner/astar.py:73-102) with the default same-call solvability verdict -- over one batch of B = 4096 synthetic 32x32 Moore-8 maps
that already sit in HBM and have NEVER been searched.  The batch comes from a loader that knows ``opt_dists`` (every sample of the reference's
maze files carries it, utils/data.py:127-134) and attaches the placement hint a ``DeviceMazeBatches`` batch carries: that counting-sort launch
runs INSIDE the timed step.  ``value`` is therefore what a caller of the reference's signature gets; the bare C-ABI launch
on preallocated outputs -- rounds 1-5's headline -- is reported beside it (``bare_launch``), and so is the same call without a hint
(``natural_order``).  With N > 1 (launched by ``python -m torch.distributed.run --nproc-per-node N``) every rank owns its own 4096 maps
(weak scaling) and the bit-packed ``AstarOutput`` of all ranks is collated with ONE RCCL all-gather per 16 steps (``--collate-bucket``; emitted by
the search launch itself, the tail bucket flushed inside the timed region), overlapped with the following steps' searches.  Rank 0 prints ONE JSON line.

The line's core (headline, roofline, cpu_baseline) is computed first; every secondary experiment (other workloads, batches in flight, encoders,
training steps, the reference run on this GPU) lives in ``bench_extras.py`` and runs in a CHILD process with a time limit: whatever happens
there, the one line that matters is printed.

Workload (``config.workload``): ``maze32`` = seeded maze-like stand-in for the absent ``mazes_032_moore_c8.npz``
(SURVEY.md section 8d-ii), VanillaAstar convention cost = map, g_ratio 0.5, eval mode (search runs to the goal).
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


FLAG_UNIT_COST = 64  # include/nastar.h NASTAR_FLAG_UNIT_COST
LONE_STEP_NS = 219.0  # step of a wavefront that has its SIMD to itself, chip at working clocks: slope of the launch time over the length of ONE maze
                      # search among 4095 two-step maps (tools/probe_latency.py, profiles/r04/lat_working_clocks.txt; on an otherwise IDLE chip: 280)
FIXED_US = 8.0  # map load + backtrack + output stores of that wavefront


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




class ApiRunner:
    """step() = what a user of the reference does per batch: ``planner(map_designs, start_maps, goal_maps)`` on the reference's [B,1,H,W]
    tensors, default (same-call) solvability verdict, outputs allocated by the call.  ``placement``:
      "dataset"  the batch carries its loader's hint -- ``ops.attach_levels(start_maps, levels)``: the start cells' optimal distances (data every
                 sample of the reference's maze files holds, utils/data.py:127-134, 200-221; what ``utils.data.DeviceMazeBatches`` sorts at
                 batch assembly) -- and ``forward()`` turns them into a placement with ONE counting-sort launch in front of its search launch:
                 INSIDE the step, i.e. inside the timed region;
      "natural"  no hint: workgroup i searches map i.
    Sets: dicts {"m", "s", "g" [B,H,W], "levels" [B] int32} (FreshBatches.batch / .fixed)."""

    def __init__(self, sets, dev, placement="dataset"):
        from neural_astar import ops
        from neural_astar.planner import VanillaAstar
        self.ops = ops
        self.dev = dev
        self.va = VanillaAstar(g_ratio=G_RATIO).to(dev).eval()
        self.sets = [dict(m=z["m"].unsqueeze(1), s=z["s"].unsqueeze(1), g=z["g"].unsqueeze(1), levels=z["levels"]) for z in sets]
        self.B, _, self.H, self.W = self.sets[0]["m"].shape
        self.placement = placement
        self.out = None
        self.packed = None
        self.collator = None
        self._i = 0

    def step(self):
        z = self.sets[self._i % len(self.sets)]
        self._i += 1
        if self.placement == "dataset":
            self.ops.attach_levels(z["s"], z["levels"])  # what the loader knows; forward() sorts it into a placement right in front of its search launch
        if self.collator is not None:  # N > 1: the search launch writes its bit-packed masks straight into the collation bucket's next slot
            self.va.astar.packed_sink = self.collator.next_slot(self.B, self.H, self.W, self.dev)
        self.out = self.va(z["m"], z["s"], z["g"])

    @property
    def iters(self):
        return self.va.astar.last_iters

    @property
    def status(self):
        return self.va.astar.last_status


def run_extras(args, timeout_s):
    """every secondary experiment in a CHILD process (bench_extras.py) with a time limit; -> dict merged into the line (or a note why not)"""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench_extras.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_natural:
        cmd.append("--no-natural")
    if args.no_cpu_baseline:
        cmd.append("--no-reference")
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=None, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired as e:
        part = (e.stdout or b"")
        part = part.decode() if isinstance(part, bytes) else part
        lines = [ln for ln in part.splitlines() if ln.startswith("{")]
        got = {}
        for ln in lines:  # (the child prints one JSON object per finished section: keep what it finished)
            try:
                got.update(json.loads(ln))
            except ValueError:
                pass
        got["extras_note"] = f"bench_extras.py stopped at its {timeout_s:.0f} s limit; sections finished until then are above"
        return got
    got = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            try:
                got.update(json.loads(ln))
            except ValueError:
                pass
    got["extras_note"] = f"bench_extras.py (child process), rc {r.returncode}, {time.perf_counter() - t0:.0f} s"
    return got


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
    ap.add_argument("--placement", default="dataset", choices=["dataset", "natural"],
                    help="dataset (default): every timed step searches a NEVER-SEARCHED batch that carries its loader's placement hint (counting sort of "
                         "the start cells' optimal distances, computed INSIDE the timed step); natural: no hint, workgroup i = map i")
    ap.add_argument("--no-natural", action="store_true", help="skip the comparison passes (natural order, bare launches): clean kernel profiles")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the untimed clock pre-warm launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip bench_extras.py (the secondary experiments)")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="time limit of the bench_extras.py child process, seconds")
    ap.add_argument("--no-collate", action="store_true", help="N>1: skip the all-gather of AstarOutput")
    ap.add_argument("--collate-bucket", type=int, default=16, help="N>1: steps per all-gather (parallel.BucketedCollator)")
    ap.add_argument("--force-collate", action="store_true", help="dev: run the N>1 collation path (pack kernel + all-gather) in a 1-rank RCCL group")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = the headline (BASELINE config 2 / 4); train = one full NeuralAstar training step per bench step "
                         "(BASELINE config 5 with --config warcraft; bench_extras.train_main)")
    ap.add_argument("--config", default="warcraft", help="--mode train: which reference training configuration (bench_extras.TRAIN_CONFIGS)")
    ap.add_argument("--batch-per-gpu", type=int, default=100, help="--mode train: maps per rank and step (the reference's batch_size is 100)")
    ap.add_argument("--encoder-backend", default="hip_f16x3", choices=["auto", "hip_f16x3", "hip_f16", "torch"],
                    help="--mode train (auto = the package default = hip_f16x3 on a HIP device)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="dev: gloo lets N ranks share ONE GPU (with --share-gpu) to exercise every world > 1 branch of this script on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="dev: every rank uses cuda:0 (only with --dist-backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: re-launch one rank per GPU the way the driver does
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env, stdout=real_stdout))
    if args.mode == "train":
        assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback in the product path)"
        import bench_extras
        bench_extras.train_main(args, real_stdout)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    dist_info = {"initialized": False}
    if world > 1 or args.force_collate:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # what a SCALE_rNN.json can be cross-checked against: the backend that really carried the collectives and its size
        dist_info = {"initialized": True, "backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_device": str(torch.device("cuda", local_rank)),
                     "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if args.dist_backend == "nccl" else None),
                     "gpus_visible": torch.cuda.device_count()}
    n_gpus = world if world > 1 else 1
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback in the product path)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)  # an inference benchmark: forward() of the reference's eval scripts runs under no_grad (training: --mode train)
    from neural_astar import _native
    fast_lane = _native.load_fastlane() is not None

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
    if not strong:
        # every step of the cold pass AND of the headline pass searches a batch that has never been searched: 2 (W + K) of them,
        # capped at MAX_FRESH_SETS -- beyond the cap the loop revisits batches (same hint, nothing learned)
        n_fresh = min(2 * per_pass + 2, MAX_FRESH_SETS)
        sets = [pool.batch(b_rank) for _ in range(n_fresh)]
    else:
        sets = [pool.fixed(k, b_rank) for k in range(N_ROTATE)]
        n_fresh = 0
    torch.cuda.synchronize(dev)
    run = ApiRunner(sets, dev, placement=args.placement)
    warm = Runner([pool.batch(b_rank) for _ in range(N_ROTATE)] if not strong else sets, dev, placement="dataset")
    Hh, Ww = run.H, run.W
    bytes_per_map = 28 * Hh * Ww  # four input tensors
    bytes_moved_per_map = 24 * Hh * Ww  # ... of which VanillaAstar's call passes ONE as cost and passable (reference astar.py:93-94): what really moves

    collate = None
    collate_note = "n/a (single GPU)"
    if (world > 1 or args.force_collate) and not args.no_collate:
        from neural_astar import parallel

        # every step's outputs are bit-packed (2 bits per cell) into a slot of a staging buffer on a side stream, and the ranks exchange ONE
        # all-gather per `--collate-bucket` steps (parallel.BucketedCollator: fewer, larger collectives -- a collective per 0.14 ms step costs
        # more host and launch time than the search itself); the partly filled last bucket is flushed INSIDE the timed region
        collator = parallel.BucketedCollator(bucket=max(1, args.collate_bucket), keep="last")

        if isinstance(run, ApiRunner):
            run.collator = collator

        def collate(pending):
            collator.add(run.out, packed=run.va.astar.last_packed if isinstance(run, ApiRunner) else None)
            return collator.flush  # timed_loop calls the last one after the K steps: the tail bucket + the wait for everything in flight
        try:
            run.step()
            collate(None)()
            torch.cuda.synchronize(dev)
            collate_note = (f"AstarOutput of every rank as 2 bits per cell (emitted by the search launch itself into the bucket's slot), ONE all-gather per "
                            f"{collator.bucket} steps over {'RCCL' if args.dist_backend == 'nccl' else args.dist_backend} (kept packed; the tail "
                            f"bucket is flushed inside the timed region), overlapped with the following steps' searches")
        except Exception as e:  # reported, not hidden: the line then says the collective was not part of the step
            collate = None
            collate_note = f"FAILED ({type(e).__name__}: {e}); steps timed without the all-gather"

    _log("problems resident; cold pass, clock pre-warm, headline pass")
    # the contract's protocol to the letter first -- W warm-up + K timed steps straight after start-up, clocks as the idle GPU left
    # them -- reported as `contract_exact_no_prewarm`; then PREWARM_S of untimed launches and the same W + K again = the headline
    dt_cold, _ = timed_loop(run, args.steps, args.warmup, world, dev, collate)
    if not args.no_prewarm:
        prewarm(warm, dev)  # (bare launches on batches of their OWN: the headline's batches stay unsearched)
    first_fresh = run._i
    dt, dev_ms = timed_loop(run, args.steps, args.warmup, world, dev, collate)
    all_first_visits = n_fresh > 0 and run._i <= len(run.sets)
    iters = run.iters.cpu().numpy()
    assert int(run.status.abs().sum().item()) == 0, "unsolvable map in the synthetic batch"
    torch.cuda.synchronize(dev)
    _log(f"headline: {dt / args.steps * 1e3:.4f} ms/step")
    total_maps = n_gpus * b_rank * args.steps
    value = total_maps / dt
    comp = {}
    if not args.no_natural and collate is None:
        # beside the headline, the same W + K steps: (a) through the API without a hint, (b) BARE C-ABI launches on preallocated outputs with the
        # order computed at batch assembly -- outside the timed region: the headline of rounds 1-5 --, (c) bare launches in the natural order
        fresh2 = sets[:per_pass] if n_fresh else sets
        for key, mk in (("api_natural_order", lambda: ApiRunner(fresh2, dev, placement="natural")),
                        ("bare_dataset_order", lambda: Runner(fresh2, dev, placement="dataset")),
                        ("bare_natural_order", lambda: Runner(fresh2, dev, placement="natural"))):
            r2 = mk()
            d2, _ = timed_loop(r2, args.steps, args.warmup, world, dev, None)
            comp[key] = {"value": total_maps / d2, "ms_per_step": d2 / args.steps * 1e3}
            del r2

    if rank == 0:
        # parity sample on the pool's seeded batch 0 (the generator's own starts), through the SAME API call
        chk = ApiRunner([pool.fixed(0, b_rank)], dev, placement=args.placement)
        chk.step()
        torch.cuda.synchronize(dev)
        hist = chk.out.histories[:, 0].cpu().numpy()
        paths = chk.out.paths[:, 0].cpu().numpy()
        # the dominant kernel's duration: HIP events around single launches on the launch stream, same placement as the headline
        krun = Runner(sets[:per_pass] if n_fresh else sets, dev, placement=args.placement)
        avg_ms, med_ms, min_ms = kernel_launch_ms(krun, min(args.steps, 100), dev)
        nat_ms = None
        if not args.no_natural:  # (--no-natural: a kernel profile of this run must hold ONE placement)
            krun_n = Runner(sets[:per_pass] if n_fresh else sets, dev, placement="natural")
            nat_ms = kernel_launch_ms(krun_n, min(args.steps, 100), dev)[0]
            del krun_n
        del krun
        achieved = bytes_moved_per_map * b_rank / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            with open(tp) as f:
                traffic = json.load(f).get(args.workload, {}).get("bytes_per_launch")
        kprof = None  # the committed rocprofv3 average of the same kernel on the same command: `frac` is reproducible from one file
        kp = os.path.join(ROOT, "profiles", "kernel_profile.json")
        if os.path.exists(kp):
            with open(kp) as f:
                kprof = json.load(f).get(args.workload)
        placement_note = {
            "dataset": ("dataset: the batch carries its loader's hint (start_maps.placement_order) -- workgroup i searches map order[i], order = counting sort "
                        "(longest first) of |opt_dist[start]|, the optimal distance of each sample's start cell, which the reference's maze files carry and "
                        "its loader reads to draw that start (utils/data.py:127-134,200-221).  The counting sort (one small launch, issued by forward() in front of its search launch) runs INSIDE "
                        "every timed step; nothing any search has measured is used; identical outputs"),
            "natural": "natural: no hint, workgroup i searches map i"}[run.placement]
        out = {
            "metric": f"map-instances/s (forward A*) {Hh}x{Ww} Moore-8 @batch {b_rank * n_gpus if strong else B_PER_GPU}",
            "value": value, "unit": "maps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {b_rank} maps/GPU of {Hh}x{Ww} Moore-8, cost=map (VanillaAstar), "
                                   f"g_ratio={G_RATIO}, eval mode (search to goal), "
                                   + (f"{n_fresh} never-searched batches assembled loader-style from a resident pool of {pool.P} problems "
                                      f"(random maps + a random start per map from the optimal-distance bands, reference utils/data.py:200-221), "
                                      f"{n_fresh * 12 * Hh * Ww * b_rank / 1e6:.0f} MB of inputs per GPU, " if n_fresh else
                                      f"{N_ROTATE} recurring batch sets in rotation, ")
                                   + (f"global batch {args.global_batch} seeded 1234+1000k, {args.shard} shards"
                                      if strong else "pool seeds 1234+rank+1000k"),
                       "step": "VanillaAstar.forward(map_designs, start_maps, goal_maps) on [B,1,H,W] tensors: outputs allocated by the call, default "
                               "check_solvable (the unsolvable-map verdict in the SAME call), "
                               + ("the loader's placement hint computed inside the step" if run.placement == "dataset" else "no placement hint"),
                       "host_lane": "native (lib/_nastar_fastlane.so: allocation + launch + completion-flag poll in C++)" if fast_lane else "python (ctypes)",
                       "batch_per_gpu": b_rank, "global_batch": n_gpus * b_rank, "H": Hh, "W": Ww,
                       "parallelism": f"shard{n_gpus}" if n_gpus > 1 else "single", "collate": collate_note, "distributed": dist_info,
                       "prewarm_s": 0.0 if args.no_prewarm else PREWARM_S,
                       "prewarm_note": "untimed bare launches on batches of their OWN (never the headline's)",
                       "placement": placement_note,
                       "every_timed_step_is_a_first_visit": bool(all_first_visits),
                       "first_visit_note": (f"headline pass = batches #{first_fresh}..#{first_fresh + per_pass - 1} of {n_fresh} assembled; none was searched before its step"
                                            if all_first_visits else "batches recur (see workload)")},
            "value_is": "maps/s a caller of the reference's forward() signature gets (VERDICT r5 item 3): placement sort + allocation + launch + same-call verdict per step",
            "natural_order": comp.get("api_natural_order"),
            "value_natural_order": comp.get("api_natural_order", {}).get("value"),
            "bare_launch": ({"dataset_order": comp.get("bare_dataset_order"), "natural_order": comp.get("bare_natural_order"),
                             "note": "nastar_forward_ex back to back through ctypes on preallocated outputs, no verdict read, the order computed at batch "
                                     "assembly OUTSIDE the timed region: the headline of rounds 1-5, what no host shim can beat"} if comp else None),
            "value_bare_launch": comp.get("bare_dataset_order", {}).get("value"),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "bytes_note": "achieved / frac on the bytes this call MUST move: 24 B/cell (VanillaAstar hands ONE tensor over as cost and passable "
                                       "map, reference astar.py:93-94; reads 3 x 4 B, writes fp32 histories + int64 paths); SURVEY 8(d)'s 28 B/cell "
                                       "(DifferentiableAstar boundary, four distinct inputs) beside it",
                         "algorithmic_bytes_per_launch": bytes_moved_per_map * b_rank,
                         "frac_28B_per_cell": bytes_per_map * b_rank / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_of_whole_step": bytes_moved_per_map * b_rank / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": "committed constant: profiles/hbm_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of the same command, same kernel and batch shape; see its note), NOT measured in this run",
                         "kernel": "nastar_forward_compact_kernel (hand-scheduled step loop: nastar_search_asm4.hip.h)",
                         "launch_ms_avg": avg_ms, "launch_ms_median": med_ms, "launch_ms_min": min_ms,
                         "launch_ms_avg_natural_order": nat_ms,
                         "frac_natural_order": (bytes_moved_per_map * b_rank / (nat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if nat_ms else None,
                         "launch_ms_note": "HIP events around single bare launches (C ABI, preallocated outputs) on the launch stream, measured in THIS run on the headline pass's batches, placement as the headline",
                         "committed_kernel_profile_us": kprof["avg_us"] if kprof else None,
                         "committed_kernel_profile_source": kprof["source"] if kprof else None,
                         "committed_frac_from_kernel_profile": (bytes_moved_per_map * b_rank / (kprof["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if kprof else None},
            # the search is a serial chain of select + update steps with all state on-chip, so next to the HBM fraction the line
            # carries what the launch costs the CU's pipes and what its longest chain alone costs (pipe_model above)
            "issue_model": (pipe_model(float(iters.sum()), avg_ms * 1e3, int(iters.max()), LONE_STEP_NS, FIXED_US)
                            if (Hh, Ww) == (32, 32) and not strong else None),
            "contract_exact_no_prewarm": {"value": total_maps / dt_cold, "ms_per_step": dt_cold / args.steps * 1e3,
                                          "note": "the same W warm-up + K timed steps run FIRST (on never-searched batches of their own), without the untimed "
                                                  "pre-warm launches: the headline `value` is the second pass (clocks out of their idle state)"},
            "expansions_per_s": float(iters.sum()) * n_gpus * args.steps / dt,
            "mean_iters_per_map": float(iters.mean()), "max_iters_per_map": int(iters.max()),
            "device_ms_per_step": dev_ms / args.steps,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            _log("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(pr, hist, paths)
        else:
            out["gpu_matches_oracle_on_sample"] = oracle_check(pr, hist, paths, 256)
        if n_gpus == 1 and not args.no_secondary:
            _log("secondary experiments: bench_extras.py (child process)")
            del run, warm, chk, sets
            torch.cuda.empty_cache()
            try:
                out.update(run_extras(args, args.extras_timeout))
            except Exception as e:  # noqa: BLE001 - never sinks the line
                out["extras_note"] = f"bench_extras.py failed to run: {type(e).__name__}: {str(e)[:300]}"
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
