"""Multi-GPU sharding of the planner: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

Maps are independent, so the batch is partitioned into contiguous row blocks, one per rank,
with NO exchange during the search.  The only collective is ONE all-gather that collates ``AstarOutput`` -- and it
moves the information content, not the reference's fat tensors: ``histories`` and ``paths`` are exact 0/1 masks, so
each rank contributes 2 bits per cell (bit-packed uint8) instead of 12 bytes per cell (fp32 + int64): 4096 maps of
32x32 are 1 MiB per rank instead of 48 MiB, which over 7 point-to-point xGMI links is latency-, not bandwidth-bound.
The fp32/int64 views are rebuilt locally on the ranks that want them.

A second, scalar collective (all-reduce MAX of the step count) is optional and only used in training so that the
batch-coupled gradient terms match a single-device run of the full batch.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops
from .planner.differentiable_astar import AstarOutput, UnsolvableMapError, _raise_unsolvable

_BIT_WEIGHTS = None
_SEARCH_STREAMS: dict = {}


def search_stream(device: torch.device) -> "torch.cuda.Stream":
    """A HIGH-priority HIP stream (one per device, cached) to launch the search on when a collective of an earlier batch is in
    flight at the same time.

    A 4096-map launch of 32x32 maps fills every LDS byte of every CU (16 maps per CU), so a concurrent kernel -- RCCL's all-gather of
    the previous batch's masks -- that is dispatched first pushes search workgroups into a second round and the launch waits for a
    late chain.  With the search queue at high priority its workgroups are dispatched first and the collective runs in the slots
    the finished maps free.  Measured on one MI355X, 1-rank RCCL group, us per step (`tools/probe_collate.py`,
    `profiles/r03/probe_collate.txt`): search alone 160.8, + overlapped all-gather 188.2 on the default stream, 171.0 on this one."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SEARCH_STREAMS:
        _SEARCH_STREAMS[key] = torch.cuda.Stream(device, priority=-1)
    return _SEARCH_STREAMS[key]


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block [lo, hi) of rank ``rank``; the first ``n % world_size`` ranks get one extra row."""
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_rows(n: int, world_size: int, rank: int, mode: str = "contiguous") -> torch.Tensor:
    """Row indices (ascending, int64) of the global batch that rank ``rank`` plans.

    ``contiguous``: the block of :func:`shard_bounds`.  ``interleaved``: rows ``b`` with ``b % world_size == rank`` -- searches
    differ 10-100x in length (SURVEY.md section 8e caveat 2) and batches are often ordered (by dataset file, by difficulty), so
    striding the rows spreads long searches over the ranks instead of handing one rank a hard block; it costs nothing, the
    collated order is restored by :func:`collated_order`."""
    if mode == "contiguous":
        lo, hi = shard_bounds(n, world_size, rank)
        return torch.arange(lo, hi, dtype=torch.int64)
    if mode == "interleaved":
        return torch.arange(rank, n, world_size, dtype=torch.int64)
    raise ValueError(f"unknown sharding mode {mode!r}")


def collated_order(n: int, world_size: int, mode: str = "contiguous") -> torch.Tensor:
    """Permutation ``perm`` such that ``gathered[perm]`` is the global batch in its original row order, where ``gathered`` is
    the rank-major concatenation an all-gather of the per-rank shards produces.  Requires equal shard sizes."""
    rows = torch.cat([shard_rows(n, world_size, r, mode) for r in range(world_size)])
    perm = torch.empty_like(rows)
    perm[rows] = torch.arange(rows.numel(), dtype=torch.int64)
    return perm


def pack_masks(histories: torch.Tensor, paths: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,1,H,W] fp32 0/1 + [B,1,H,W] int64 0/1 -> [B, 2*ceil(HW/8)] uint8 (histories bits, then path bits); ``out``: a contiguous
    uint8 tensor of that shape to pack into (a slot of a collation bucket) instead of a fresh one.

    Device tensors go through ``nastar_pack_outputs`` (one HIP kernel on the current stream); host tensors (the gloo
    tests) use the equivalent torch expression below."""
    B = histories.shape[0]
    hw = histories[0].numel()
    nb = (hw + 7) // 8
    if histories.is_cuda:
        from . import _native
        lib = _native.load()
        H, W = histories.shape[-2:]
        h = histories.contiguous()
        p = paths.contiguous()
        if out is None:
            out = torch.empty((B, 2 * nb), dtype=torch.uint8, device=h.device)
        elif out.shape != (B, 2 * nb) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != h.device:
            raise ValueError(f"pack_masks(out=...): expected a contiguous uint8 tensor of shape {(B, 2 * nb)} on {h.device}")
        with torch.cuda.device(h.device):
            rc = lib.nastar_pack_outputs(h.data_ptr(), p.data_ptr(), B, H, W, out.data_ptr(),
                                         torch.cuda.current_stream(h.device).cuda_stream)
        _native.check(rc, "nastar_pack_outputs")
        return out
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8, device=histories.device)

    def pk(x: torch.Tensor) -> torch.Tensor:
        bits = (x.reshape(B, hw) != 0).to(torch.uint8)
        if nb * 8 != hw:
            bits = torch.nn.functional.pad(bits, (0, nb * 8 - hw))
        return (bits.reshape(B, nb, 8) * w).sum(-1, dtype=torch.uint8)

    res = torch.cat((pk(histories), pk(paths)), dim=1).contiguous()
    if out is not None:
        out.copy_(res)
        return out
    return res


def unpack_masks(packed: torch.Tensor, H: int, W: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of :func:`pack_masks`: -> histories [B,1,H,W] fp32, paths [B,1,H,W] int64."""
    B = packed.shape[0]
    hw = H * W
    nb = (hw + 7) // 8
    if packed.is_cuda:
        from . import _native
        lib = _native.load()
        pk = packed.contiguous()
        hist = torch.empty((B, 1, H, W), dtype=torch.float32, device=pk.device)
        paths = torch.empty((B, 1, H, W), dtype=torch.int64, device=pk.device)
        with torch.cuda.device(pk.device):
            rc = lib.nastar_unpack_outputs(pk.data_ptr(), B, H, W, hist.data_ptr(), paths.data_ptr(),
                                           torch.cuda.current_stream(pk.device).cuda_stream)
        _native.check(rc, "nastar_unpack_outputs")
        return hist, paths
    shifts = torch.tensor([7, 6, 5, 4, 3, 2, 1, 0], dtype=torch.uint8, device=packed.device)

    def un(x: torch.Tensor) -> torch.Tensor:
        bits = (x.unsqueeze(-1) >> shifts) & 1
        return bits.reshape(B, nb * 8)[:, :hw].reshape(B, 1, H, W)

    return un(packed[:, :nb]).to(torch.float32), un(packed[:, nb:]).to(torch.int64)


def _equal_shards(n_local: int, group: Optional[dist.ProcessGroup], device: torch.device) -> None:
    """all_gather_into_tensor needs the same number of rows on every rank (a ragged last DataLoader batch or n % world != 0
    would otherwise end in an RCCL error or a hang): check it with one tiny all-reduce and fail with a clear message."""
    t = torch.tensor([n_local, -n_local], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t[0]) != -int(t[1]):
        raise ValueError(f"sharded collation needs equally sized shards on every rank: this rank has {n_local} rows, the "
                         f"group has between {-int(t[1])} and {int(t[0])}; pad the batch or drop the ragged tail")


def all_gather_output(out: AstarOutput, group: Optional[dist.ProcessGroup] = None,
                      async_op: bool = False, unpack: bool = True, order: Optional[torch.Tensor] = None,
                      check_sizes: Optional[bool] = None):
    """Collate the per-rank ``AstarOutput`` of equally sized shards with ONE all-gather (RCCL on GPUs).

    Returns ``AstarOutput`` of the full batch (rank-major row order, or the original order when ``order`` =
    :func:`collated_order` is given), or, with ``async_op=True``, a tuple ``(work, finish)`` where ``finish()`` waits and
    returns the collated output -- lets the caller overlap the collective with the next batch's search on the compute stream.
    ``unpack=False`` keeps the collated batch in its bit-packed form (``[world*B, 2*ceil(HW/8)] uint8``;
    :func:`unpack_masks` materialises the reference's fp32 / int64 tensors on demand): expanding 2 bits per cell to 12 bytes
    per cell for the whole global batch on every rank is the expensive part of the collation and few consumers need it.
    The collated output carries no autograd graph (masks only) and, like the differentiable path of the reference, an empty
    ``intermediate_results`` list."""
    H, W = out.histories.shape[-2:]
    packed = pack_masks(out.histories.detach(), out.paths)
    # the shard-size check is an extra all-reduce + a host sync: on by default for the blocking form only -- the async form exists to
    # overlap the collective with the next batch's search and must not block on the host (pass check_sizes=True to force it; the
    # decision must not depend on per-rank state, or a rank that skips the check would leave the others waiting in its all-reduce)
    if check_sizes or (check_sizes is None and not async_op):
        _equal_shards(packed.shape[0], group, packed.device)
    world = dist.get_world_size(group)
    gathered = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=torch.uint8, device=packed.device)
    work = dist.all_gather_into_tensor(gathered, packed, group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        g = gathered if order is None else gathered[order.to(gathered.device)]
        if not unpack:
            return g
        h, p = unpack_masks(g, H, W)
        return AstarOutput(h, p, [])

    if async_op:
        return work, finish
    return finish()


def all_gather_packed(packed: torch.Tensor, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
    """All-gather an already bit-packed shard (``nastar_forward_packed`` emits it in the search launch itself).
    Returns the collated ``[world*B, 2*ceil(HW/8)] uint8`` tensor, or ``(work, finish)`` with ``async_op=True``."""
    world = dist.get_world_size(group)
    gathered = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=torch.uint8, device=packed.device)
    work = dist.all_gather_into_tensor(gathered, packed, group=group, async_op=async_op)

    def finish() -> torch.Tensor:
        if work is not None:
            work.wait()
        return gathered

    if async_op:
        return work, finish
    return finish()


class BucketedCollator:
    """Collation of a SEQUENCE of sharded steps: every rank's ``AstarOutput`` of step i is bit-packed into slot i % bucket of a
    staging buffer (2 bits per cell) and the ranks exchange ONE all-gather per ``bucket`` steps -- fewer, larger collectives: per step
    a collective costs tens of microseconds of host and launch time whatever it moves (the search of a 4096-map batch is 0.12-0.14 ms),
    and on xGMI a ring all-gather of 1 MiB per rank is latency-, not link-bound.  Packing and collective run on a SIDE stream behind
    the step's outputs; two staging buffers alternate, so steps keep packing while the previous bucket is in flight.

        col = BucketedCollator(bucket=8)
        for batch in shard_loader:
            col.add(planner(*batch))          # returns at once
        gathered = col.flush()                # list of [world, steps, B, 2*ceil(HW/8)] uint8 tensors, one per bucket, in order

    Without a pack launch at all: ``planner.astar.packed_sink = col.next_slot(B, H, W, device)`` before the call and
    ``col.add(out, packed=planner.astar.last_packed)`` after it -- the search launch then writes the slot itself.

    ``keep="all"`` keeps every collated bucket until ``flush()``; ``keep="last"`` only the latest (a benchmark, a consumer that
    reduces each bucket as it completes: ``on_bucket(tensor, n_steps)`` is called, on the host, when a bucket's all-gather has been
    waited for).  Shards must be equally sized on every rank (``all_gather_into_tensor``); host tensors (gloo tests) work."""

    def __init__(self, bucket: int = 8, group: Optional[dist.ProcessGroup] = None, keep: str = "all", on_bucket=None):
        if bucket < 1:
            raise ValueError("bucket must be >= 1")
        if keep not in ("all", "last"):
            raise ValueError('keep must be "all" or "last"')
        self.bucket = int(bucket)
        self.group = group
        self.keep = keep
        self.on_bucket = on_bucket
        self._stage: List[Optional[torch.Tensor]] = [None, None]
        self._cur = 0            # staging buffer being filled
        self._n = 0              # steps packed into it
        self._pending = None     # (work, gathered, n_steps, buffer index) of the bucket in flight
        self._done: List[torch.Tensor] = []
        self._stream = None
        self._event = None
        self.collectives = 0
        self._sync_main = False
        self._slots: List[list] = [[], []]
        self._given = None

    def _side(self, device: torch.device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device)
            self._event = torch.cuda.Event()
        return self._stream

    def _wait_pending(self) -> None:
        if self._pending is None:
            return
        work, gathered, n, idx = self._pending
        self._pending = None
        if work is not None:
            work.wait()
        if self.on_bucket is not None:
            self.on_bucket(gathered, n)
        if self.keep == "all":
            self._done.append(gathered)
        else:
            self._done = [gathered]

    def _launch(self) -> None:
        """all-gather of the staging buffer being filled (its first ``_n`` slots); the other buffer takes the next steps"""
        self._wait_pending()  # (one bucket in flight: its staging buffer is the one we switch to)
        idx, n = self._cur, self._n
        stage = self._stage[idx][:n]
        world = dist.get_world_size(self.group)
        gathered = torch.empty((world,) + tuple(stage.shape), dtype=torch.uint8, device=stage.device)
        side = self._side(stage.device)
        if side is not None:
            # once per bucket the side stream is ordered behind the caller's stream: slots written by the steps' own launches (next_slot)
            # are complete there, and `gathered` -- allocated just now for the CALLER's stream -- may be a block that kernels still
            # pending on that stream use
            self._event.record(torch.cuda.current_stream(stage.device))
            side.wait_event(self._event)
            self._sync_main = False
            with torch.cuda.stream(side):  # the collective is ordered behind the packs of this bucket, which ran on the side stream
                work = dist.all_gather_into_tensor(gathered.view(world * n * stage.shape[1], stage.shape[2]),
                                                   stage.reshape(n * stage.shape[1], stage.shape[2]), group=self.group, async_op=True)
        else:
            work = dist.all_gather_into_tensor(gathered.view(world * n * stage.shape[1], stage.shape[2]),
                                               stage.reshape(n * stage.shape[1], stage.shape[2]), group=self.group, async_op=True)
        self.collectives += 1
        self._pending = (work, gathered, n, idx)
        self._cur, self._n = 1 - idx, 0

    def _staging(self, B: int, nb2: int, device: torch.device) -> torch.Tensor:
        st = self._stage[self._cur]
        if st is None or st.shape[1] != B or st.shape[2] != nb2 or st.device != device:
            if st is not None and (self._n or self._pending is not None):
                raise ValueError("BucketedCollator: every step of a run must have the same shard shape (flush() between runs)")
            self._stage = [torch.empty((self.bucket, B, nb2), dtype=torch.uint8, device=device) for _ in range(2)]
            self._slots = [[b[i] for i in range(self.bucket)] for b in self._stage]  # (views made once: a step costs no tensor construction)
            st = self._stage[self._cur]
        return st

    def next_slot(self, B: int, H: int, W: int, device: torch.device) -> torch.Tensor:
        """The staging slot the NEXT ``add`` fills ([B, 2*ceil(HW/8)] uint8): hand it to the planner (``planner.astar.packed_sink``) and the
        search launch emits the packed masks itself -- ``add(out, packed=planner.astar.last_packed)`` then has nothing left to launch."""
        self._staging(B, 2 * ((H * W + 7) // 8), device)
        self._given = self._slots[self._cur][self._n]
        return self._given

    def add(self, out: AstarOutput, packed: Optional[torch.Tensor] = None) -> None:
        if packed is not None and packed is self._given:
            # the step's own launch wrote the slot on the caller's stream: nothing to launch here; the collective is ordered behind that stream
            self._given = None
            self._sync_main = True
            self._n += 1
            if self._n == self.bucket:
                self._launch()
            return
        self._given = None
        h, p = out.histories.detach(), out.paths
        B = h.shape[0]
        nb2 = 2 * ((h[0].numel() + 7) // 8)
        st = self._staging(B, nb2, h.device)
        side = self._side(h.device)
        if side is not None:
            # the outputs belong to the caller's stream: the side stream starts behind whatever is pending there (a deferred verdict leaves
            # the search itself pending), and the allocator is told that the side stream reads them (record_stream: their blocks are handed
            # out again once the pack is over -- HOLDING them for a whole bucket would make every step of the first buckets allocate afresh)
            self._event.record(torch.cuda.current_stream(h.device))
            side.wait_event(self._event)
            with torch.cuda.stream(side):
                pack_masks(h, p, out=st[self._n])
            h.record_stream(side)
            p.record_stream(side)
        else:
            pack_masks(h, p, out=st[self._n])
        self._n += 1
        if self._n == self.bucket:
            self._launch()

    def flush(self) -> List[torch.Tensor]:
        """Send the partly filled bucket, wait for everything in flight and return the collated buckets
        (``[world, steps, B, 2*ceil(HW/8)] uint8`` each; :func:`unpack_masks` on ``t[r, i]`` gives rank r's step i)."""
        if self._n:
            self._launch()
        self._wait_pending()
        done, self._done = self._done, []
        return done


def global_t_batch(group: Optional[dist.ProcessGroup] = None):
    """``BatchCoupling.mode`` callable: t_batch = max over ALL ranks of (iters) - 1 (one int32 all-reduce MAX)."""

    def fn(iters: torch.Tensor) -> torch.Tensor:
        t = (iters.amax() - 1).to(torch.int32).reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return t

    return fn


class InFlightPlanner:
    """Batches IN FLIGHT behind the planner API: whole batches go round-robin to ``streams`` HIP streams, so the tail of one launch (its
    longest search, a serial chain that leaves most of the chip idle) overlaps the bulk of the next ones.  One launch at a time is bound
    by its longest search (4096 mazes: ~12 % of the HBM roofline); with 3-4 batches in flight the same kernels sustain 2-3x the maps/s
    (DESIGN.md section 4.1).  This is what an evaluation loop over a data set wants (reference utils/training.py:63-87,
    scripts/train.py:43-50: a validation pass is a sequence of independent ``planner(map, start, goal)`` calls).

    ``planner``: a ``VanillaAstar`` / ``NeuralAstar`` (eval-mode budget, no gradients: this is an inference path).  Outputs are identical
    to sequential ``planner.forward()`` calls.

        fly = InFlightPlanner(planner, streams=4)
        outs = fly.plan_many(loader)                 # list of AstarOutput, one per batch, in order
        for out in fly.plan_iter(loader, window=8):  # or lazily, at most `window` batches in flight
            ...

    Status policy: no host wait per batch -- every launch writes its status summary into a pinned ``ops.StatusBoard`` row and the rows
    are read ONCE when the results are collected.  ``unit_cost="auto"`` (default) therefore needs no per-call wait either: a
    ``VanillaAstar`` batch goes to the unit-cost kernel optimistically and is re-run on the general kernel at collection time in the
    rare case that one of its maps is not binary.  An unsolvable map raises ``UnsolvableMapError`` at collection (naming the batch)
    unless ``check_solvable=False``."""

    def __init__(self, planner: torch.nn.Module, streams: int = 4, check_solvable: bool = True, unit_cost="auto", use_placement: bool = False):
        if streams < 1:
            raise ValueError("streams must be >= 1")
        self.planner = planner
        # a batch's placement hint (start_maps.placement_order) is for ONE launch on an otherwise empty chip: with batches in flight it front-loads
        # every launch's long searches and starves the short ones of overlap (rand32: 188 -> 151 M maps/s) -- ignored unless asked for
        self.use_placement = use_placement
        self.n_streams = int(streams)
        self.check_solvable = check_solvable
        self.unit_cost = unit_cost
        self._streams: List[torch.cuda.Stream] = []
        self._events: List[torch.cuda.Event] = []
        self._ptrs: List[int] = []
        self._device = None
        self._inflight: List[tuple] = []  # (ticket, stream index, row, inputs, outputs, flags, order, check, workspace)
        self._k = 0
        self.reruns = 0  # batches that were re-run at collection (a non-binary map under unit_cost="auto"; the batch-coupled note with g_ratio in [0.5, 1))

    def __del__(self):  # dropped with batches in flight: their status rows go back once the device has caught up (StatusBoard.retire)
        try:
            if self._inflight and self._device is not None:
                # the outputs of those batches were allocated for the current stream and are being written on the side streams: nothing may
                # hand their memory out again before those launches are over
                for st in self._streams:
                    st.synchronize()
                board = ops.StatusBoard.of(self._device)
                for item in self._inflight:
                    if item[2] >= 0:
                        board.retire(item[2], None)
                self._inflight = []
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _setup(self, device: torch.device) -> None:
        if self._device != device:
            if self._inflight:
                raise RuntimeError("InFlightPlanner: collect the batches in flight before switching devices")
            self._device = device
            self._streams = [torch.cuda.Stream(device) for _ in range(self.n_streams)]
            self._ptrs = [s.cuda_stream for s in self._streams]
            self._events = [torch.cuda.Event() for _ in range(self.n_streams)]  # "the inputs of the batch going to stream k are complete"

    def submit(self, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor, inputs_ready: bool = False) -> int:
        """Queue one batch (the reference's [B,1,H,W] tensors); returns its ticket (0, 1, 2, ... since the last collection).
        ``inputs_ready=True``: the caller guarantees the three tensors are complete (e.g. resident data-set tensors) -- the launch stream
        then does not wait for the current stream."""
        if not map_designs.is_cuda:
            raise RuntimeError("InFlightPlanner needs tensors on a HIP device (no CPU path)")
        planner = self.planner
        encode = getattr(planner, "encode", None)
        with torch.no_grad():
            if encode is not None:
                # NeuralAstar: the ENCODER stays on the current stream, one batch after the other -- its kernels fill the chip on their own and
                # share one activation workspace per module (encoder_hip.py) -- and only the search, the serial chain worth overlapping, goes to
                # the batch's stream, behind the encoder
                cost = encode(map_designs, start_maps, goal_maps)
                passable = map_designs if not planner.learn_obstacles else torch.ones_like(start_maps)
                return self.submit_search(cost, start_maps, goal_maps, passable, False)
            return self.submit_search(map_designs, start_maps, goal_maps, map_designs, inputs_ready)

    def submit_search(self, cost: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor, passable: torch.Tensor,
                      inputs_ready: bool = False) -> int:
        """Queue ONE search launch on the next stream: ``planner.astar``'s g_ratio, eval-mode budget, no gradients (the validation pair of
        ``utils.training.PlannerModule.validate`` stacks the planner's and the VanillaAstar problem into one such launch).  ``passable is
        cost`` (one binary tensor) is what the unit-cost kernel needs."""
        dev = cost.device
        self._setup(dev)
        astar = self.planner.astar
        k = self._k % self.n_streams
        st = self._streams[k]
        same = passable is cost
        with torch.no_grad():
            # copies of strided inputs are made HERE, on the current stream and before the two streams are ordered, and are held with the batch
            cost, start_maps_c, goal_maps = cost.contiguous(), start_maps.contiguous(), goal_maps.contiguous()
            passable = cost if same else passable.contiguous()
            if start_maps_c is not start_maps and hasattr(start_maps, "placement_order"):
                start_maps_c.placement_order = start_maps.placement_order
            start_maps = start_maps_c
            cur = torch.cuda.current_stream(dev)
            # the batch's inputs AND the memory the allocator hands out for its outputs belong to the CURRENT stream (a recycled block may
            # still be read by kernels pending there): the launch stream must come after whatever is still pending -- `inputs_ready` only
            # says the inputs are complete, the outputs' blocks are not the caller's to vouch for.  An idle current stream --
            # resident inputs, the usual case of an evaluation sweep -- needs no dependency, and it matters: a cross-stream event costs
            # ~20 us of GPU time per batch (probe_boundary: 45 M instead of 66 M maps/s on 4096-map maze batches)
            if not cur.query():
                ev = self._events[k]
                ev.record(cur)
                st.wait_event(ev)
            max_iters = ops.max_iters_for(start_maps.shape[-1], 1.0, False)
            unit = same and self.unit_cost in (True, "auto")
            flags = ops.FLAG_UNIT_COST if unit else 0
            # batch semantics: outside g_ratio in [0.5, 1) a finished map may not be at a fixed point of the reference's
            # batch loop -- the exact pipeline (marks + lock-step re-run of the marked maps) goes to the stream with the search; unit costs never are
            exact = (not unit) and start_maps.shape[0] > 1 and ops.coupling_possible(astar.g_ratio)
            order = check = None
            hint = getattr(start_maps, "placement_order", None) if self.use_placement else None
            if hint is not None and ops.in_lds(start_maps.shape[-2], start_maps.shape[-1]):
                o = hint.order if isinstance(hint, ops.OrderHint) else hint
                if torch.is_tensor(o) and o.numel() == start_maps.shape[0] and o.dtype == torch.int32 and o.device == dev:
                    order, check = o.reshape(-1), not getattr(hint, "trusted", False)
            board = ops.StatusBoard.of(dev)
            row = board.acquire() if (self.check_solvable or (unit and self.unit_cost == "auto")) else -1
            keep: list = []  # the launch's workspace (maps larger than LDS, a checked order): allocated for the current stream, used on stream k
            try:
                out = ops.search_nograd(cost, start_maps, goal_maps, passable, astar.g_ratio, max_iters, False, flags, order, None, bool(check),
                                        board.ptr(row) if row >= 0 else 0, self._ptrs[k], True, 0, keep, exact)
            except BaseException:
                if row >= 0:
                    board.release(row)
                raise
        ticket = len(self._inflight)
        self._inflight.append((ticket, k, row, (cost, start_maps, goal_maps, passable), out, flags, order, check, keep, exact))
        self._k += 1
        return ticket

    def collect(self) -> List[AstarOutput]:
        """Wait for every batch in flight and return their outputs in submission order (re-running / raising as the class docstring says)."""
        if not self._inflight:
            return []
        dev = self._device
        board = ops.StatusBoard.of(dev)
        for st in self._streams:
            st.synchronize()
        outs: List[AstarOutput] = []
        failed = None
        astar = self.planner.astar
        # every launch is over: read all the rows and hand them back first, so that an exception further down (a re-run that fails) leaves
        # neither rows nor batches behind
        inflight, self._inflight, self._k = self._inflight, [], 0
        summaries = []
        for item in inflight:
            row = item[2]
            r = board.read(row) if row >= 0 else None
            summaries.append(None if r is None else r.copy())
            if row >= 0:
                board.release(row)
        for (ticket, k, row, ins, out, flags, order, check, _keep, exact), summ in zip(inflight, summaries):
            hist, paths, iters, status, _ = out
            if summ is not None and summ[ops.STATUS_NOT_UNIT_COST] and self.unit_cost == "auto":
                # a non-binary map in a batch that went to the unit-cost kernel optimistically: the batch again on the general kernel
                self.reruns += 1
                cost, start_maps, goal_maps, passable = ins
                r2 = board.acquire()
                try:
                    exact = start_maps.shape[0] > 1 and ops.coupling_possible(astar.g_ratio)
                    hist, paths, iters, status, _ = ops.search_nograd(cost, start_maps, goal_maps, passable, astar.g_ratio,
                                                                      ops.max_iters_for(start_maps.shape[-1], 1.0, False), False, 0, order, None,
                                                                      bool(check), board.ptr(r2), None, True, 0, None, exact)
                    torch.cuda.current_stream(dev).synchronize()
                    r = board.read(r2)
                    summ = None if r is None else r.copy()
                finally:
                    board.release(r2)
            coupled = summ is not None and summ[ops.SUMMARY_COUPLED] and not summ[ops.SUMMARY_ERRORS].any() and status.numel() > 1
            if coupled and not exact:
                # a finished map of this batch is not at a fixed point of the reference's batch loop although g_ratio is in [0.5, 1) (negative
                # costs): the batch again through the exact pipeline, exactly as planner.forward() does
                self.reruns += 1
                cost, start_maps, goal_maps, passable = ins
                hist, paths, iters, status, _ = astar.exact_search(cost, start_maps, goal_maps, passable, ops.max_iters_for(start_maps.shape[-1], 1.0, False))
            if summ is not None and summ[ops.SUMMARY_ERRORS].any() and self.check_solvable and failed is None:
                failed = (ticket, status)
            outs.append(AstarOutput(hist, paths, []))
            astar.last_status, astar.last_iters = status, iters
        if failed is not None:
            try:
                _raise_unsolvable(failed[1], failed[0])
            except UnsolvableMapError as e:
                raise UnsolvableMapError(f"InFlightPlanner, batch #{failed[0]} since the last collection: {e}") from None
        return outs

    def plan_many(self, batches: Iterable) -> List[AstarOutput]:
        """``[planner(*b[:3]) for b in batches]`` with the batches in flight; each item is ``(map_designs, start_maps, goal_maps, ...)``."""
        try:
            for b in batches:
                self.submit(b[0], b[1], b[2])
        except BaseException:
            self.collect_quietly()  # (launches already issued keep writing their outputs: wait for them before the tensors can go away)
            raise
        return self.collect()

    def collect_quietly(self) -> None:
        """wait for every launch in flight and hand the status rows back, reporting nothing (the clean-up of an exception path)"""
        if self._inflight and self._device is not None:
            for st in self._streams:
                st.synchronize()
            board = ops.StatusBoard.of(self._device)
            for item in self._inflight:
                if item[2] >= 0:
                    board.release(item[2])
            self._inflight, self._k = [], 0

    def plan_iter(self, batches: Iterable, window: Optional[int] = None) -> Iterator[AstarOutput]:
        """Lazy form: yields the outputs in order while keeping at most ``window`` (default 2 x streams) batches in flight; a window is
        collected as a whole, so the consumer's work on window i overlaps nothing -- size it to the memory you want held."""
        window = int(window or 2 * self.n_streams)
        n = 0
        for b in batches:
            self.submit(b[0], b[1], b[2])
            n += 1
            if n == window:
                yield from self.collect()
                n = 0
        yield from self.collect()


class ShardedPlanner(torch.nn.Module):
    """Wraps a planner (``VanillaAstar`` / ``NeuralAstar``): each rank plans ITS rows of the batch; in eval mode the full-batch
    output is collated on every rank with one all-gather.

    ``forward`` takes the rank-local shard (the usual data-parallel convention: every rank's DataLoader yields its own rows;
    :func:`shard_rows` says which rows of a global batch those are for ``sharding`` = "contiguous" | "interleaved").
    ``gather``: True / False, or None (default) = collate only when not training -- the collated output is a pair of masks
    without an autograd graph, so a training step keeps the local rows (whose ``histories`` carry the gradient) and lets the
    data-parallel wrapper all-reduce parameter gradients.  ``global_batch`` (rows of the whole batch) restores the original row
    order after an interleaved gather."""

    def __init__(self, planner: torch.nn.Module, group: Optional[dist.ProcessGroup] = None, gather: Optional[bool] = None,
                 sharding: str = "contiguous"):
        super().__init__()
        self.planner = planner
        self.group = group
        self.gather = gather
        self.sharding = sharding

    def forward(self, map_designs, start_maps, goal_maps, store_intermediate_results: bool = False) -> AstarOutput:
        out = self.planner(map_designs, start_maps, goal_maps, store_intermediate_results)
        gather = (not self.training) if self.gather is None else self.gather
        if not gather or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return out
        world = dist.get_world_size(self.group)
        order = None
        if self.sharding != "contiguous":
            order = collated_order(world * out.histories.shape[0], world, self.sharding)
        return all_gather_output(out, self.group, order=order)

    def plan_many(self, batches: Iterable, streams: int = 4) -> List[AstarOutput]:
        """An evaluation sweep: every rank plans ITS shard of each batch with the batches in flight (``InFlightPlanner``), then each
        batch is collated with its one all-gather (eval-mode semantics of ``forward``)."""
        fly = InFlightPlanner(self.planner, streams=streams)
        outs = fly.plan_many(batches)
        if self.gather is False or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return outs
        world = dist.get_world_size(self.group)
        res = []
        for out in outs:
            order = None
            if self.sharding != "contiguous":
                order = collated_order(world * out.histories.shape[0], world, self.sharding)
            res.append(all_gather_output(out, self.group, order=order))
        return res
