"""PyTorch-ROCm custom ops over the C ABI (``include/nastar.h``).

``torch.ops.nastar.astar_forward`` / ``astar_backward_replay`` hand raw device pointers and torch's current HIP stream to
``libnastar_hip.so``.  PyTorch is plumbing here (allocation, streams, autograd bookkeeping); the search itself is
the hand-written HIP kernel.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _native

__all__ = ["astar_forward", "astar_backward_replay", "astar_backward_l1_replay", "l1_loss", "astar_l1_loss", "heuristic", "max_iters_for", "search_nograd", "order_from_levels", "OrderHint", "attach_order", "attach_levels",
           "StatusBoard"]


def max_iters_for(W: int, Tmax: float, training: bool) -> int:
    """Search budget exactly as the reference computes it (differentiable_astar.py:200-202)."""
    t = Tmax if training else 1.0
    return int(t * W * W)


def _require_device(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "neural_astar (MI355X-native): tensors must live on a HIP device; this package has no CPU "
                "fallback for the A* search (the reference's CPU path is the oracle under oracle/, test-only).")
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32 maps, got {t.dtype}")


FLAG_UNIT_COST = 64  # include/nastar.h NASTAR_FLAG_UNIT_COST
FLAG_LOCKSTEP = 1024  # NASTAR_FLAG_LOCKSTEP: the reference's batch loop to the letter (no exit at the goal; exactly max_iters steps)
FLAG_CHECK_ORDER = 256  # NASTAR_FLAG_CHECK_ORDER: the launch verifies `order` on the device and ignores it when it is not a permutation
FLAG_MARK_COUPLED = 32768  # NASTAR_FLAG_MARK_COUPLED: the launch marks the maps of the batch-coupled class for nastar_forward_batchloop_finish
STATUS_UNSOLVABLE = 3  # NASTAR_ERR_UNSOLVABLE (per-map status)
STATUS_NOT_UNIT_COST = 7  # NASTAR_ERR_NOT_UNIT_COST (per-map status)
SUMMARY_WORDS = 16  # NASTAR_SUMMARY_WORDS
SUMMARY_BAD_ORDER = 15  # NASTAR_SUMMARY_BAD_ORDER
SUMMARY_COUPLED = 14  # NASTAR_SUMMARY_COUPLED: a NOTE (a finished map is not at a fixed point of the reference's batch loop), cells 1..13 are errors
SUMMARY_ERRORS = slice(1, 14)
# development knob: flag bits OR-ed into every forward launch.  NASTAR_FLAG_UNIT_COST = 64 works with the product library; the A/B switches
# of csrc/nastar_dev_flags.h (NO_ASM = 8, ASM_V2 = 16, NO_DIVE = 32, ASM_V3 = 128) need the development build: NASTAR_LIB=.../libnastar_hip_dev.so
FORWARD_FLAGS = int(os.environ.get("NASTAR_FORWARD_FLAGS", "0"))
if FORWARD_FLAGS & ~(8 | 16 | 32 | 64 | 128):
    raise ValueError(f"NASTAR_FORWARD_FLAGS={FORWARD_FLAGS}: unknown flag bits (include/nastar.h NASTAR_FLAG_*, csrc/nastar_dev_flags.h)")


def coupling_possible(g_ratio: float) -> bool:
    """Can a map that reached its goal fail to be at a fixed point of the reference's batch loop with costs >= 0?
    f(n) - f(goal) = (2 g_ratio - 1) c_goal + (1 - g_ratio)(h0(n) + c_n) is positive for every g_ratio in [0.5, 1): never there.  (With
    NEGATIVE costs any g_ratio can: the launch's status summary reports it, NASTAR_SUMMARY_COUPLED.)"""
    return not (0.5 <= float(g_ratio) < 1.0)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _order_ptr(order: torch.Tensor, B: int, dev: torch.device, buffer_ok: bool = False) -> int:
    """``buffer_ok``: also accept the [B + 1] ``order_out`` buffer of the forward launch as it is (the replay backward reads its first B cells)"""
    if (order.dtype != torch.int32 or (order.numel() != B and not (buffer_ok and order.numel() == B + 1)) or order.device != dev
            or not order.is_contiguous()):
        raise ValueError(f"order must be a contiguous int32 tensor of exactly {B} elements on {dev} (got {tuple(order.shape)} {order.dtype} on {order.device})")
    return order.data_ptr()


# batches from this size on replay their backward longest-first, by the order the forward's searches finished in (one 4 x (B + 1)-byte
# fill per step buys it; 4096 mazes at Tmax 0.25: 187 -> 133 us for the replay; below ~1000 maps every search has a SIMD to itself)
PLACEMENT_MIN_BATCH = 1024


def _maps3(t: torch.Tensor) -> torch.Tensor:
    """[B,1,H,W] or [B,C,H,W] (channel 0 is used, differentiable_astar.py:177-180) -> contiguous [B,H,W]."""
    if t.ndim == 4:
        t = t[:, 0]
    return t.contiguous()


class StatusBoard:
    """Pinned host memory the search launches write their STATUS SUMMARY into (include/nastar.h: status_summary of nastar_forward_ex): one
    row of NASTAR_SUMMARY_WORDS int32 per launch in flight; cell c (1..15) becomes 1 when some map of that launch ended with per-map status
    c, cell 0 when every search of the launch is over (completion_counter: one device cell per row).  "Did any map of this batch fail?" is
    then a poll of one host word and a 64-byte read -- no reduction launch, no device-to-host copy, no stream wait, nothing on a side
    stream.  One board per device; rows are handed out and returned by the callers."""

    _boards: dict = {}

    def __init__(self, device: torch.device, rows: int = 256):
        with torch.cuda.device(device):
            self.t = torch.zeros((rows, SUMMARY_WORDS), dtype=torch.int32).pin_memory()
            self.counters = torch.zeros((rows,), dtype=torch.int32, device=device)
        self.np = self.t.numpy()
        self.base = self.t.data_ptr()
        self.cbase = self.counters.data_ptr()
        self.device = device
        self.free = list(range(rows - 1, -1, -1))
        self.zombies: list = []  # rows whose owner went away before its launch was known to be over (retire()): reaped by acquire()
        self.lib = _native.load()

    @classmethod
    def of(cls, device: torch.device) -> "StatusBoard":
        key = device.index if device.index is not None else torch.cuda.current_device()
        b = cls._boards.get(key)
        if b is None:
            b = cls._boards[key] = cls(torch.device("cuda", key))
        return b

    def acquire(self) -> int:
        if self.zombies:
            self._reap(False)
        if not self.free and self.zombies:
            self._reap(True)
        if not self.free:
            raise RuntimeError("more than 256 search launches with an unread status: call raise_if_unsolvable() / collect() on the planners that issued them")
        return self.free.pop()

    def retire(self, row: int, event: Optional["torch.cuda.Event"]) -> None:
        """give a row back although nobody read it (its planner was dropped with verdicts pending): it is reused only once its launch is over
        -- the completion flag is up, or ``event`` (recorded behind the launch) has passed"""
        self.zombies.append((row, event))

    def _reap(self, wait: bool) -> None:
        keep = []
        for row, ev in self.zombies:
            if self.np[row, 0] or (ev is not None and ev.query()):
                self.release(row)
            elif wait:
                if ev is not None:
                    ev.synchronize()
                else:
                    torch.cuda.synchronize(self.device)
                self.release(row)
            else:
                keep.append((row, ev))
        self.zombies = keep

    def ptr(self, row: int) -> int:
        return self.base + 4 * SUMMARY_WORDS * row

    def counter_ptr(self, row: int) -> int:
        return self.cbase + 4 * row

    def done(self, row: int) -> bool:
        """has the launch that was handed ``ptr(row)`` AND ``counter_ptr(row)`` finished every search?  (one host read)"""
        return bool(self.np[row, 0])

    def wait(self, row: int, stream: Optional["torch.cuda.Stream"] = None, spin_us: int = 2000) -> None:
        """return once the verdict of the launch that owns ``row`` is complete: poll the completion flag for at most ``spin_us``, then fall
        back to waiting for ``stream`` (default: the device) -- a launch that carried no completion counter (maps larger than LDS) or one
        that takes longer than the spin budget ends up there"""
        if self.np[row, 0] or (spin_us > 0 and self.lib.nastar_host_wait_nonzero(self.ptr(row), spin_us)):
            return
        if stream is not None:
            stream.synchronize()
        else:
            torch.cuda.synchronize(self.device)
        if spin_us > 0 and not self.np[row, 0]:
            # the launch was given the row's completion counter (callers spin only then) and is over without having raised the flag: an aborted
            # launch left the cell out of phase.  (Launches WITHOUT a counter -- custom-op path, maps larger than LDS -- end up here on every
            # call and must not pay a device write for it: ADVICE r5.)
            self.counters[row] = 0

    def read(self, row: int):
        """the row as a numpy view if any STATUS cell (1..15) is set, else None (the launch that was handed the row must be over: wait())"""
        r = self.np[row]
        return r if r[1:].any() else None  # (incl. the notes in cells 14 / 15: the caller tells errors -- SUMMARY_ERRORS -- from notes)

    def release(self, row: int) -> None:
        self.np[row] = 0
        self.free.append(row)


_IN_LDS: dict = {}


def in_lds(H: int, W: int) -> bool:
    """does the search state of an H x W map live in LDS (no HBM workspace; placements apply)?  Cached per size."""
    v = _IN_LDS.get((H, W))
    if v is None:
        v = _IN_LDS[(H, W)] = int(_native.load().nastar_workspace_bytes(1, H, W, 0)) == 0
    return v


def _launch_search(lib, cost, start, goal, passable, B, H, W, g_ratio, max_iters, want_log, flags, order, order_out, check_order, summary_ptr, dev,
                   one_meta=False, stream_ptr=None, out_4d=False, counter_ptr=0, keep=None, exact=False):
    """allocate the five outputs and issue ONE nastar_forward_ex launch on torch's current stream (shared by the custom ops and the
    no-autograd fast path).  cost / start / goal / passable: contiguous fp32 tensors of B*H*W elements (any leading shape).
    ``keep``: a list that receives the launch's temporaries (its workspace) when the launch goes to ANOTHER stream than the one the
    caching allocator hands the memory out for -- the caller holds them until that stream is done.
    ``exact``: the reference's BATCH LOOP to the letter (include/nastar.h: nastar_forward_batchloop_finish) -- the launch marks the maps that
    are not at a fixed point of that loop when they reach their goal, and three more launches on the same stream re-run exactly those in
    lock-step mode up to the step at which every map of the batch selects its goal.  No host round trip; nothing happens when no map is
    marked (always so for g_ratio in [0.5, 1) with costs >= 0)."""
    shape = (B, 1, H, W) if out_4d else (B, H, W)
    hist = torch.empty(shape, dtype=torch.float32, device=dev)
    paths = torch.empty(shape, dtype=torch.int64, device=dev)
    if one_meta:  # iters, status: one allocation (not for the custom ops, whose outputs must not alias each other)
        meta = torch.empty((2, B), dtype=torch.int32, device=dev)
        iters, status = meta[0], meta[1]
    else:
        iters = torch.empty((B,), dtype=torch.int32, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
    # entries at positions >= iters[b] are never read (the backward replays iters[b] steps, _intermediate_results masks by iters)
    sel_log = torch.empty((B, max_iters), dtype=torch.int32, device=dev) if want_log else (None if out_4d else torch.empty((0,), dtype=torch.int32, device=dev))
    flags = int(flags) | FORWARD_FLAGS
    op = oo = 0
    if order is not None:
        op = _order_ptr(order, B, dev)
        if check_order:
            flags |= FLAG_CHECK_ORDER
    if order_out is not None:
        if order_out.dtype != torch.int32 or order_out.numel() != B + 1 or order_out.device != dev or not order_out.is_contiguous():
            raise ValueError(f"order_out must be a contiguous int32 tensor of exactly {B + 1} elements on {dev} (ops.new_placement_buffer)")
        oo = order_out.data_ptr()
    # workspace: > 0 for maps too large for LDS, for a checked order and for the marks / probe bitmaps of an exact launch
    if exact and B > 1 and not (flags & FLAG_LOCKSTEP):
        flags |= FLAG_MARK_COUPLED
        ws_bytes = int(lib.nastar_batchloop_workspace_bytes(B, H, W, int(max_iters)))
    else:
        exact = False  # (a map on its own is its own batch: the loop ends at its goal step)
        ws_bytes = (16 if flags & FLAG_CHECK_ORDER else 0) if in_lds(H, W) else int(lib.nastar_workspace_bytes(B, H, W, flags))
    workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
    if stream_ptr is not None and workspace is not None:
        if keep is None:
            raise ValueError("a launch on a foreign stream that needs a workspace must be given a `keep` list (the allocator would hand the "
                             "workspace to the next launch on the current stream while this one still runs)")
        keep.append(workspace)
    sp = stream_ptr if stream_ptr is not None else torch.cuda.current_stream(dev).cuda_stream
    args = (cost.data_ptr(), start.data_ptr(), goal.data_ptr(), passable.data_ptr(), B, H, W, float(g_ratio), int(max_iters), hist.data_ptr(),
            paths.data_ptr(), sel_log.data_ptr() if want_log else None, iters.data_ptr(), status.data_ptr(), None,
            workspace.data_ptr() if workspace is not None else None, ws_bytes, flags, op or None, oo or None, summary_ptr or None,
            (counter_ptr or None) if summary_ptr else None, sp)
    fin = None
    if exact:
        fin = args[:9] + (hist.data_ptr(), paths.data_ptr(), sel_log.data_ptr() if want_log else None, iters.data_ptr(), status.data_ptr(),
                          workspace.data_ptr(), ws_bytes, sp)
    if dev.index is None or torch.cuda.current_device() == dev.index:
        rc = lib.nastar_forward_ex(*args)
        if not rc and fin is not None:
            rc = lib.nastar_forward_batchloop_finish(*fin)
    else:
        with torch.cuda.device(dev):
            rc = lib.nastar_forward_ex(*args)
            if not rc and fin is not None:
                rc = lib.nastar_forward_batchloop_finish(*fin)
    if rc:
        _native.check(rc, "nastar_forward_ex")
    return hist, paths, iters, status, sel_log


@torch.library.custom_op("nastar::astar_forward", mutates_args=())
def astar_forward(cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor,
                  g_ratio: float, max_iters: int, want_log: bool, flags: int = 0, summary_ptr: int = 0, exact: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (histories [B,H,W] f32, paths [B,H,W] i64, iters [B] i32, status [B] i32, sel_log [B,T] i32 or [0]).
    ``flags``: NASTAR_FLAG_* of include/nastar.h (e.g. ``FLAG_UNIT_COST`` when cost and passable are ONE binary tensor);
    ``summary_ptr``: address of a ``StatusBoard`` row (0 = none) that receives the launch's status summary; ``exact``: see ``_launch_search``."""
    _require_device(cost, start, goal, passable)
    lib = _native.load()
    cost, start, goal, passable = (x.contiguous() for x in (cost, start, goal, passable))
    B, H, W = cost.shape
    return _launch_search(lib, cost, start, goal, passable, B, H, W, g_ratio, max_iters, want_log, flags, None, None, False, summary_ptr, cost.device,
                          exact=exact)


@astar_forward.register_fake
def _(cost, start, goal, passable, g_ratio, max_iters, want_log, flags=0, summary_ptr=0, exact=False):
    B, H, W = cost.shape
    return (cost.new_empty((B, H, W)), cost.new_empty((B, H, W), dtype=torch.int64),
            cost.new_empty((B,), dtype=torch.int32), cost.new_empty((B,), dtype=torch.int32),
            cost.new_empty((B, max_iters) if want_log else (0,), dtype=torch.int32))


@torch.library.custom_op("nastar::astar_forward_ordered", mutates_args=("order_out",))
def astar_forward_ordered(cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor, g_ratio: float,
                          max_iters: int, want_log: bool, flags: int, order: Optional[torch.Tensor],
                          order_out: Optional[torch.Tensor], check_order: bool = True, summary_ptr: int = 0, exact: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``astar_forward`` with a placement (include/nastar.h: nastar_forward_ex): workgroup i searches map ``order[i]`` (int32 [B], a
    permutation of 0..B-1, or None = identity).  Same five outputs as ``astar_forward``.  ``order_out`` (int32 [B + 1] from
    ``new_placement_buffer``, or None) receives in [:B] the maps in reverse order of search completion in this launch -- the ``order``
    for the next visit of the same batch; its last cell is the launch's counter (0 before and after).  ``check_order`` (default): the
    launch verifies ``order`` on the device (one small launch) and searches in the natural order when it is not a permutation -- every
    map is searched whatever the caller passed; False only for orders that are permutations by construction (an earlier launch's
    ``order_out``, ``order_from_levels``, ``placement_predict``, an argsort).  No autograd."""
    _require_device(cost, start, goal, passable)
    lib = _native.load()
    cost, start, goal, passable = (x.contiguous() for x in (cost, start, goal, passable))
    B, H, W = cost.shape
    return _launch_search(lib, cost, start, goal, passable, B, H, W, g_ratio, max_iters, want_log, flags, order, order_out, check_order,
                          summary_ptr, cost.device, exact=exact)


@astar_forward_ordered.register_fake
def _(cost, start, goal, passable, g_ratio, max_iters, want_log, flags, order, order_out, check_order=True, summary_ptr=0, exact=False):
    B, H, W = cost.shape
    return (cost.new_empty((B, H, W)), cost.new_empty((B, H, W), dtype=torch.int64),
            cost.new_empty((B,), dtype=torch.int32), cost.new_empty((B,), dtype=torch.int32),
            cost.new_empty((B, max_iters) if want_log else (0,), dtype=torch.int32))


def search_nograd(cost_maps: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor, obstacles_maps: torch.Tensor, g_ratio: float,
                  max_iters: int, want_log: bool = False, flags: int = 0, order: Optional[torch.Tensor] = None,
                  order_out: Optional[torch.Tensor] = None, check_order: bool = True, summary_ptr: int = 0, stream_ptr: Optional[int] = None,
                  out_4d: bool = False, counter_ptr: int = 0, keep: Optional[list] = None, exact: bool = False, lib=None):
    """The search launch WITHOUT the torch.library dispatch: what ``DifferentiableAstar.forward`` calls when no gradient can flow
    (``torch.no_grad()`` / inputs that do not require one) and nothing is being traced -- the custom-op machinery costs more host time
    than the launch itself at 4096 maps.  Takes the reference's [B,1,H,W] tensors (or [B,H,W]) as they are; same five outputs
    (``out_4d``: histories / paths as [B,1,H,W], the AstarOutput layout, and None instead of an empty selection log).
    ``summary_ptr`` / ``counter_ptr``: a ``StatusBoard`` row (status summary in pinned memory + its completion counter on the device).
    ``stream_ptr``: a hipStream_t to launch on instead of torch's current stream (``parallel.InFlightPlanner``; the outputs are
    allocated on the CURRENT stream: the caller orders the two streams before anyone reads or frees them, passes contiguous inputs --
    a copy made here would be made on the current stream, after the caller ordered the streams -- and holds ``keep``, the list that
    receives the launch's workspace, until that stream is done).  ``exact``: the reference's batch loop to the letter (``_launch_search``).
    ``lib``: another build of the C ABI (``_native.load_dev()``: stream-equality tests)."""
    if stream_ptr is not None and not (cost_maps.is_contiguous() and start_maps.is_contiguous() and goal_maps.is_contiguous()
                                       and obstacles_maps.is_contiguous()):
        raise ValueError("search_nograd(stream_ptr=...): the maps must be contiguous (make the copies before ordering the streams)")
    if not (cost_maps.is_cuda and start_maps.is_cuda and goal_maps.is_cuda and obstacles_maps.is_cuda
            and cost_maps.dtype == start_maps.dtype == goal_maps.dtype == obstacles_maps.dtype == torch.float32):
        _require_device(cost_maps, start_maps, goal_maps, obstacles_maps)
    same = obstacles_maps is cost_maps
    if cost_maps.ndim == 4 and not (cost_maps.shape[1] == start_maps.shape[1] == goal_maps.shape[1] == obstacles_maps.shape[1] == 1):
        cost_maps, start_maps, goal_maps, obstacles_maps = (x[:, 0] for x in (cost_maps, start_maps, goal_maps, obstacles_maps))
    B, H, W = cost_maps.shape[0], cost_maps.shape[-2], cost_maps.shape[-1]
    n = B * H * W
    if not (start_maps.numel() == n and goal_maps.numel() == n and obstacles_maps.numel() == n
            and start_maps.shape[-2:] == cost_maps.shape[-2:] == goal_maps.shape[-2:] == obstacles_maps.shape[-2:]):
        raise ValueError("cost / start / goal / obstacle maps must have one shape")
    if not cost_maps.is_contiguous():
        cost_maps = cost_maps.contiguous()
    if same:
        obstacles_maps = cost_maps
    elif not obstacles_maps.is_contiguous():
        obstacles_maps = obstacles_maps.contiguous()
    if not start_maps.is_contiguous():
        start_maps = start_maps.contiguous()
    if not goal_maps.is_contiguous():
        goal_maps = goal_maps.contiguous()
    return _launch_search(lib if lib is not None else _native.load(), cost_maps, start_maps, goal_maps, obstacles_maps, B, H, W, g_ratio, max_iters,
                          want_log, flags, order, order_out, check_order, summary_ptr, cost_maps.device, True, stream_ptr, out_4d, counter_ptr, keep, exact)


def order_from_levels(levels: torch.Tensor) -> torch.Tensor:
    """``order`` ([B] int32) for ``astar_forward_ordered`` from data the caller already HAS: ``levels[b]`` = any non-negative number that
    grows with the expected length of map b's search -- for the reference's maze data sets the optimal distance of the sampled start
    cell, ``|opt_dists[start]|``, which every sample carries (reference utils/data.py:127-134, :200-221).  Largest first; one
    counting-sort launch (include/nastar.h: nastar_placement_from_levels).  A permutation by construction (``check_order=False``)."""
    if not levels.is_cuda:
        raise RuntimeError("order_from_levels: levels must live on a HIP device")
    lv = levels.reshape(-1).abs().to(torch.int32).contiguous() if levels.dtype != torch.int32 else levels.reshape(-1).contiguous()
    B = lv.numel()
    order = torch.empty((B,), dtype=torch.int32, device=lv.device)
    with torch.cuda.device(lv.device):
        rc = _native.load().nastar_placement_from_levels(lv.data_ptr(), B, order.data_ptr(), _stream_ptr(lv.device))
    _native.check(rc, "nastar_placement_from_levels")
    return order


class OrderHint:
    """A placement for ONE batch, attached to its ``start_maps`` tensor as ``start_maps.placement_order`` by whoever assembled the batch
    (``DeviceMazeBatches``, ``order_hint_from_distances``): the reference's 4-tuple batches keep their shape, ``PlannerModule`` and
    ``DifferentiableAstar.forward`` pick the hint up from the tensor.  ``trusted`` = a permutation by construction."""

    __slots__ = ("order", "trusted", "levels")

    def __init__(self, order: Optional[torch.Tensor], trusted: bool = False, levels: Optional[torch.Tensor] = None):
        self.order, self.trusted, self.levels = order, trusted, levels

    def resolve(self) -> Optional[torch.Tensor]:
        """the order, computed from the levels on first use (``attach_levels``)"""
        if self.order is None and self.levels is not None:
            self.order, self.trusted = order_from_levels(self.levels), True
        return self.order


def attach_order(start_maps: torch.Tensor, levels: torch.Tensor) -> torch.Tensor:
    """tag ``start_maps`` with the placement ``order_from_levels(levels)`` (the sort runs NOW: batch assembly); returns ``start_maps``"""
    start_maps.placement_order = OrderHint(order_from_levels(levels), trusted=True)
    return start_maps


def attach_levels(start_maps: torch.Tensor, levels: torch.Tensor) -> torch.Tensor:
    """tag ``start_maps`` with the LEVELS its loader has (``levels[b]`` = |opt_dists[start]| of map b, int32 [B] on the device; see
    ``order_from_levels``) and leave the sort to the ``forward()`` call that searches the batch: its counting-sort launch then goes out right in
    front of the search launch, from the same native call (csrc/nastar_fastlane.cpp).  Returns ``start_maps``."""
    lv = levels.reshape(-1)
    if lv.dtype != torch.int32:
        lv = lv.abs().to(torch.int32)
    start_maps.placement_order = OrderHint(None, trusted=True, levels=lv.contiguous())
    return start_maps


def workspace_bytes(shape) -> int:
    """bytes of HBM workspace a [B, H, W] search needs (0: the state of every map lives in LDS)"""
    B, H, W = (int(x) for x in shape[-3:])
    if in_lds(H, W):
        return 0
    return int(_native.load().nastar_workspace_bytes(B, H, W, 0))


def placement_supported(shape) -> bool:
    """sizes ``placement_predict`` handles (csrc/nastar_placement.hip.h)"""
    H, W = int(shape[-2]), int(shape[-1])
    return H == W and W in (32, 64)


def placement_predict(passable: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, return_levels: bool = False):
    """``order`` ([B] int32) for ``astar_forward_ordered`` on a batch that has never been searched: maps sorted, longest first, by the
    length of their shortest 8-connected route over passable cells (include/nastar.h: nastar_placement_predict; two small launches on
    the current stream).  [B,H,W] or [B,1,H,W] fp32 maps, 32x32 or 64x64."""
    _require_device(passable, start, goal)
    lib = _native.load()
    p, s, g = (_maps3(x) for x in (passable, start, goal))
    B, H, W = p.shape
    dev = p.device
    order = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nastar_placement_predict(p.data_ptr(), s.data_ptr(), g.data_ptr(), B, H, W, order.data_ptr(), ws.data_ptr(), B * 4, _stream_ptr(dev))
    _native.check(rc, "nastar_placement_predict")
    return (order, ws) if return_levels else order


def new_placement_buffer(B: int, device) -> torch.Tensor:
    """an ``order_out`` buffer for ``astar_forward_ordered``: int32 [B + 1], zeroed (the trailing counter cell must start at 0)"""
    return torch.zeros((B + 1,), dtype=torch.int32, device=device)


def placement_from_iters(iters: torch.Tensor) -> torch.Tensor:
    """an ``order`` for ``astar_forward_ordered`` from known (or predicted) step counts: longest searches first"""
    return torch.argsort(iters, descending=True, stable=True).to(torch.int32)


@torch.library.custom_op("nastar::astar_backward_replay", mutates_args=())
def astar_backward_replay(grad_hist: torch.Tensor, cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor,
                          passable: torch.Tensor, sel_log: torch.Tensor, g_ratio: float, max_iters: int, iters: torch.Tensor,
                          t_batch: Optional[torch.Tensor], order: Optional[torch.Tensor] = None, flags: int = 0) -> torch.Tensor:
    """dL/dcost by replaying the forward's selection log (csrc/nastar_backward_replay.hip.h): any map size the forward takes
    (1,179,648 cells; 32-bit history stamps above 65519), O(9) accounting work per step.  ``order`` (int32 permutation of 0..B-1): workgroup i replays map order[i] --
    the forward's own completion order (``astar_forward_ordered``'s ``order_out``) puts the longest replays first.  ``flags``:
    ``FLAG_LOCKSTEP`` for the log of an ``exact`` forward (goal selections before the last entry: the general replay loop)."""
    _require_device(grad_hist, cost, start, goal, passable)
    lib = _native.load()
    grad_hist, cost, start, goal, passable, sel_log = (x.contiguous() for x in (grad_hist, cost, start, goal, passable, sel_log))
    B, H, W = cost.shape
    dev = cost.device
    grad_cost = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    ws_bytes = int(lib.nastar_backward_workspace_bytes(B, H, W, int(max_iters)))
    if ws_bytes == 0:
        raise RuntimeError(f"nastar_backward_replay: unsupported map size {H}x{W}")
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if order is None:
            rc = lib.nastar_backward_replay(grad_hist.data_ptr(), cost.data_ptr(), start.data_ptr(), goal.data_ptr(),
                                            passable.data_ptr(), sel_log.data_ptr(), B, H, W, float(g_ratio), int(max_iters),
                                            iters.data_ptr(), t_batch.data_ptr() if t_batch is not None else None,
                                            grad_cost.data_ptr(), ws.data_ptr(), ws_bytes, int(flags), _stream_ptr(dev))
        else:
            rc = lib.nastar_backward_replay_ordered(grad_hist.data_ptr(), None, None, None, cost.data_ptr(), start.data_ptr(), goal.data_ptr(),
                                                    passable.data_ptr(), sel_log.data_ptr(), B, H, W, float(g_ratio), int(max_iters),
                                                    iters.data_ptr(), t_batch.data_ptr() if t_batch is not None else None,
                                                    grad_cost.data_ptr(), ws.data_ptr(), ws_bytes, int(flags), _order_ptr(order, B, dev, True), _stream_ptr(dev))
    _native.check(rc, "nastar_backward_replay")
    return grad_cost


@astar_backward_replay.register_fake
def _(grad_hist, cost, start, goal, passable, sel_log, g_ratio, max_iters, iters, t_batch, order=None, flags=0):
    return torch.empty_like(cost)


def _setup_context(ctx, inputs, output):
    cost, start, goal, passable, g_ratio, max_iters = inputs[:6]
    _, _, iters, _, sel_log = output
    ctx.save_for_backward(cost, start, goal, passable, iters, sel_log)
    ctx.g_ratio = g_ratio
    ctx.max_iters = max_iters
    ctx.lockstep = bool(inputs[9]) if len(inputs) > 9 else False  # `exact`: the log may hold goal selections before its last entry
    ctx.set_materialize_grads(False)  # no zero-filled gradient tensors for paths / iters / status / sel_log (4 fill launches per step)


def _backward(ctx, g_hist, g_paths, g_iters, g_status, g_log):
    cost, start, goal, passable, iters, sel_log = ctx.saved_tensors
    if g_hist is None:
        return (None,) * 10
    # t_batch: the reference's batch-wide loop index (differentiable_astar.py:251-255).  BatchCoupling lets the
    # sharded planner substitute the maximum over ALL ranks so gradients match a single-device run.
    t_batch = BatchCoupling.t_batch(iters)
    if sel_log.numel() == 0:
        raise RuntimeError("backward needs the forward's selection log: call astar_forward(..., want_log=True) "
                           "(DifferentiableAstar.forward does whenever cost_maps.requires_grad)")
    grad_cost = torch.ops.nastar.astar_backward_replay(g_hist.contiguous(), cost, start, goal, passable, sel_log,
                                                       ctx.g_ratio, ctx.max_iters, iters, t_batch, None, FLAG_LOCKSTEP if ctx.lockstep else 0)
    return (grad_cost,) + (None,) * 9


astar_forward.register_autograd(_backward, setup_context=_setup_context)


class BatchCoupling:
    """How a map's gradient is coupled to the rest of its batch.

    ``mode``:
      * ``"batch"`` (default, reference semantics): t_batch = max(iters) - 1 over the local batch;
      * ``"none"``: every map is its own batch (no fixed-point terms, shard-size independent);
      * a callable ``iters -> int32 device scalar`` (used by the sharded planner to all-reduce the maximum).
    """

    mode = "batch"

    @classmethod
    def t_batch(cls, iters: torch.Tensor) -> Optional[torch.Tensor]:
        if cls.mode == "none":
            return None
        if callable(cls.mode):
            return cls.mode(iters)
        return (iters.amax() - 1).to(torch.int32).reshape(1)


# ---- training step with the L1 loss fused in (SURVEY.md 8f "next #3"; reference utils/training.py:55-61) ------------------
@torch.library.custom_op("nastar::l1_loss", mutates_args=())
def l1_loss(histories: torch.Tensor, opt_trajs: torch.Tensor) -> torch.Tensor:
    """mean |histories - opt_trajs| as a 1-element device tensor (fixed-order reduction: bitwise reproducible)."""
    _require_device(histories, opt_trajs)
    if histories.shape != opt_trajs.shape:
        raise ValueError(f"shape mismatch {tuple(histories.shape)} vs {tuple(opt_trajs.shape)}")
    lib = _native.load()
    h, t = histories.contiguous(), opt_trajs.contiguous()
    dev = h.device
    out = torch.empty((1,), dtype=torch.float32, device=dev)
    ws = torch.empty((2048,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.nastar_l1_loss(h.data_ptr(), t.data_ptr(), h.numel(), out.data_ptr(), ws.data_ptr(), 2048, _stream_ptr(dev))
    _native.check(rc, "nastar_l1_loss")
    return out


@l1_loss.register_fake
def _(histories, opt_trajs):
    return histories.new_empty((1,))


@torch.library.custom_op("nastar::astar_backward_l1_replay", mutates_args=())
def astar_backward_l1_replay(histories: torch.Tensor, opt_trajs: torch.Tensor, grad_loss: Optional[torch.Tensor],
                             cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor,
                             sel_log: torch.Tensor, g_ratio: float, max_iters: int, iters: torch.Tensor,
                             t_batch: Optional[torch.Tensor], order: Optional[torch.Tensor] = None, flags: int = 0) -> torch.Tensor:
    """dL/dcost for L = grad_loss * mean|histories - opt_trajs| by replay of the selection log: the sign gradient is formed while the
    upstream values are loaded (no gradient tensor is materialised).  ``flags``: ``FLAG_LOCKSTEP`` for the log of an ``exact`` forward."""
    _require_device(histories, opt_trajs, cost, start, goal, passable)
    lib = _native.load()
    histories, opt_trajs, cost, start, goal, passable, sel_log = (
        x.contiguous() for x in (histories, opt_trajs, cost, start, goal, passable, sel_log))
    B, H, W = cost.shape
    dev = cost.device
    gl = grad_loss.reshape(1).to(torch.float32).contiguous() if grad_loss is not None else None
    grad_cost = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    ws_bytes = int(lib.nastar_backward_workspace_bytes(B, H, W, int(max_iters)))
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if order is None and not flags:
            rc = lib.nastar_backward_l1_replay(histories.data_ptr(), opt_trajs.data_ptr(), gl.data_ptr() if gl is not None else None,
                                               cost.data_ptr(), start.data_ptr(), goal.data_ptr(), passable.data_ptr(),
                                               sel_log.data_ptr(), B, H, W, float(g_ratio), int(max_iters), iters.data_ptr(),
                                               t_batch.data_ptr() if t_batch is not None else None, grad_cost.data_ptr(),
                                               ws.data_ptr(), ws_bytes, _stream_ptr(dev))
        else:
            rc = lib.nastar_backward_replay_ordered(None, histories.data_ptr(), opt_trajs.data_ptr(), gl.data_ptr() if gl is not None else None,
                                                    cost.data_ptr(), start.data_ptr(), goal.data_ptr(), passable.data_ptr(),
                                                    sel_log.data_ptr(), B, H, W, float(g_ratio), int(max_iters), iters.data_ptr(),
                                                    t_batch.data_ptr() if t_batch is not None else None, grad_cost.data_ptr(),
                                                    ws.data_ptr(), ws_bytes, int(flags), _order_ptr(order, B, dev, True) if order is not None else None,
                                                    _stream_ptr(dev))
    _native.check(rc, "nastar_backward_l1_replay")
    return grad_cost


@astar_backward_l1_replay.register_fake
def _(histories, opt_trajs, grad_loss, cost, start, goal, passable, sel_log, g_ratio, max_iters, iters, t_batch, order=None, flags=0):
    return torch.empty_like(cost)


class _AstarL1Loss(torch.autograd.Function):
    """search + L1 loss as ONE autograd node: forward = nastar_forward_ex + nastar_l1_loss, backward = nastar_backward_l1_replay."""

    @staticmethod
    def forward(ctx, cost, start, goal, passable, opt_trajs, g_ratio, max_iters, order_in, check_order, summary_ptr, exact):
        B = cost.shape[0]
        with torch.no_grad():
            order = None
            if (order_in is not None or B >= PLACEMENT_MIN_BATCH) and workspace_bytes(cost.shape) == 0:
                order = new_placement_buffer(B, cost.device)  # the forward writes the order its searches finish in
                hist, paths, iters, status, sel_log = torch.ops.nastar.astar_forward_ordered(cost, start, goal, passable, g_ratio, max_iters,
                                                                                             True, 0, order_in, order, check_order, summary_ptr, exact)
            else:
                hist, paths, iters, status, sel_log = torch.ops.nastar.astar_forward(cost, start, goal, passable, g_ratio, max_iters, True, 0, summary_ptr,
                                                                                     exact)
            loss = torch.ops.nastar.l1_loss(hist, opt_trajs)
        # (an exact forward may have re-run maps after the launch ranked their completion: the replay then takes the natural order)
        ctx.order = order[:B] if (order is not None and not exact) else None
        ctx.lockstep = bool(exact)
        ctx.save_for_backward(cost, start, goal, passable, opt_trajs, hist, iters, sel_log)
        ctx.g_ratio, ctx.max_iters = g_ratio, max_iters
        ctx.mark_non_differentiable(hist, paths, iters, status)
        ctx.set_materialize_grads(False)  # otherwise autograd zero-fills a gradient for every unused output: 5 fill launches per step
        return loss.reshape(()), hist, paths, iters, status

    @staticmethod
    def backward(ctx, g_loss, g_hist, g_paths, g_iters, g_status):
        cost, start, goal, passable, opt_trajs, hist, iters, sel_log = ctx.saved_tensors
        if g_loss is None:
            return (None,) * 11
        grad_cost = torch.ops.nastar.astar_backward_l1_replay(hist, opt_trajs, g_loss, cost, start, goal, passable, sel_log,
                                                              ctx.g_ratio, ctx.max_iters, iters, BatchCoupling.t_batch(iters), ctx.order,
                                                              FLAG_LOCKSTEP if ctx.lockstep else 0)
        return (grad_cost,) + (None,) * 10


class _AstarForwardPlaced(torch.autograd.Function):
    """``astar_forward`` for large batches under autograd: the forward launch writes the order its searches finish in and the replay
    backward starts its workgroups in that order (same values as the registered autograd of ``astar_forward``; outputs other than
    ``histories`` carry no gradient).  ``order_in`` / ``order_out``: the forward's own placement (planner.Placement / OrderHint) or None."""

    @staticmethod
    def forward(ctx, cost, start, goal, passable, g_ratio, max_iters, flags, order_in, order_out, check_order, summary_ptr, exact):
        B = cost.shape[0]
        with torch.no_grad():
            if order_out is None:
                order_out = new_placement_buffer(B, cost.device)
            hist, paths, iters, status, sel_log = torch.ops.nastar.astar_forward_ordered(cost, start, goal, passable, g_ratio, max_iters, True,
                                                                                         flags, order_in, order_out, check_order, summary_ptr, exact)
        ctx.order = order_out[:B] if not exact else None
        ctx.lockstep = bool(exact)
        ctx.save_for_backward(cost, start, goal, passable, iters, sel_log)
        ctx.g_ratio, ctx.max_iters = g_ratio, max_iters
        ctx.mark_non_differentiable(paths, iters, status, sel_log)
        ctx.set_materialize_grads(False)
        return hist, paths, iters, status, sel_log

    @staticmethod
    def backward(ctx, g_hist, g_paths, g_iters, g_status, g_log):
        if g_hist is None:
            return (None,) * 12
        cost, start, goal, passable, iters, sel_log = ctx.saved_tensors
        grad_cost = torch.ops.nastar.astar_backward_replay(g_hist.contiguous(), cost, start, goal, passable, sel_log, ctx.g_ratio, ctx.max_iters,
                                                           iters, BatchCoupling.t_batch(iters), ctx.order, FLAG_LOCKSTEP if ctx.lockstep else 0)
        return (grad_cost,) + (None,) * 11


def astar_forward_placed(cost, start, goal, passable, g_ratio: float, max_iters: int, flags: int = 0, order_in=None, order_out=None,
                         check_order: bool = True, summary_ptr: int = 0, exact: bool = False):
    """differentiable ``astar_forward`` (selection log kept) whose backward replays longest-first; see ``_AstarForwardPlaced``"""
    return _AstarForwardPlaced.apply(cost, start, goal, passable, float(g_ratio), int(max_iters), int(flags), order_in, order_out,
                                     bool(check_order), int(summary_ptr), bool(exact))


def astar_l1_loss(cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor,
                  opt_trajs: torch.Tensor, g_ratio: float, max_iters: int, order_in: Optional[torch.Tensor] = None,
                  check_order: bool = True, summary_ptr: int = 0, exact: bool = False):
    """[B,H,W] maps -> (loss scalar, histories, paths, iters, status); only ``loss`` carries gradient (to ``cost``).
    ``order_in``: a placement for the forward launch (``OrderHint.order``); the backward replays by the forward's completion order.
    ``exact``: the reference's batch loop to the letter (``_launch_search``)."""
    return _AstarL1Loss.apply(cost, start, goal, passable, opt_trajs, float(g_ratio), int(max_iters), order_in, bool(check_order), int(summary_ptr),
                              bool(exact))


def heuristic(goal_maps: torch.Tensor) -> torch.Tensor:
    """h0 = get_heuristic(goal_maps) on the device (differentiable_astar.py:26-52); parity/debug helper."""
    _require_device(goal_maps)
    lib = _native.load()
    shape = goal_maps.shape
    g = _maps3(goal_maps)
    B, H, W = g.shape
    out = torch.empty_like(g)
    with torch.cuda.device(g.device):
        rc = lib.nastar_heuristic(g.data_ptr(), B, H, W, out.data_ptr(), _stream_ptr(g.device))
    _native.check(rc, "nastar_heuristic")
    return out.reshape(shape)
