"""``DifferentiableAstar`` -- host shim over the HIP search kernel.

Mirrors the public surface of the reference module (``planner/differentiable_astar.py``): ``AstarOutput``
(:16-23), ``DifferentiableAstar(g_ratio, Tmax)`` (:128-148) with the ``neighbor_filter`` parameter kept for
state-dict compatibility (:140-143), and ``forward(cost_maps, start_maps, goal_maps, obstacles_maps,
store_intermediate_results)`` (:150-267).  The body of the reference's loop is NOT here: it is
``csrc/nastar_search.hip.h`` reached through ``torch.ops.nastar.astar_forward``.
"""
from __future__ import annotations

from typing import List, NamedTuple, Optional

import torch
import torch.nn as nn

from .. import ops


class AstarOutput(NamedTuple):
    """Output structure of A* search planners (same fields/order as the reference, :16-23)."""

    histories: torch.Tensor
    paths: torch.Tensor
    intermediate_results: Optional[List[dict]] = None


class UnsolvableMapError(RuntimeError):
    """Raised (when ``check_solvable``) for maps whose goal is unreachable.

    The reference produces NaNs and then an ``IndexError`` inside ``backtrack`` for the whole batch
    (SURVEY.md section 0.4); here the kernel reports a per-map status instead."""


_SIDE_STREAMS: dict = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[key]


class _PendingStatus:
    """status of an earlier call on its way to the host: a device-side any() + a non-blocking copy into pinned memory + an event, all on
    a SIDE stream that waits for the search launch -- the stream the caller keeps launching on never sees these three small ops"""

    def __init__(self, status: torch.Tensor, seq: int = 0):
        self.status = status
        self.seq = seq
        self.flag = torch.empty((1,), dtype=torch.bool, pin_memory=True)
        main = torch.cuda.current_stream(status.device)
        side = _side_stream(status.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.flag.copy_((status != 0).any().reshape(1), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(side)
        status.record_stream(side)

    def done(self) -> bool:
        return self.event.query()

    def raise_if_unsolvable(self) -> None:
        self.event.synchronize()
        if bool(self.flag[0]):
            _raise_unsolvable(self.status, self.seq, deferred=True)


def _capturing(t: torch.Tensor) -> bool:
    """is a hipGraph capture underway on the stream this tensor's work goes to?  (CPU tensors: no -- the op itself rejects them)"""
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def _raise_unsolvable(status: torch.Tensor, seq: int = 0, deferred: bool = False) -> None:
    nu = torch.nonzero(status == ops.STATUS_NOT_UNIT_COST).flatten().tolist()
    if nu:
        raise ValueError(f"unit_cost=True, but {len(nu)} map(s) hold values other than 0.0 / 1.0 (batch rows {nu[:16]}"
                         f"{'...' if len(nu) > 16 else ''} of search call #{seq}): their outputs are empty; use unit_cost='auto' or False")
    bad = torch.nonzero(status != 0).flatten().tolist()
    where = f"search call #{seq} of this module" + (" (an EARLIER call: check_solvable='deferred' delivers verdicts late)" if deferred else "")
    raise UnsolvableMapError(
        f"{len(bad)} map(s) have no start->goal route or a non-one-hot start/goal map "
        f"(batch rows {bad[:16]}{'...' if len(bad) > 16 else ''} of {where})")


class Placement:
    """Memory of ONE recurring batch (a validation batch, an evaluation set searched every epoch): the order in which its searches
    finished at the previous visit.  ``planner.astar.placement = p`` before a call makes that call start the longest searches first
    (``nastar_forward_ordered``, include/nastar.h) and leaves the order for the next visit in ``p`` -- no extra launch, no host
    synchronisation, identical outputs.  The first visit (or a visit with another batch size) runs in the natural order."""

    def __init__(self):
        self.bufs: Optional[List[torch.Tensor]] = None
        self.k = 0            # bufs[k] holds the order of the latest visit
        self.valid = False    # ... once one visit has run

    def buffers(self, B: int, device) -> "tuple[Optional[torch.Tensor], torch.Tensor]":
        """(order to use now or None, buffer that receives the order of this visit).  Nothing changes until ``commit()``: a call that
        fails before its launch must not leave a half-initialised order behind for the next visit."""
        if self.bufs is None or self.bufs[0].numel() != B + 1 or self.bufs[0].device != device:
            self.bufs = [ops.new_placement_buffer(B, device) for _ in range(2)]
            self.k, self.valid = 0, False
        return (self.bufs[self.k] if self.valid else None), self.bufs[self.k ^ 1]

    def commit(self) -> None:
        """the launch that was handed ``buffers()`` has been issued: its output buffer is the order of the latest visit"""
        self.k ^= 1
        self.valid = True

    def __getstate__(self):  # device scratch is not state (deepcopy / pickle of a planner that holds one)
        return {"bufs": None, "k": 0, "valid": False}


def _search(cost, start, goal, passable, g_ratio, max_iters, want_log, flags, order=None, order_out=None):
    """the one launch (csrc/nastar_capi.hip::nastar_forward / nastar_forward_ordered) behind DifferentiableAstar.forward"""
    if order is None and order_out is None:
        return torch.ops.nastar.astar_forward(cost, start, goal, passable, g_ratio, max_iters, want_log, flags)
    return torch.ops.nastar.astar_forward_ordered(cost, start, goal, passable, g_ratio, max_iters, want_log, flags, order, order_out)


class DifferentiableAstar(nn.Module):
    def __init__(self, g_ratio: float = 0.5, Tmax: float = 1.0, check_solvable=True, unit_cost="auto"):
        """
        Args:
            g_ratio: weight of g(v) in f = g_ratio*g + (1-g_ratio)*h; 0 = best-first search (reference :129-135).
            Tmax: fraction of W*W search steps allowed in training mode (reference :135,:200-202).
            check_solvable: what to do about maps whose open list ran empty (extension over the reference, which crashes with an
                ``IndexError`` inside ``backtrack`` for the whole batch):
                ``True`` / ``"sync"`` (default) -- wait for the kernel and raise ``UnsolvableMapError`` IN THE SAME CALL, before
                the caller can consume garbage histories or step an optimiser on them (one device->host sync per call: ~27 us on
                a 4096-map 32x32 batch; the reference synchronises once per search ITERATION);
                ``"deferred"`` (opt-in: benchmarks, pipelined inference loops) -- NO host synchronisation in ``forward()``: the
                per-map status travels to the host asynchronously (a device-side any() + a copy into pinned memory + an event)
                and the error is raised by the first LATER ``forward()`` call that finds the verdict already on the host, or by
                ``raise_if_unsolvable()``, which waits (call it after the last batch / before consuming results); the message
                names the call it belongs to;
                ``False`` -- never raise.  Inside a hipGraph capture nothing is checked (nothing may synchronise there).
                The per-map status of the latest call is always available as ``self.last_status``.
            unit_cost: the UNIT-COST search kernel (``NASTAR_FLAG_UNIT_COST``, csrc/nastar_search_unit.hip.h) for calls in which
                the cost map and the obstacle map are ONE tensor -- ``VanillaAstar.forward`` (reference astar.py:93-94) -- and no
                gradient or selection log is wanted.  On binary maps every cell the search can touch then costs 1.0, the LDS state
                needs no cost word, and 29 instead of 16 maps of 32x32 are resident per CU: same outputs, ~1.5x the maps/s with
                several batches in flight.  The kernel CHECKS the promise per map.  ``"auto"`` (default): taken whenever this call
                reads the status itself (``check_solvable`` True / "sync", no graph capture) -- a batch with a non-binary map is then
                re-run on the general kernel inside the same call, so the result never depends on the promise; ``True``: always
                (a map that breaks the promise raises ``ValueError``, with "deferred" checking possibly from a later call);
                ``False``: never.
        """
        super().__init__()
        nf = torch.ones(1, 1, 3, 3)
        nf[0, 0, 1, 1] = 0
        # never read by the kernel (the Moore-8 stencil is hard-wired) but part of every reference checkpoint
        self.neighbor_filter = nn.Parameter(nf, requires_grad=False)
        self.g_ratio = g_ratio
        assert (Tmax > 0) & (Tmax <= 1), "Tmax must be within (0, 1]"
        self.Tmax = Tmax
        self.check_solvable = check_solvable
        self.unit_cost = unit_cost
        self.last_status: Optional[torch.Tensor] = None
        self.last_iters: Optional[torch.Tensor] = None
        self._pending: List[_PendingStatus] = []
        self._calls = 0  # searches launched through this module (names the call in UnsolvableMapError)
        # placement memory of the batch the NEXT forward() searches (see Placement); consumed by that call
        self.placement: Optional[Placement] = None

    # run-time bookkeeping (device events, pinned flags, the latest status tensors) is not module state: copy.deepcopy(planner)
    # (EMA / best-model snapshots), pickling and torch.save(planner) must work after any forward()
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_pending"] = []
        state["last_status"] = None
        state["last_iters"] = None
        state["placement"] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("_pending", [])
        self.__dict__.setdefault("_calls", 0)
        self.__dict__.setdefault("placement", None)

    def raise_if_unsolvable(self, wait: bool = True) -> None:
        """Deliver the deferred verdicts: raise ``UnsolvableMapError`` if an earlier ``forward()`` call met an unsolvable map.
        ``wait=True`` (explicit calls) waits for every outstanding launch; ``wait=False`` (what ``forward()`` does) only looks at
        verdicts that have already reached the host."""
        while self._pending and (wait or self._pending[0].done()):
            self._pending.pop(0).raise_if_unsolvable()

    def note_status(self, status: torch.Tensor, iters: torch.Tensor, clean: Optional[bool] = None) -> None:
        """record a launch's per-map status / step counts and apply the ``check_solvable`` policy (also used by the fused training
        step and the validation pair, which launch the search themselves).  ``clean``: the caller has already read ``status`` on the
        host (True = all zero, False = some map failed) -- the "sync" policy then does not wait a second time."""
        self.last_status, self.last_iters = status, iters
        self._calls += 1
        mode = self.check_solvable
        if not mode or _capturing(status):  # nothing may synchronise inside a hipGraph capture
            return
        if mode != "deferred":  # True / "sync": the verdict belongs to THIS call
            if (not clean) if clean is not None else bool((status != 0).any()):
                _raise_unsolvable(status, self._calls)
            return
        self._pending.append(_PendingStatus(status, self._calls))
        if len(self._pending) > 64:  # a caller that never lets the device catch up: bound the queue (one wait)
            self._pending.pop(0).raise_if_unsolvable()

    def forward(self, cost_maps: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                obstacles_maps: torch.Tensor, store_intermediate_results: bool = False) -> AstarOutput:
        assert cost_maps.ndim == 4
        assert start_maps.ndim == 4
        assert goal_maps.ndim == 4
        assert obstacles_maps.ndim == 4

        cost = cost_maps[:, 0]
        start = start_maps[:, 0]
        goal = goal_maps[:, 0]
        passable = obstacles_maps[:, 0]
        W = cost.shape[-1]
        max_iters = ops.max_iters_for(W, self.Tmax, self.training)

        # the selection log doubles as the tape of the backward (replayed by nastar_backward_replay): keep it whenever
        # autograd will need it
        want_log = bool(store_intermediate_results) or (
            torch.is_grad_enabled() and cost_maps.requires_grad)
        if not _capturing(cost_maps):
            self.raise_if_unsolvable(wait=False)  # deferred verdicts of earlier calls that have reached the host
        # VanillaAstar hands ONE tensor over as cost and obstacle map: the unit-cost kernel (see __init__) applies when it is binary,
        # which the kernel itself checks; in "auto" mode only when this call reads the status anyway and can fall back
        same = (cost.data_ptr() == passable.data_ptr() and cost.shape == passable.shape and cost.stride() == passable.stride())
        sync_check = self.check_solvable in (True, "sync") and not _capturing(cost_maps)
        unit = (same and not want_log and not (torch.is_grad_enabled() and cost_maps.requires_grad)
                and (self.unit_cost is True or (self.unit_cost == "auto" and sync_check)))
        # a recurring batch starts its longest searches first (Placement); gradients need the autograd-registered op, and maps whose
        # state lives in HBM take no placement
        order = order_out = None
        pl, self.placement = self.placement, None
        in_lds = ops.workspace_bytes(cost.shape) == 0
        needs_grad = torch.is_grad_enabled() and cost_maps.requires_grad
        if pl is not None and in_lds:
            order, order_out = pl.buffers(cost.shape[0], cost.device)
        if needs_grad and in_lds and (order_out is not None or cost.shape[0] >= ops.PLACEMENT_MIN_BATCH):
            # large batches under autograd: the replay backward starts longest-first, by the order THIS forward's searches finish in
            hist, paths, iters, status, sel_log = ops.astar_forward_placed(cost, start, goal, passable, self.g_ratio, max_iters, 0, order, order_out)
        else:
            hist, paths, iters, status, sel_log = _search(
                cost, start, goal, passable, float(self.g_ratio), max_iters, want_log, ops.FLAG_UNIT_COST if unit else 0, order, order_out)
        clean = None
        if unit and self.unit_cost == "auto":
            clean = not bool((status != 0).any())  # the ONE device->host wait of this call (note_status does not wait again)
            if not clean and bool((status == ops.STATUS_NOT_UNIT_COST).any()):
                # a map with values other than 0 / 1: the whole batch again on the general kernel (same call, same outputs contract)
                hist, paths, iters, status, sel_log = _search(
                    cost, start, goal, passable, float(self.g_ratio), max_iters, want_log, 0, order, order_out)
                clean = None
        if pl is not None and order_out is not None:
            pl.commit()
        self.note_status(status, iters, clean)

        intermediate_results: List[dict] = []
        if store_intermediate_results:
            intermediate_results = _intermediate_results(hist, paths, goal, iters, sel_log)
        return AstarOutput(hist.unsqueeze(1), paths.unsqueeze(1), intermediate_results)


def _intermediate_results(hist, paths, goal, iters, sel_log) -> List[dict]:
    """Rebuild the reference's per-step side channel (:210-216,:257-263) from the kernel's selection log.

    Entry t holds the histories BEFORE step t and the node selected AT step t; a map that already reached its goal
    keeps re-selecting it (fixed point) until the slowest map of the batch is done; one final entry carries the
    final histories and the int64 path maps."""
    B, H, W = hist.shape
    t_batch = int(iters.max().item()) - 1
    goal_idx = goal.reshape(B, -1).argmax(-1)
    rows = torch.arange(B, device=hist.device)
    cur = torch.zeros((B, H * W), dtype=hist.dtype, device=hist.device)
    out: List[dict] = []
    for t in range(t_batch + 1):
        sel = torch.where(t < iters, sel_log[:, t].long(), goal_idx)
        onehot = torch.zeros_like(cur)
        onehot[rows, sel] = 1
        out.append({"histories": cur.reshape(B, 1, H, W).clone(), "paths": onehot.reshape(B, 1, H, W)})
        cur[rows, sel] = 1
    out.append({"histories": hist.unsqueeze(1).detach(), "paths": paths.unsqueeze(1).detach()})
    return out
