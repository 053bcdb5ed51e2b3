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

from .. import _native, ops

_ERROR_BITS = sum(1 << c for c in range(1, 14))  # summary cells 1..13 are per-map error codes (ops.SUMMARY_ERRORS); 14 / 15 are notes


class AstarOutput(NamedTuple):
    """Output structure of A* search planners (same fields/order as the reference, :16-23)."""

    histories: torch.Tensor
    paths: torch.Tensor
    intermediate_results: Optional[List[dict]] = None


# ---- the reference module's public helpers, kept importable under their names (differentiable_astar.py:26-52, :77-93, :96-125) -------------
# The search kernel fuses all three (a-2, a-4, a-7 of SURVEY.md 8a): nothing in this package calls them.  They exist for code that imports them
# from the reference's module -- same arguments, same results -- and work on device tensors (no CPU path in this package).

def get_heuristic(goal_maps: torch.Tensor, tb_factor: float = 0.001) -> torch.Tensor:
    """Chebyshev distance + ``tb_factor`` x Euclidean distance to the one-hot goal of every map (reference :26-52), same shape as ``goal_maps``.
    The default ``tb_factor`` runs the ``nastar_heuristic`` kernel (bit-exact with the reference's fp32 arithmetic, tests/test_gpu_parity.py);
    any other factor is the same sequence of fp32 operations as device tensor ops."""
    ops._require_device(goal_maps)
    if float(tb_factor) == 0.001 and goal_maps.dtype == torch.float32:
        return ops.heuristic(goal_maps)
    B, H, W = goal_maps.shape[0], goal_maps.shape[-2], goal_maps.shape[-1]
    flat = goal_maps.reshape(B, H * W)
    rows = torch.arange(H, device=goal_maps.device).repeat_interleave(W).to(goal_maps.dtype)
    cols = torch.arange(W, device=goal_maps.device).repeat(H).to(goal_maps.dtype)
    gr, gc = flat @ rows, flat @ cols                                # (one-hot goal maps: the goal's row / column)
    dr, dc = rows[None, :] - gr[:, None], cols[None, :] - gc[:, None]
    ar, ac = dr.abs(), dc.abs()
    cheb = (ar + ac) - torch.minimum(ar, ac)
    return (cheb + tb_factor * torch.sqrt(dr * dr + dc * dc)).reshape(goal_maps.shape)


def expand(x: torch.Tensor, neighbor_filter: torch.Tensor) -> torch.Tensor:
    """The 8-neighbourhood of the selected nodes ``x`` [B,H,W]: one grouped 3x3 convolution with zero padding (reference :77-93;
    ``neighbor_filter`` [B,1,3,3], ``DifferentiableAstar.neighbor_filter`` repeated per map)."""
    ops._require_device(x)
    y = torch.nn.functional.conv2d(x[None], neighbor_filter, padding=1, groups=x.shape[0])
    return y.squeeze().squeeze(0)  # (the reference's squeezes: a batch of one comes back as [H,W])


def backtrack(start_maps: torch.Tensor, goal_maps: torch.Tensor, parents: torch.Tensor, current_t: int) -> torch.Tensor:
    """Path maps (int64, shape of ``goal_maps``) from a parent table [B,HW]: the goal, then ``current_t`` hops along ``parents`` starting at the
    goal's parent, every visited cell marked (reference :96-125; like it, the walk does not stop at the start -- the initial table points every cell
    at the goal, so it re-walks cells it has marked)."""
    ops._require_device(parents)
    B = parents.shape[0]
    par = parents.to(torch.long).reshape(B, -1)
    path = goal_maps.to(torch.long).clone()
    flat = path.view(B, -1)
    loc = (par * flat).sum(-1, keepdim=True)                          # the goal's parent
    one = torch.ones_like(loc)
    for _ in range(int(current_t)):
        flat.scatter_(1, loc, one)
        loc = par.gather(1, loc)
    return path


class UnsolvableMapError(RuntimeError):
    """Raised (when ``check_solvable``) for maps whose goal is unreachable.

    The reference produces NaNs and then an ``IndexError`` inside ``backtrack`` for the whole batch
    (SURVEY.md section 0.4); here the kernel reports a per-map status instead."""


class _PendingStatus:
    """status of an earlier launch on its way to the host: the launch itself writes its summary -- and, when every search is over, its
    completion flag -- into a pinned ``ops.StatusBoard`` row.  Nothing is recorded, copied or launched for it: ``done()`` is one host
    read; ``stream`` is only the fall-back for launches that carry no completion counter (maps larger than LDS)."""

    def __init__(self, status: torch.Tensor, row: int, seq: int = 0, flagged: bool = True, repair=None):
        self.status = status
        self.seq = seq
        self.row = row
        self.repair = repair  # callable or None: completes a launch that reported NASTAR_SUMMARY_COUPLED to the reference's batch loop, in place
        self.board = ops.StatusBoard.of(status.device)
        self.stream = torch.cuda.current_stream(status.device)
        self.event = None
        self.released = False
        if not flagged:
            self.event = torch.cuda.Event()
            self.event.record(self.stream)

    def done(self) -> bool:
        return self.board.done(self.row) if self.event is None else self.event.query()

    def raise_if_unsolvable(self) -> None:
        if self.event is None:
            self.board.wait(self.row, self.stream)  # (flagged launch: spins on the completion flag first)
        else:
            self.event.synchronize()
        r = self.board.read(self.row)
        bad = r is not None and bool(r[ops.SUMMARY_ERRORS].any())
        coupled = r is not None and bool(r[ops.SUMMARY_COUPLED]) and self.status.numel() > 1
        if r is not None and r[ops.SUMMARY_BAD_ORDER]:
            _warn_bad_order()
        self.board.release(self.row)
        self.released = True
        repair, self.repair = self.repair, None
        if bad:
            _raise_unsolvable(self.status, self.seq, deferred=True)
        if coupled and repair is not None:
            repair()

    def __del__(self):  # a planner dropped with verdicts pending: the row goes back once its launch is over (never while it may still be written)
        try:
            if not self.released:
                self.released = True
                self.board.retire(self.row, self.event)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


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
    """Memory of ONE batch that truly recurs (a fixed evaluation set held in memory -- same maps AND same start cells at every visit): the
    order in which its searches finished at the previous visit.  ``planner.astar.placement = p`` before a call makes that call start the
    longest searches first (``order`` / ``order_out`` of ``nastar_forward_ex``, include/nastar.h) and leaves the order for the next visit
    in ``p`` -- no extra launch, no host synchronisation, identical outputs.  The first visit (or a visit with another batch size) runs in
    the natural order -- unless the batch carries its loader's hint (``start_maps.placement_order``, ``ops.OrderHint``), which always takes
    precedence: the reference's loaders re-draw every start cell at every visit (utils/data.py:152-166), so for THEIR batches only the
    hint describes the searches at hand."""

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
                BATCH SEMANTICS are exact in every mode for costs >= 0: a ``g_ratio`` outside [0.5, 1) -- where a map that reached its goal
                may not be at a fixed point of the reference's batch loop -- runs the exact pipeline (``ops._launch_search``: marks + lock-step
                re-run of the marked maps, no host round trip); inside [0.5, 1) only NEGATIVE costs can do that: the same-call verdict
                re-runs such a batch, a deferred verdict completes the outputs in place when it is delivered (or raises, under autograd), and
                ``False`` -- which reads nothing back -- is then each map as if searched alone (the one documented gap).
            unit_cost: the UNIT-COST search kernel (``NASTAR_FLAG_UNIT_COST``, csrc/nastar_search_unit.hip.h) for calls in which
                the cost map and the obstacle map are ONE tensor -- ``VanillaAstar.forward`` (reference astar.py:93-94) -- and no
                gradient or selection log is wanted.  On binary maps every cell the search can touch then costs 1.0, the LDS state
                needs no cost word, and 29 instead of 16 maps of 32x32 are resident per CU: same outputs, ~1.5x the maps/s with
                SEVERAL batches in flight.  One launch at a time is a serial chain that gains nothing from the layout, so
                ``forward()`` takes it only with ``True`` (the kernel CHECKS the promise per map; a map that breaks it raises
                ``ValueError``, with "deferred" checking possibly from a later call); ``"auto"`` (default) and ``False`` run the
                general kernel here.  ``parallel.InFlightPlanner`` is where "auto" means unit-cost first (with a re-run of the
                rare non-binary batch at collection).
        """
        super().__init__()
        nf = torch.ones(1, 1, 3, 3)
        nf[0, 0, 1, 1] = 0
        # never read by the kernel (the Moore-8 stencil is hard-wired) but part of every reference checkpoint
        self.neighbor_filter = nn.Parameter(nf, requires_grad=False)
        # the reference keeps its heuristic as an INSTANCE attribute (:143), i.e. as something a user may replace; the kernels hard-wire that
        # function (Chebyshev + 0.001 Euclidean): forward() refuses, loudly, to run with anything else in this attribute
        self.get_heuristic = get_heuristic
        self.g_ratio = g_ratio
        assert (Tmax > 0) & (Tmax <= 1), "Tmax must be within (0, 1]"
        self.Tmax = Tmax
        self.check_solvable = check_solvable
        self.unit_cost = unit_cost
        # collation of sharded steps (parallel.BucketedCollator): a contiguous uint8 tensor [B, 2 * ceil(HW / 8)] the NEXT call's search launch
        # writes its bit-packed masks into (2 bits per cell; fused into the launch for 32x32 / 64x64 maps).  `last_packed` is that tensor when
        # the call did fill it (the checked no-grad call through the native host lane), None otherwise (the collator then packs the outputs itself)
        self.packed_sink = None
        self.last_packed = None
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

    # ---- status protocol of ONE search launch: row = begin_launch(); launch(..., summary_ptr=summary_ptr(row)); note_status(..., row=row) ----
    def begin_launch(self, like: torch.Tensor) -> int:
        """Reserve the pinned status-summary row the next search launch of this module writes into (``ops.StatusBoard``); -1 when
        nothing will be checked (``check_solvable`` False, or a hipGraph capture is underway: nothing may synchronise there)."""
        if not self.check_solvable or not like.is_cuda or _capturing(like):
            return -1
        return ops.StatusBoard.of(like.device).acquire()

    @staticmethod
    def summary_ptr(row: int, like: torch.Tensor) -> int:
        return ops.StatusBoard.of(like.device).ptr(row) if row >= 0 else 0

    @staticmethod
    def counter_ptr(row: int, like: torch.Tensor) -> int:
        """the completion counter of the row (device cell), for launches of LDS-resident sizes; 0 otherwise"""
        return ops.StatusBoard.of(like.device).counter_ptr(row) if (row >= 0 and ops.in_lds(like.shape[-2], like.shape[-1])) else 0

    def _collect_sync(self, row: int, device: torch.device, flagged: bool = False):
        """wait for the launch that owns ``row`` and return a COPY of its summary row (None = every map ended with status 0);
        ``flagged``: the launch was given the row's completion counter"""
        board = ops.StatusBoard.of(device)
        # the ONE device->host wait of a checked call: a poll of the launch's completion flag in pinned memory (no stream wait, no driver
        # wake-up; the outputs themselves stay stream-ordered), falling back to the stream for launches without a completion counter
        board.wait(row, torch.cuda.current_stream(device), 2000 if flagged else 0)
        r = board.np[row]
        r = r.copy() if r[1:].any() else None
        board.release(row)
        return r

    def note_status(self, status: torch.Tensor, iters: torch.Tensor, clean: Optional[bool] = None, row: int = -1, flagged: bool = False,
                    repair=None) -> bool:
        """record a launch's per-map status / step counts and apply the ``check_solvable`` policy (also used by the fused training
        step and the validation pair, which launch the search themselves).  ``row``: the ``begin_launch()`` row whose address the
        launch was given as ``summary_ptr`` (``flagged``: and its ``counter_ptr``) -- the verdict is then a poll of the row's completion flag
        (unflagged: a stream wait / an event) and one 64-byte host read; without a row the status tensor is reduced on the device (one
        more launch + a blocking copy).  ``clean``: the
        caller has already read the verdict on the host (True = all zero) -- the "sync" policy then does not wait a second time.
        Returns True when the launch reported NASTAR_SUMMARY_COUPLED and the verdict was read in THIS call: the caller then runs the batch
        again with ``exact=True`` (the reference's batch loop to the letter, ``ops._launch_search``).  ``repair``: what a DEFERRED verdict
        that reports the note calls to do the same in place (``_repair_in_place``)."""
        self.last_status, self.last_iters = status, iters
        self._calls += 1
        mode = self.check_solvable
        if clean is True and row < 0 and mode is not False and mode != "deferred":
            return False  # the caller has read a clean verdict for THIS call already
        if not mode or _capturing(status) or torch.compiler.is_compiling():  # nothing may synchronise inside a hipGraph capture / a trace
            if row >= 0:
                ops.StatusBoard.of(status.device).release(row)
            return False
        if mode != "deferred":  # True / "sync": the verdict belongs to THIS call
            if clean is None and row >= 0:
                summ = self._collect_sync(row, status.device, flagged)
                row = -1
                clean = summ is None or not (summ[ops.SUMMARY_ERRORS].any())
                if summ is not None and summ[ops.SUMMARY_BAD_ORDER]:
                    _warn_bad_order()
                coupled = summ is not None and bool(summ[ops.SUMMARY_COUPLED]) and status.numel() > 1
            else:
                coupled = False
            if row >= 0:
                ops.StatusBoard.of(status.device).release(row)
            if (not clean) if clean is not None else bool((status != 0).any()):
                _raise_unsolvable(status, self._calls)
            return coupled
        if row < 0:  # a launch that carried no summary: reduce on the device into a fresh row's worth of pinned memory
            row = ops.StatusBoard.of(status.device).acquire()
            ops.StatusBoard.of(status.device).t[row, ops.STATUS_UNSOLVABLE:ops.STATUS_UNSOLVABLE + 1].copy_((status != 0).any().reshape(1), non_blocking=True)
            flagged = False
        self._pending.append(_PendingStatus(status, row, self._calls, flagged, repair))
        if len(self._pending) > 64:  # a caller that never lets the device catch up: bound the queue (one wait)
            self._pending.pop(0).raise_if_unsolvable()
        return False

    def exact_search(self, cost_maps, start_maps, goal_maps, passable, max_iters, want_log=False, out_4d=True):
        """The reference's batch loop to the letter for a batch in which a finished map is NOT at a fixed point (NASTAR_SUMMARY_COUPLED;
        DESIGN.md section 2.3): every map of the class is stepped, goal selections included, until the first step at which ALL maps of the
        batch select their goal (reference :219-225, :251) or the budget ends -- the search launch with marks + nastar_forward_batchloop_finish
        (include/nastar.h), any map size, no host round trip.  -> (histories, paths, iters, status, sel_log)"""
        return ops.search_nograd(cost_maps, start_maps, goal_maps, passable, self.g_ratio, max_iters, want_log, 0, None, None, False, 0, None, out_4d,
                                 0, None, True)

    def _repair_in_place(self, inputs, outputs, max_iters, want_log):
        """for a DEFERRED verdict: the launch it belongs to reported the note after its outputs had been handed out -- run the exact search now
        and overwrite those tensors (histories, paths, iters, status, sel_log) before the caller, who asked for the verdict first, reads them"""
        import weakref
        # weak references only: a verdict still pending must not keep the outputs of a pipelined loop (48 MB per 4096-map call, up to 64 calls
        # deep) or its inputs alive -- outputs nobody holds any more need no repair
        ins = [weakref.ref(t) for t in inputs]
        outs = [weakref.ref(t) if t is not None else None for t in outputs]

        def repair():
            live = [(r() if r is not None else None) for r in outs]
            if all(t is None for t in live):
                return
            src = [r() for r in ins]
            if any(t is None for t in src):
                raise RuntimeError("a batch searched with check_solvable='deferred' holds a map that is not at a fixed point of the reference's batch loop "
                                   "(negative costs), and its inputs were released before the verdict was collected: the outputs still alive are those of "
                                   "each map searched alone.  Keep the inputs until raise_if_unsolvable(), or use check_solvable=True (DESIGN.md section 2.3)")
            new = self.exact_search(*src, max_iters, want_log, out_4d=False)
            for old, fresh in zip(live, new):
                if old is not None and fresh is not None and old.numel() == fresh.numel():
                    old.data.copy_(fresh.reshape(old.shape))
        return repair

    def resolve_placement(self, B: int, start_maps: torch.Tensor, in_lds: bool):
        """(order, order_out, check_order, placement) for the next launch.  The ``OrderHint`` the batch's loader attached to ``start_maps``
        (``start_maps.placement_order``: by the optimal distance of THIS batch's start cells, data the sample carries) comes first -- the
        reference's loaders draw new start cells at every visit of a map (utils/data.py:152-166), so only a hint describes the batch at
        hand; without one, the module's ``placement`` (a batch that really recurs: the order its searches finished in last time).  A
        ``placement`` still records this launch's completion order either way."""
        pl, self.placement = self.placement, None
        if not in_lds:
            return None, None, False, None
        order = order_out = None
        check = False
        prev = None
        if pl is not None:
            prev, order_out = pl.buffers(B, start_maps.device)
        hint = getattr(start_maps, "placement_order", None)
        if hint is not None:
            o = hint.resolve() if isinstance(hint, ops.OrderHint) else hint
            if torch.is_tensor(o) and o.numel() == B and o.device == start_maps.device and o.dtype == torch.int32:
                order, check = o.reshape(-1), not getattr(hint, "trusted", False)
        if order is None and prev is not None:
            order = prev[:B]
        return order, order_out, check, pl

    def _forward_fast(self, cost_maps, start_maps, goal_maps, obstacles_maps) -> Optional[AstarOutput]:
        """The common call -- no gradient, default checking, an LDS-resident size, nothing tracing or capturing -- with its host side in native
        code (csrc/nastar_fastlane.cpp: output allocation, nastar_forward_ex, the poll of the launch's completion flag).  None = not a call for
        this lane (the general path below takes it and raises whatever needs raising).  Same launch, same outputs, same verdict policy."""
        fl = _native.load_fastlane()
        if fl is None:
            return None
        B, _, H, W = cost_maps.shape
        dev = cost_maps.device
        if (not cost_maps.is_cuda or not ops.in_lds(H, W) or (B > 1 and ops.coupling_possible(self.g_ratio)) or torch.compiler.is_compiling()
                or torch.cuda.is_current_stream_capturing() or dev.index != torch.cuda.current_device()):
            return None
        same = obstacles_maps is cost_maps
        flags = ops.FLAG_UNIT_COST if (same and self.unit_cost is True) else 0
        flags |= ops.FORWARD_FLAGS
        order = order_out = pl = levels = None
        ws_bytes = 0
        hint = getattr(start_maps, "placement_order", None)
        if (self.placement is None and type(hint) is ops.OrderHint and hint.order is None and hint.levels is not None and hint.levels.numel() == B
                and hint.levels.device == dev):
            levels = hint.levels  # the loader's levels: the native call sorts them into a placement right in front of the search launch
        elif self.placement is not None or hint is not None:
            order, order_out, check_order, pl = self.resolve_placement(B, start_maps, True)
            if order is not None and check_order:
                flags |= ops.FLAG_CHECK_ORDER
                ws_bytes = 16
        sink = self.packed_sink  # (a collation slot: the search launch emits the 2-bit-per-cell masks itself, include/nastar.h: packed_out)
        board = ops.StatusBoard.of(dev)
        row = board.acquire()
        try:
            hist, paths, iters, status, _, rc, verdict = fl[0].search(
                fl[1], cost_maps, start_maps, goal_maps, None if same else obstacles_maps, float(self.g_ratio),
                ops.max_iters_for(W, self.Tmax, self.training), False, flags, order, order_out, ws_bytes, board.ptr(row), board.counter_ptr(row),
                torch._C._cuda_getCurrentRawStream(dev.index), 2000, levels, fl[2], sink)
        except BaseException:
            board.release(row)
            raise
        if rc:
            board.release(row)
            if rc == -1:
                return None  # (strided / mistyped inputs: the general path copies or complains)
            _native.check(rc, "nastar_forward_ex")
        if pl is not None and order_out is not None:
            pl.commit()
        self.last_status, self.last_iters = status, iters
        self._calls += 1
        self.last_packed = sink
        if verdict == 0:  # every map ended with status 0 (the row came back zeroed)
            board.free.append(row)
            return AstarOutput(hist, paths, [])
        if verdict < 0:  # the flag did not come up within the spin budget (a very long launch): wait for the stream, read the row
            summ = self._collect_sync(row, dev, True)
            verdict = 0 if summ is None else int(sum((1 << c) for c in range(1, ops.SUMMARY_WORDS) if summ[c]))
        else:
            board.free.append(row)
        if verdict & (1 << ops.SUMMARY_BAD_ORDER):
            _warn_bad_order()
        if verdict & _ERROR_BITS:
            _raise_unsolvable(status, self._calls)
        if verdict & (1 << ops.SUMMARY_COUPLED) and B > 1:
            # a finished map of this batch is not at a fixed point of the reference's batch loop (negative costs): the batch again, exactly
            hist, paths, iters, status, _ = self.exact_search(cost_maps, start_maps, goal_maps, cost_maps if same else obstacles_maps,
                                                              ops.max_iters_for(W, self.Tmax, self.training))
            self.last_status, self.last_iters = status, iters
            self.last_packed = None  # (the slot holds the masks of the first launch)
        return AstarOutput(hist, paths, [])

    def forward(self, cost_maps: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                obstacles_maps: torch.Tensor, store_intermediate_results: bool = False) -> AstarOutput:
        assert cost_maps.ndim == 4
        assert start_maps.ndim == 4
        assert goal_maps.ndim == 4
        assert obstacles_maps.ndim == 4
        if self.get_heuristic is not get_heuristic:
            raise NotImplementedError("DifferentiableAstar.get_heuristic was replaced: the MI355X search kernels hard-wire the reference's heuristic "
                                      "(Chebyshev + 0.001 x Euclidean, differentiable_astar.py:26-52) and would silently ignore another one")
        self.last_packed = None
        if (self.check_solvable is True and not store_intermediate_results and not self._pending and type(cost_maps) is torch.Tensor
                and not (cost_maps.requires_grad and torch.is_grad_enabled())):
            out = self._forward_fast(cost_maps, start_maps, goal_maps, obstacles_maps)
            if out is not None:
                return out

        B, _, H, W = cost_maps.shape
        max_iters = ops.max_iters_for(W, self.Tmax, self.training)
        needs_grad = cost_maps.requires_grad and torch.is_grad_enabled()
        # the selection log doubles as the tape of the backward (replayed by nastar_backward_replay): keep it whenever
        # autograd will need it
        want_log = bool(store_intermediate_results) or needs_grad
        capturing = cost_maps.is_cuda and torch.cuda.is_current_stream_capturing()
        if self._pending and not capturing:
            self.raise_if_unsolvable(wait=False)  # deferred verdicts of earlier calls that have reached the host
        # VanillaAstar hands ONE tensor over as cost and obstacle map (reference astar.py:93-94): the kernel then loads it once
        same = obstacles_maps is cost_maps or (cost_maps.data_ptr() == obstacles_maps.data_ptr() and cost_maps.shape == obstacles_maps.shape
                                               and cost_maps.stride() == obstacles_maps.stride())
        mode = self.check_solvable
        # the unit-cost layout pays with SEVERAL launches in flight (more maps resident per CU); one launch at a time is a serial chain whose
        # length does not depend on the layout (probe_boundary: 117 us unit vs 114 us general per placed 4096-map launch), so forward()
        # takes it only on request -- parallel.InFlightPlanner is where "auto" means "unit-cost first"
        unit = same and not want_log and self.unit_cost is True
        in_lds = ops.in_lds(H, W)
        # a recurring batch starts its longest searches first (Placement), a fresh one by its loader's hint; maps whose state lives in HBM take no placement
        if self.placement is None and not hasattr(start_maps, "placement_order"):
            order = order_out = pl = None
            check_order = False
        else:
            order, order_out, check_order, pl = self.resolve_placement(B, start_maps, in_lds)
        dev = cost_maps.device
        compiling = torch.compiler.is_compiling()
        # (no status protocol while a hipGraph is captured or torch.compile traces: its host side would run once, at capture / trace time)
        board = ops.StatusBoard.of(dev) if (mode and cost_maps.is_cuda and not capturing and not compiling) else None
        row = board.acquire() if board is not None else -1
        sptr = board.ptr(row) if board is not None else 0
        cptr = board.counter_ptr(row) if (board is not None and in_lds) else 0
        flags = ops.FLAG_UNIT_COST if unit else 0
        traced = needs_grad or type(cost_maps) is not torch.Tensor or compiling
        passable_maps = cost_maps if same else obstacles_maps
        # Batch semantics.  For g_ratio in [0.5, 1) with costs >= 0 a finished map is at a fixed point of the reference's
        # batch loop and ONE launch is the whole story; the launch reports the rare exception (negative costs) in its status summary.  Outside
        # that range the class is reachable with ordinary costs: the exact pipeline runs straight away (marks + three launches that do nothing
        # when no map is marked) -- under autograd, with deferred or no checking, inside a hipGraph capture or a trace alike.
        exact = B > 1 and ops.coupling_possible(self.g_ratio) and not unit

        def launch(exact_now: bool, sptr_now: int, cptr_now: int):
            if not traced:
                # no gradient can flow and nothing is tracing: straight to the C ABI (no torch.library dispatch)
                return ops.search_nograd(cost_maps, start_maps, goal_maps, passable_maps, self.g_ratio, max_iters, want_log, flags, order, order_out,
                                         check_order, sptr_now, None, True, cptr_now, None, exact_now)
            cost, start, goal, passable = cost_maps[:, 0], start_maps[:, 0], goal_maps[:, 0], obstacles_maps[:, 0]
            if needs_grad and in_lds and (order is not None or order_out is not None or B >= ops.PLACEMENT_MIN_BATCH):
                # large batches under autograd: the replay backward starts longest-first, by the order THIS forward's searches finish in
                o = ops.astar_forward_placed(cost, start, goal, passable, self.g_ratio, max_iters, 0, order, order_out, check_order, sptr_now, exact_now)
            elif order is None and order_out is None:
                o = torch.ops.nastar.astar_forward(cost, start, goal, passable, float(self.g_ratio), max_iters, want_log, flags, sptr_now, exact_now)
            else:
                o = torch.ops.nastar.astar_forward_ordered(cost, start, goal, passable, float(self.g_ratio), max_iters, want_log, flags, order, order_out,
                                                           check_order, sptr_now, exact_now)
            return o[0].unsqueeze(1), o[1].unsqueeze(1), o[2], o[3], o[4]

        try:
            hist, paths, iters, status, sel_log = launch(exact, sptr, cptr)
        except BaseException:
            if row >= 0:
                board.release(row)
            raise
        if pl is not None and order_out is not None:
            pl.commit()
        repair = None
        if mode == "deferred" and not exact and B > 1 and row >= 0:
            if needs_grad:
                repair = _refuse_late_repair
            else:  # (no graph holds these tensors: a late verdict that reports the note completes them in place)
                repair = self._repair_in_place((cost_maps, start_maps, goal_maps, passable_maps), (hist, paths, iters, status, sel_log), max_iters, want_log)
        coupled = self.note_status(status, iters, None, row, flagged=bool(cptr) and not traced, repair=repair)
        if coupled and not exact:
            # the same-call verdict says a finished map of this batch is not at a fixed point (negative costs): the batch again, exactly
            order = order_out = None
            hist, paths, iters, status, sel_log = launch(True, 0, 0)
            self.last_status, self.last_iters = status, iters

        intermediate_results: List[dict] = []
        if store_intermediate_results:
            intermediate_results = _intermediate_results(hist[:, 0], paths[:, 0], goal_maps[:, 0], iters, sel_log)
        return AstarOutput(hist, paths, intermediate_results)


_BAD_ORDER_WARNED = False


def _refuse_late_repair() -> None:
    """``check_solvable="deferred"`` under autograd, and the late verdict reports NASTAR_SUMMARY_COUPLED (possible only with NEGATIVE costs when
    g_ratio is in [0.5, 1); every other g_ratio runs the exact pipeline up front): the graph of that call already holds the early-exit
    launch's selection log.  Loud, not silent."""
    raise RuntimeError("a batch searched with check_solvable='deferred' under autograd holds a map that is not at a fixed point of the reference's "
                       "batch loop (negative costs with g_ratio in [0.5, 1)): its histories and gradients depend on the rest of the batch and the call "
                       "that could have completed them has returned.  Use check_solvable=True (the default) for such inputs (DESIGN.md section 2.3)")


def _warn_bad_order() -> None:
    global _BAD_ORDER_WARNED
    if not _BAD_ORDER_WARNED:
        _BAD_ORDER_WARNED = True
        import warnings
        warnings.warn("a placement order handed to the search was not a permutation of 0..B-1: it was ignored (natural order, identical "
                      "outputs); build orders with ops.order_from_levels / Placement", RuntimeWarning, stacklevel=3)


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
