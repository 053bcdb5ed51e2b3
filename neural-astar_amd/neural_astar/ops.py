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

__all__ = ["astar_forward", "astar_backward_replay", "astar_backward_l1_replay", "l1_loss", "astar_l1_loss", "heuristic", "max_iters_for"]


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
STATUS_NOT_UNIT_COST = 7  # NASTAR_ERR_NOT_UNIT_COST (per-map status)
# development knob: NASTAR_FORWARD_FLAGS=1 forces the LDS-resident kernel, =2 the register-resident one (include/nastar.h)
FORWARD_FLAGS = int(os.environ.get("NASTAR_FORWARD_FLAGS", "0"))
CHECK_ORDER = os.environ.get("NASTAR_CHECK_ORDER", "0") not in ("", "0")  # verify every placement handed to astar_forward_ordered (debug)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _order_ptr(order: torch.Tensor, B: int, dev: torch.device) -> int:
    if order.dtype != torch.int32 or order.numel() < B or order.device != dev or not order.is_contiguous():
        raise ValueError(f"order must be a contiguous int32 tensor of at least {B} elements on {dev}")
    return order.data_ptr()


# batches from this size on replay their backward longest-first, by the order the forward's searches finished in (one 4 x (B + 1)-byte
# fill per step buys it; 4096 mazes at Tmax 0.25: 187 -> 133 us for the replay; below ~1000 maps every search has a SIMD to itself)
PLACEMENT_MIN_BATCH = 1024


def _maps3(t: torch.Tensor) -> torch.Tensor:
    """[B,1,H,W] or [B,C,H,W] (channel 0 is used, differentiable_astar.py:177-180) -> contiguous [B,H,W]."""
    if t.ndim == 4:
        t = t[:, 0]
    return t.contiguous()


@torch.library.custom_op("nastar::astar_forward", mutates_args=())
def astar_forward(cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor,
                  g_ratio: float, max_iters: int, want_log: bool, flags: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (histories [B,H,W] f32, paths [B,H,W] i64, iters [B] i32, status [B] i32, sel_log [B,T] i32 or [0]).
    ``flags``: NASTAR_FLAG_* of include/nastar.h (e.g. ``FLAG_UNIT_COST`` when cost and passable are ONE binary tensor)."""
    _require_device(cost, start, goal, passable)
    lib = _native.load()
    cost, start, goal, passable = (x.contiguous() for x in (cost, start, goal, passable))
    B, H, W = cost.shape
    dev = cost.device
    hist = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    paths = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    # entries at positions >= iters[b] are never read (the backward replays iters[b] steps, _intermediate_results masks by iters)
    sel_log = torch.empty((B, max_iters) if want_log else (0,), dtype=torch.int32, device=dev)
    flags = int(flags) | FORWARD_FLAGS
    ws_bytes = int(lib.nastar_workspace_bytes(B, H, W, flags))  # > 0 only for maps too large for LDS
    workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
    with torch.cuda.device(dev):
        rc = lib.nastar_forward(cost.data_ptr(), start.data_ptr(), goal.data_ptr(), passable.data_ptr(), B, H, W,
                                float(g_ratio), int(max_iters), hist.data_ptr(), paths.data_ptr(),
                                sel_log.data_ptr() if want_log else None, iters.data_ptr(), status.data_ptr(),
                                workspace.data_ptr() if workspace is not None else None, ws_bytes, flags,
                                _stream_ptr(dev))
    _native.check(rc, "nastar_forward")
    return hist, paths, iters, status, sel_log


@astar_forward.register_fake
def _(cost, start, goal, passable, g_ratio, max_iters, want_log, flags=0):
    B, H, W = cost.shape
    return (cost.new_empty((B, H, W)), cost.new_empty((B, H, W), dtype=torch.int64),
            cost.new_empty((B,), dtype=torch.int32), cost.new_empty((B,), dtype=torch.int32),
            cost.new_empty((B, max_iters) if want_log else (0,), dtype=torch.int32))


@torch.library.custom_op("nastar::astar_forward_ordered", mutates_args=("order_out",))
def astar_forward_ordered(cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor, g_ratio: float,
                          max_iters: int, want_log: bool, flags: int, order: Optional[torch.Tensor],
                          order_out: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``astar_forward`` with a placement (include/nastar.h: nastar_forward_ordered): workgroup i searches map ``order[i]`` (int32
    permutation of 0..B-1, or None = identity).  Same five outputs as ``astar_forward``.  ``order_out`` (int32 [B + 1] from
    ``new_placement_buffer``, or None) receives in [:B] the maps in reverse order of search completion in this launch -- the ``order``
    for the next visit of the same batch; its last cell is the launch's counter (0 before and after).  No autograd."""
    _require_device(cost, start, goal, passable)
    lib = _native.load()
    cost, start, goal, passable = (x.contiguous() for x in (cost, start, goal, passable))
    B, H, W = cost.shape
    dev = cost.device
    for name, t, n in (("order", order, B), ("order_out", order_out, B + 1)):
        if t is not None and (t.dtype != torch.int32 or t.numel() < n or t.device != dev or not t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous int32 tensor of at least {n} elements on {dev}")
    if CHECK_ORDER and order is not None and not torch.cuda.is_current_stream_capturing():
        # debugging aid (NASTAR_CHECK_ORDER=1; one host synchronisation): a map that `order` never names is never searched
        if not torch.equal(torch.sort(order[:B].long()).values, torch.arange(B, device=dev)):
            raise ValueError("order is not a permutation of 0..B-1")
    hist = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    paths = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    sel_log = torch.empty((B, max_iters) if want_log else (0,), dtype=torch.int32, device=dev)
    flags = int(flags) | FORWARD_FLAGS
    with torch.cuda.device(dev):
        rc = lib.nastar_forward_ordered(cost.data_ptr(), start.data_ptr(), goal.data_ptr(), passable.data_ptr(), B, H, W,
                                        float(g_ratio), int(max_iters), hist.data_ptr(), paths.data_ptr(),
                                        sel_log.data_ptr() if want_log else None, iters.data_ptr(), status.data_ptr(), None, None, 0,
                                        flags, order.data_ptr() if order is not None else None,
                                        order_out.data_ptr() if order_out is not None else None, _stream_ptr(dev))
    _native.check(rc, "nastar_forward_ordered")
    return hist, paths, iters, status, sel_log


@astar_forward_ordered.register_fake
def _(cost, start, goal, passable, g_ratio, max_iters, want_log, flags, order, order_out):
    B, H, W = cost.shape
    return (cost.new_empty((B, H, W)), cost.new_empty((B, H, W), dtype=torch.int64),
            cost.new_empty((B,), dtype=torch.int32), cost.new_empty((B,), dtype=torch.int32),
            cost.new_empty((B, max_iters) if want_log else (0,), dtype=torch.int32))


def workspace_bytes(shape) -> int:
    """bytes of HBM workspace a [B, H, W] search needs (0: the state of every map lives in LDS)"""
    B, H, W = (int(x) for x in shape[-3:])
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
                          t_batch: Optional[torch.Tensor], order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/dcost by replaying the forward's selection log (csrc/nastar_backward_replay.hip.h): any map size the forward takes
    up to 65519 cells, O(9) accounting work per step.  ``order`` (int32 permutation of 0..B-1): workgroup i replays map order[i] --
    the forward's own completion order (``astar_forward_ordered``'s ``order_out``) puts the longest replays first."""
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
                                            grad_cost.data_ptr(), ws.data_ptr(), ws_bytes, 0, _stream_ptr(dev))
        else:
            rc = lib.nastar_backward_replay_ordered(grad_hist.data_ptr(), None, None, None, cost.data_ptr(), start.data_ptr(), goal.data_ptr(),
                                                    passable.data_ptr(), sel_log.data_ptr(), B, H, W, float(g_ratio), int(max_iters),
                                                    iters.data_ptr(), t_batch.data_ptr() if t_batch is not None else None,
                                                    grad_cost.data_ptr(), ws.data_ptr(), ws_bytes, 0, _order_ptr(order, B, dev), _stream_ptr(dev))
    _native.check(rc, "nastar_backward_replay")
    return grad_cost


@astar_backward_replay.register_fake
def _(grad_hist, cost, start, goal, passable, sel_log, g_ratio, max_iters, iters, t_batch, order=None):
    return torch.empty_like(cost)


def _setup_context(ctx, inputs, output):
    cost, start, goal, passable, g_ratio, max_iters = inputs[:6]
    _, _, iters, _, sel_log = output
    ctx.save_for_backward(cost, start, goal, passable, iters, sel_log)
    ctx.g_ratio = g_ratio
    ctx.max_iters = max_iters
    ctx.set_materialize_grads(False)  # no zero-filled gradient tensors for paths / iters / status / sel_log (4 fill launches per step)


def _backward(ctx, g_hist, g_paths, g_iters, g_status, g_log):
    cost, start, goal, passable, iters, sel_log = ctx.saved_tensors
    if g_hist is None:
        return None, None, None, None, None, None, None, None
    # t_batch: the reference's batch-wide loop index (differentiable_astar.py:251-255).  BatchCoupling lets the
    # sharded planner substitute the maximum over ALL ranks so gradients match a single-device run.
    t_batch = BatchCoupling.t_batch(iters)
    if sel_log.numel() == 0:
        raise RuntimeError("backward needs the forward's selection log: call astar_forward(..., want_log=True) "
                           "(DifferentiableAstar.forward does whenever cost_maps.requires_grad)")
    grad_cost = torch.ops.nastar.astar_backward_replay(g_hist.contiguous(), cost, start, goal, passable, sel_log,
                                                       ctx.g_ratio, ctx.max_iters, iters, t_batch)
    return grad_cost, None, None, None, None, None, None, None


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
                             t_batch: Optional[torch.Tensor], order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/dcost for L = grad_loss * mean|histories - opt_trajs| by replay of the selection log: the sign gradient is formed while the
    upstream values are loaded (no gradient tensor is materialised)."""
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
        if order is None:
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
                                                    ws.data_ptr(), ws_bytes, 0, _order_ptr(order, B, dev), _stream_ptr(dev))
    _native.check(rc, "nastar_backward_l1_replay")
    return grad_cost


@astar_backward_l1_replay.register_fake
def _(histories, opt_trajs, grad_loss, cost, start, goal, passable, sel_log, g_ratio, max_iters, iters, t_batch, order=None):
    return torch.empty_like(cost)


class _AstarL1Loss(torch.autograd.Function):
    """search + L1 loss as ONE autograd node: forward = nastar_forward + nastar_l1_loss, backward = nastar_backward_l1_replay."""

    @staticmethod
    def forward(ctx, cost, start, goal, passable, opt_trajs, g_ratio, max_iters):
        with torch.no_grad():
            order = None
            if cost.shape[0] >= PLACEMENT_MIN_BATCH and workspace_bytes(cost.shape) == 0:
                order = new_placement_buffer(cost.shape[0], cost.device)  # the forward writes the order its searches finish in
                hist, paths, iters, status, sel_log = torch.ops.nastar.astar_forward_ordered(cost, start, goal, passable, g_ratio, max_iters,
                                                                                             True, 0, None, order)
            else:
                hist, paths, iters, status, sel_log = torch.ops.nastar.astar_forward(cost, start, goal, passable, g_ratio, max_iters, True)
            loss = torch.ops.nastar.l1_loss(hist, opt_trajs)
        ctx.order = order
        ctx.save_for_backward(cost, start, goal, passable, opt_trajs, hist, iters, sel_log)
        ctx.g_ratio, ctx.max_iters = g_ratio, max_iters
        ctx.mark_non_differentiable(hist, paths, iters, status)
        ctx.set_materialize_grads(False)  # otherwise autograd zero-fills a gradient for every unused output: 5 fill launches per step
        return loss.reshape(()), hist, paths, iters, status

    @staticmethod
    def backward(ctx, g_loss, g_hist, g_paths, g_iters, g_status):
        cost, start, goal, passable, opt_trajs, hist, iters, sel_log = ctx.saved_tensors
        if g_loss is None:
            return None, None, None, None, None, None, None
        grad_cost = torch.ops.nastar.astar_backward_l1_replay(hist, opt_trajs, g_loss, cost, start, goal, passable, sel_log,
                                                              ctx.g_ratio, ctx.max_iters, iters, BatchCoupling.t_batch(iters), ctx.order)
        return grad_cost, None, None, None, None, None, None


class _AstarForwardPlaced(torch.autograd.Function):
    """``astar_forward`` for large batches under autograd: the forward launch writes the order its searches finish in and the replay
    backward starts its workgroups in that order (same values as the registered autograd of ``astar_forward``; outputs other than
    ``histories`` carry no gradient).  ``order_in`` / ``order_out``: the forward's own placement (planner.Placement) or None."""

    @staticmethod
    def forward(ctx, cost, start, goal, passable, g_ratio, max_iters, flags, order_in, order_out):
        with torch.no_grad():
            if order_out is None:
                order_out = new_placement_buffer(cost.shape[0], cost.device)
            hist, paths, iters, status, sel_log = torch.ops.nastar.astar_forward_ordered(cost, start, goal, passable, g_ratio, max_iters, True,
                                                                                         flags, order_in, order_out)
        ctx.order = order_out
        ctx.save_for_backward(cost, start, goal, passable, iters, sel_log)
        ctx.g_ratio, ctx.max_iters = g_ratio, max_iters
        ctx.mark_non_differentiable(paths, iters, status, sel_log)
        ctx.set_materialize_grads(False)
        return hist, paths, iters, status, sel_log

    @staticmethod
    def backward(ctx, g_hist, g_paths, g_iters, g_status, g_log):
        if g_hist is None:
            return (None,) * 9
        cost, start, goal, passable, iters, sel_log = ctx.saved_tensors
        grad_cost = torch.ops.nastar.astar_backward_replay(g_hist.contiguous(), cost, start, goal, passable, sel_log, ctx.g_ratio, ctx.max_iters,
                                                           iters, BatchCoupling.t_batch(iters), ctx.order)
        return (grad_cost,) + (None,) * 8


def astar_forward_placed(cost, start, goal, passable, g_ratio: float, max_iters: int, flags: int = 0, order_in=None, order_out=None):
    """differentiable ``astar_forward`` (selection log kept) whose backward replays longest-first; see ``_AstarForwardPlaced``"""
    return _AstarForwardPlaced.apply(cost, start, goal, passable, float(g_ratio), int(max_iters), int(flags), order_in, order_out)


def astar_l1_loss(cost: torch.Tensor, start: torch.Tensor, goal: torch.Tensor, passable: torch.Tensor,
                  opt_trajs: torch.Tensor, g_ratio: float, max_iters: int):
    """[B,H,W] maps -> (loss scalar, histories, paths, iters, status); only ``loss`` carries gradient (to ``cost``)."""
    return _AstarL1Loss.apply(cost, start, goal, passable, opt_trajs, float(g_ratio), int(max_iters))


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
