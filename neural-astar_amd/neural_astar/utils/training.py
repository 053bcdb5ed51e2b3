"""Device-side training step (SURVEY.md section 8f "next #3").

The reference's ``PlannerModule.training_step`` (``utils/training.py:55-61``) is

    outputs = planner(map_designs, start_maps, goal_maps)
    loss = nn.L1Loss()(outputs.histories, opt_trajs)

followed by ``loss.backward()``: autograd materialises ``sign(histories - opt_trajs) / numel`` and hands it to the search's
backward.  That code keeps working unchanged against this package.  ``fused_l1_step`` is the same computation as ONE autograd
node: the loss is a fixed-order device reduction and the sign gradient is formed inside ``nastar_backward_l1_replay`` -- no gradient
tensor, three fewer elementwise launches per step (they are a visible fraction of a 100-map training batch).
"""
from __future__ import annotations

import random
import re
from glob import glob
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..planner.astar import VanillaAstar
from ..planner.differentiable_astar import AstarOutput, Placement, _refuse_late_repair
from .metrics import plan_with_vanilla, validation_metrics

try:  # the reference's trainer; optional here (not in the MI355X image)
    import pytorch_lightning as pl
    _ModuleBase = pl.LightningModule
except ImportError:  # pragma: no cover - depends on the environment
    pl = None
    _ModuleBase = nn.Module

__all__ = ["load_from_ptl_checkpoint", "PlannerModule", "set_global_seeds", "fused_l1_step", "validate_in_flight"]


def load_from_ptl_checkpoint(checkpoint_path: str) -> dict:
    """state_dict of the planner stored in the newest ``.ckpt`` under ``checkpoint_path`` (keys with the ``planner.`` prefix
    removed; same contract as reference utils/training.py:18-39).  Tensors are mapped to the CPU; ``load_state_dict`` moves them."""
    ckpt_file = sorted(glob(f"{checkpoint_path}/**/*.ckpt", recursive=True))[-1]
    print(f"load {ckpt_file}")
    state_dict = torch.load(ckpt_file, map_location="cpu", weights_only=False)["state_dict"]
    return {re.split("planner.", k)[-1]: v for k, v in state_dict.items() if "planner" in k}


def set_global_seeds(seed: int) -> None:
    """torch / numpy / random seeds (reference utils/training.py:90-106); numpy's drives ``MazeDataset``'s start sampling."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    np.random.seed(seed)
    random.seed(seed)


class PlannerModule(_ModuleBase):
    """Training / validation steps of the reference's Lightning module (utils/training.py:42-87), with the same attribute names
    (``planner``, ``vanilla_astar``, ``config``) and logged metric names.  A ``pytorch_lightning.LightningModule`` when that
    package is installed, otherwise a plain ``nn.Module`` whose ``log`` collects into ``self.logged`` (drive it with any loop).

    MI355X specifics: the training step is one fused autograd node (``fused_l1_step``); the validation step runs the planner's
    and the VanillaAstar search in ONE launch and reduces the three metrics on the device (``utils/metrics.py``)."""

    def __init__(self, planner, config):
        super().__init__()
        self.planner = planner
        self.vanilla_astar = VanillaAstar()
        self.config = config
        self.logged = {}
        # validation MAPS recur every epoch in the same order (reference scripts/train.py:43-50: unshuffled loader, utils/data.py:40-47) but the
        # loaders re-draw every start cell (utils/data.py:152-166): a batch that carries its loader's placement hint (DeviceMazeBatches) is
        # placed by that; one without a hint by the order the same batch index finished in last epoch (planner/differentiable_astar.py: Placement)
        self._val_placements = {}

    if pl is None:
        def log(self, name, value, *args, **kwargs):  # noqa: D401 - Lightning's signature
            self.logged[name] = value

    def forward(self, map_designs, start_maps, goal_maps):
        return self.planner(map_designs, start_maps, goal_maps)

    def configure_optimizers(self) -> torch.optim.Optimizer:
        from .optim import FusedRMSprop  # torch.optim.RMSprop with a one-launch step on the device (same state and update)
        return FusedRMSprop(self.planner.parameters(), self.config.params.lr)

    def training_step(self, train_batch, batch_idx):
        map_designs, start_maps, goal_maps, opt_trajs = train_batch
        loss, _ = fused_l1_step(self.planner, map_designs, start_maps, goal_maps, opt_trajs)
        self.log("metrics/train_loss", loss)
        return loss

    def validate(self, loader, streams: int = 4) -> dict:
        """a whole validation pass with the searches in flight: see ``validate_in_flight``"""
        return validate_in_flight(self, loader, streams)

    def validation_step(self, val_batch, batch_idx):
        map_designs, start_maps, goal_maps, opt_trajs = val_batch
        astar = getattr(self.planner, "astar", None)
        if astar is not None and hasattr(astar, "placement") and map_designs.is_cuda:
            astar.placement = self._val_placements.setdefault(int(batch_idx), Placement())
        if map_designs.shape[1] == 1 and hasattr(self.planner, "encode"):  # shortest-path problems (:72-85)
            outputs, va_outputs = plan_with_vanilla(self.planner, map_designs, start_maps, goal_maps)
            loss = nn.L1Loss()(outputs.histories, opt_trajs)
            self.log("metrics/val_loss", loss)
            m = validation_metrics(outputs, va_outputs)
            self.log("metrics/p_opt", m.p_opt)
            self.log("metrics/p_exp", m.p_exp)
            self.log("metrics/h_mean", m.h_mean)
            return loss
        outputs = self.forward(map_designs, start_maps, goal_maps)
        loss = nn.L1Loss()(outputs.histories, opt_trajs)
        self.log("metrics/val_loss", loss)
        return loss


def validate_in_flight(module: "PlannerModule", loader, streams: int = 4, window: int = 32) -> dict:
    """One validation pass over ``loader`` with the searches IN FLIGHT (``parallel.InFlightPlanner``): the mean of what
    ``PlannerModule.validation_step`` logs per batch (``metrics/val_loss`` and, for shortest-path problems, ``p_opt`` / ``p_exp`` /
    ``h_mean``; reference utils/training.py:63-87), each batch weighted by its SIZE -- Lightning's epoch-end reduction of ``self.log`` in a
    validation step weights by batch size, which matters when the last batch is smaller.  Batches are collected in windows of ``window``
    (inputs, outputs and status rows of at most that many launches are alive at a time).  Per batch the planner's
    and the VanillaAstar search are ONE launch, as in ``validation_step``; the launches of consecutive batches overlap (a 4096-map
    launch is as long as its longest search: 3-4 in flight sustain 2-3x the maps/s), the encoder runs batch after batch on the current
    stream, the metrics are reduced on the device when everything has been collected.  Falls back to ``validation_step`` per batch for
    planners the pair launch does not cover."""
    from ..parallel import InFlightPlanner
    planner = module.planner
    pair_ok = (hasattr(planner, "encode") and not planner.training and float(planner.g_ratio) == 0.5)
    sums: dict = {}
    n = 0
    if not pair_ok:
        for i, batch in enumerate(loader):
            module.logged = {}
            module.validation_step(batch, i)
            w = batch[0].shape[0]
            for k, v in module.logged.items():
                sums[k] = sums.get(k, 0.0) + v.detach().double() * w
            n += w
        return {k: v / max(n, 1) for k, v in sums.items()}
    fly = InFlightPlanner(planner, streams=streams, unit_cost=False)
    kept = []

    def drain():
        nonlocal n
        outs = fly.collect()
        for out, (opt_trajs, shortest, B) in zip(outs, kept):
            o = AstarOutput(out.histories[:B], out.paths[:B], [])
            vals = {"metrics/val_loss": nn.L1Loss()(o.histories, opt_trajs)}
            if shortest:
                m = validation_metrics(o, AstarOutput(out.histories[B:], out.paths[B:], []))
                vals.update({"metrics/p_opt": m.p_opt, "metrics/p_exp": m.p_exp, "metrics/h_mean": m.h_mean})
            for k, v in vals.items():
                sums[k] = sums.get(k, 0.0) + v.double() * B
            n += B
        kept.clear()

    with torch.no_grad():
        for map_designs, start_maps, goal_maps, opt_trajs in loader:
            shortest = map_designs.shape[1] == 1
            cost = planner.encode(map_designs, start_maps, goal_maps)
            passable = map_designs if not planner.learn_obstacles else torch.ones_like(start_maps)
            if shortest:  # the planner's problem and the VanillaAstar problem stacked along the batch dimension (maps are independent)
                fly.submit_search(torch.cat((cost, map_designs), 0), torch.cat((start_maps, start_maps), 0), torch.cat((goal_maps, goal_maps), 0),
                                  torch.cat((passable, map_designs), 0))
            else:
                fly.submit_search(cost, start_maps, goal_maps, passable)
            kept.append((opt_trajs, shortest, map_designs.shape[0]))
            if len(kept) >= window:
                drain()
        drain()
    res = {k: v / max(n, 1) for k, v in sums.items()}
    for k, v in res.items():
        module.log(k, v)
    return res


def fused_l1_step(planner: VanillaAstar, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                  opt_trajs: torch.Tensor) -> Tuple[torch.Tensor, AstarOutput]:
    """``(loss, outputs)`` with ``loss == nn.L1Loss()(outputs.histories, opt_trajs)``; ``loss.backward()`` reaches the encoder.

    Works for ``VanillaAstar`` and ``NeuralAstar`` (whose ``encode`` supplies the cost map); search budget and mode follow
    ``planner.astar`` exactly as in ``planner.forward`` (``Tmax`` applies in training mode only)."""
    # The fused node computes mean|histories - opt_trajs| for ONE trajectory per map.  The reference's L1Loss broadcasts
    # histories [B,1,H,W] against opt_trajs [B,S,H,W] when num_starts S > 1 (training.py:58), and use_differentiable_astar=False
    # selects another planner altogether: both go through the planner's own forward + nn.L1Loss, exactly like the reference.
    if opt_trajs.shape[1] != 1 or opt_trajs.shape[-2:] != start_maps.shape[-2:] or not getattr(planner, "use_differentiable_astar", True):
        outputs = planner(map_designs, start_maps, goal_maps)
        return nn.L1Loss()(outputs.histories, opt_trajs), outputs
    if hasattr(planner, "encode"):
        cost_maps = planner.encode(map_designs, start_maps, goal_maps)
        obstacles = map_designs if not planner.learn_obstacles else torch.ones_like(start_maps)
    else:
        cost_maps, obstacles = map_designs, map_designs
    astar = planner.astar
    if not (cost_maps.is_cuda and torch.cuda.is_current_stream_capturing()):
        astar.raise_if_unsolvable(wait=False)  # deferred verdicts of earlier steps that have reached the host
    W = cost_maps.shape[-1]
    max_iters = ops.max_iters_for(W, astar.Tmax, astar.training)
    # a batch assembled by DeviceMazeBatches carries a placement (start_maps.placement_order: by the optimal distance of its start cells)
    # (read-only use: this step records no completion order, so a Placement the caller attached stays attached for the call it was meant for -- ADVICE r5)
    pl_keep = astar.placement
    order, _, check, _ = astar.resolve_placement(cost_maps.shape[0], start_maps, ops.workspace_bytes(cost_maps.shape) == 0)
    astar.placement = pl_keep
    row = astar.begin_launch(cost_maps)
    # batch semantics as in DifferentiableAstar.forward: outside g_ratio in [0.5, 1) the exact pipeline runs with the launch; inside, the same-call
    # verdict re-runs a batch that reports the batch-coupled note (negative costs), a deferred one refuses (the graph is built by then)
    exact = cost_maps.shape[0] > 1 and ops.coupling_possible(astar.g_ratio)
    args = (cost_maps[:, 0], start_maps[:, 0], goal_maps[:, 0], obstacles[:, 0], opt_trajs[:, 0], astar.g_ratio, max_iters)
    try:
        loss, hist, paths, iters, status = ops.astar_l1_loss(*args, order, check, astar.summary_ptr(row, cost_maps), exact)
    except BaseException:
        if row >= 0:
            ops.StatusBoard.of(cost_maps.device).release(row)
        raise
    repair = _refuse_late_repair if (astar.check_solvable == "deferred" and not exact and cost_maps.shape[0] > 1) else None
    # same contract as DifferentiableAstar.forward (default: raises in THIS call, before any backward)
    if astar.note_status(status, iters, row=row, repair=repair) and not exact:
        loss, hist, paths, iters, status = ops.astar_l1_loss(*args, None, False, 0, True)
        astar.last_status, astar.last_iters = status, iters
    return loss, AstarOutput(hist.unsqueeze(1), paths.unsqueeze(1), [])
