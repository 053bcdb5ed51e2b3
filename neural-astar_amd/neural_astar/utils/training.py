"""Device-side training step (SURVEY.md section 8f "next #3").

The reference's ``PlannerModule.training_step`` (``utils/training.py:55-61``) is

    outputs = planner(map_designs, start_maps, goal_maps)
    loss = nn.L1Loss()(outputs.histories, opt_trajs)

followed by ``loss.backward()``: autograd materialises ``sign(histories - opt_trajs) / numel`` and hands it to the search's
backward.  That code keeps working unchanged against this package.  ``fused_l1_step`` is the same computation as ONE autograd
node: the loss is a fixed-order device reduction and the sign gradient is formed inside ``nastar_backward_l1`` -- no gradient
tensor, three fewer elementwise launches per step (they are a visible fraction of a 100-map training batch).
"""
from __future__ import annotations

from typing import Tuple

import torch

from .. import ops
from ..planner.astar import VanillaAstar
from ..planner.differentiable_astar import AstarOutput


def fused_l1_step(planner: VanillaAstar, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                  opt_trajs: torch.Tensor) -> Tuple[torch.Tensor, AstarOutput]:
    """``(loss, outputs)`` with ``loss == nn.L1Loss()(outputs.histories, opt_trajs)``; ``loss.backward()`` reaches the encoder.

    Works for ``VanillaAstar`` and ``NeuralAstar`` (whose ``encode`` supplies the cost map); search budget and mode follow
    ``planner.astar`` exactly as in ``planner.forward`` (``Tmax`` applies in training mode only)."""
    if hasattr(planner, "encode"):
        cost_maps = planner.encode(map_designs, start_maps, goal_maps)
        obstacles = map_designs if not planner.learn_obstacles else torch.ones_like(start_maps)
    else:
        cost_maps, obstacles = map_designs, map_designs
    astar = planner.astar
    W = cost_maps.shape[-1]
    max_iters = ops.max_iters_for(W, astar.Tmax, astar.training)
    loss, hist, paths, iters, status = ops.astar_l1_loss(cost_maps[:, 0], start_maps[:, 0], goal_maps[:, 0], obstacles[:, 0],
                                                         opt_trajs[:, 0], astar.g_ratio, max_iters)
    astar.last_status, astar.last_iters = status, iters
    return loss, AstarOutput(hist.unsqueeze(1), paths.unsqueeze(1), [])
