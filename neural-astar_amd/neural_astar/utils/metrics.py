"""Validation metrics of the reference's training harness, kept on the device (SURVEY.md section 8f "next #2").

``PlannerModule.validation_step`` (reference ``utils/training.py:63-87``) runs the hot path TWICE on the same maps --
the learned planner and a ``VanillaAstar`` -- and then reduces ``paths`` / ``histories`` to per-map sums on the host with
numpy.  Here both searches go out as ONE launch (the two cost maps are stacked along the batch dimension; maps are
independent) and the three metrics are computed with device reductions, so a validation step needs one kernel launch
and no host round trip per metric.
"""
from __future__ import annotations

from typing import NamedTuple, Tuple

import torch

from ..planner.astar import NeuralAstar, VanillaAstar
from ..planner.differentiable_astar import AstarOutput


class ValidationMetrics(NamedTuple):
    p_opt: torch.Tensor   # fraction of maps whose path is as short as VanillaAstar's         (training.py:73-75)
    p_exp: torch.Tensor   # mean relative reduction of node expansions, clipped at 0           (training.py:77-79)
    h_mean: torch.Tensor  # harmonic mean of the two                                          (training.py:81)


def validation_metrics(planner_out: AstarOutput, vanilla_out: AstarOutput) -> ValidationMetrics:
    pathlen_model = planner_out.paths.sum((1, 2, 3))
    pathlen_astar = vanilla_out.paths.sum((1, 2, 3))
    p_opt = (pathlen_astar == pathlen_model).double().mean()
    exp_astar = vanilla_out.histories.detach().sum((1, 2, 3)).double()
    exp_na = planner_out.histories.detach().sum((1, 2, 3)).double()
    p_exp = torch.clamp((exp_astar - exp_na) / exp_astar, min=0.0).mean()
    h_mean = 2.0 / (1.0 / (p_opt + 1e-10) + 1.0 / (p_exp + 1e-10))
    return ValidationMetrics(p_opt, p_exp, h_mean)


def plan_with_vanilla(planner: NeuralAstar, map_designs: torch.Tensor, start_maps: torch.Tensor,
                      goal_maps: torch.Tensor, g_ratio_vanilla: float = 0.5) -> Tuple[AstarOutput, AstarOutput]:
    """Learned planner + VanillaAstar on the same problems in ONE search launch (eval-mode budgets, no gradients).

    Requires ``planner.g_ratio == g_ratio_vanilla`` (the reference's ``PlannerModule`` builds ``VanillaAstar()`` with the
    default 0.5, training.py:46) because one launch has one ``g_ratio``; otherwise it falls back to two launches."""
    with torch.no_grad():
        cost = planner.encode(map_designs, start_maps, goal_maps)
        passable = map_designs if not planner.learn_obstacles else torch.ones_like(start_maps)
        if float(planner.g_ratio) != float(g_ratio_vanilla) or planner.training:
            va = VanillaAstar(g_ratio=g_ratio_vanilla).to(map_designs.device).eval()
            return planner.perform_astar(cost, start_maps, goal_maps, passable), va(map_designs, start_maps, goal_maps)
        B = map_designs.shape[0]
        both = planner.astar(torch.cat((cost, map_designs[:, :1]), 0), torch.cat((start_maps, start_maps), 0),
                             torch.cat((goal_maps, goal_maps), 0), torch.cat((passable, map_designs[:, :1]), 0))
        return (AstarOutput(both.histories[:B], both.paths[:B], []), AstarOutput(both.histories[B:], both.paths[B:], []))
