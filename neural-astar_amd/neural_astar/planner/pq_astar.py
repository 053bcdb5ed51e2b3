"""Placeholder for the reference's ``planner/pq_astar.py`` (CPU numpy + ``pqdict``: a DIFFERENT, non-differentiable algorithm -- it charges the
NEIGHBOUR's cost, pq_astar.py:138-144 -- that the reference recommends for large maps, astar.py:36-37).  Out of scope for the MI355X hot path
(SURVEY.md section 2 #4): the HIP ``DifferentiableAstar`` takes maps up to 1024x1024 and has no large-map penalty.  The module exists so that
``from neural_astar.planner.pq_astar import pq_astar`` imports; calling it fails loudly."""
import numpy as np


def pq_astar(pred_costs: np.ndarray, start_maps: np.ndarray, goal_maps: np.ndarray, map_designs: np.ndarray,
             store_intermediate_results: bool = False, g_ratio: float = 0.5):
    raise NotImplementedError(
        "pq_astar is the reference's CPU-only priority-queue A* (numpy + pqdict), which is out of scope for the MI355X-native hot path; "
        "use VanillaAstar / NeuralAstar with use_differentiable_astar=True (the default): the HIP search kernels take maps up to 1024x1024")
