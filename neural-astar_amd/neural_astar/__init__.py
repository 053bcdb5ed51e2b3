"""MI355X-native drop-in for the ``neural_astar`` package of omron-sinicx/neural-astar.

Only the hot path is re-implemented: ``neural_astar.planner.{VanillaAstar, NeuralAstar}`` ->
``DifferentiableAstar.forward`` runs as one hand-written HIP kernel launch on gfx950.
"""
__version__ = "0.1.0"
