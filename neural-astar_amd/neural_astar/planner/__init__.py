from .astar import NeuralAstar, VanillaAstar  # noqa: F401  (same export list as the reference's planner/__init__.py:1)
