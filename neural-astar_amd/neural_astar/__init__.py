"""MI355X-native drop-in for the ``neural_astar`` package of omron-sinicx/neural-astar.

Only the hot path is re-implemented: ``neural_astar.planner.{VanillaAstar, NeuralAstar}`` ->
``DifferentiableAstar.forward`` runs as one hand-written HIP kernel launch on gfx950, its backward as a second one, and -- opt-in via
``planner.encoder_backend = "hip_*"`` -- the cost-map encoders (CNN, CNNDownSize, Unet) run inference and training on MFMA kernels
(see DESIGN.md).
"""
__version__ = "0.3.0"
