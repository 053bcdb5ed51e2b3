"""MI355X-native drop-in for the ``neural_astar`` package of omron-sinicx/neural-astar.

Only the hot path is re-implemented: ``neural_astar.planner.{VanillaAstar, NeuralAstar}`` ->
``DifferentiableAstar.forward`` runs as one hand-written HIP kernel launch on gfx950, its backward as a second one, and the cost-map
encoders (CNN, CNNDownSize, Unet) run inference and training on MFMA kernels (``planner.encoder_backend``: "auto" = "hip_f16x3" on a
HIP device; "torch" keeps torch.nn).
"""
__version__ = "0.6.0"  # == NASTAR_VERSION 600 of include/nastar.h (tests/test_capi_library.py checks the pair)
