"""``VanillaAstar`` / ``NeuralAstar`` -- the drop-in boundary (reference ``planner/astar.py``).

Same constructors, attributes (``astar``, ``encoder``, ``g_ratio``, ``encode``, ``perform_astar``) and state-dict
keys (``astar.neighbor_filter``, ``encoder.model.<n>.*``) as the reference (astar.py:17-46,105-152), so
``scripts/train.py`` and ``utils/training.py`` run unchanged against this package.

Not drop-in (by design): ``use_differentiable_astar=False`` (the reference's CPU ``pq_astar``, a
different algorithm; the reference's own ``tests/astar_test.py::test_pq_astar`` therefore needs the reference package) raises;
CPU tensors raise (no CPU fallback), so ``scripts/create_gif.py``, which never moves the planner or its data to a device, needs
a one-line ``.cuda()`` on both to run.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import encoder
from .differentiable_astar import AstarOutput, DifferentiableAstar
from .pq_astar import pq_astar  # noqa: F401  (the reference's astar.py imports it here: a stub that fails loudly when called)


def _is_depth4_cnn(enc: nn.Module) -> bool:
    """The HIP encoder kernels implement the reference's default depth-4 CNN (.. -> 32 -> 64 -> 128 -> 256 -> 1) only; any other
    depth keeps the torch encoder."""
    convs = [m for m in enc.model if isinstance(m, nn.Conv2d)]
    return [c.out_channels for c in convs] == [32, 64, 128, 256, 1] and convs[0].in_channels <= 16


def _downsize_cls():
    from ..encoder_hip import HipCnnDownSizeEncoder
    return HipCnnDownSizeEncoder


class VanillaAstar(nn.Module):
    def __init__(self, g_ratio: float = 0.5, use_differentiable_astar: bool = True):
        """Vanilla A*: cost map = obstacle map = ``map_designs`` (reference astar.py:17-46,93-94).

        Examples:
            >>> planner = VanillaAstar().cuda()
            >>> outputs = planner(map_designs, start_maps, goal_maps)
            >>> outputs.histories, outputs.paths
        """
        super().__init__()
        self.astar = DifferentiableAstar(g_ratio=g_ratio, Tmax=1.0)
        self.g_ratio = g_ratio
        self.use_differentiable_astar = use_differentiable_astar

    def perform_astar(self, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                      obstacles_maps: torch.Tensor, store_intermediate_results: bool = False) -> AstarOutput:
        if not self.use_differentiable_astar:
            # reference astar.py:57-61 dispatches to pq_astar (CPU numpy + pqdict): a different, non-differentiable
            # algorithm (it charges the NEIGHBOUR's cost, pq_astar.py:138-144) that is outside the hot path.
            raise NotImplementedError(
                "use_differentiable_astar=False selects the reference's CPU-only pq_astar, which is out of scope for "
                "the MI355X-native hot path; the HIP DifferentiableAstar kernel has no large-map penalty, use it.")
        return self.astar(map_designs, start_maps, goal_maps, obstacles_maps, store_intermediate_results)

    def forward(self, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                store_intermediate_results: bool = False) -> AstarOutput:
        return self.perform_astar(map_designs, start_maps, goal_maps, map_designs, store_intermediate_results)


class NeuralAstar(VanillaAstar):
    def __init__(self, g_ratio: float = 0.5, Tmax: float = 1.0, encoder_input: str = "m+",
                 encoder_arch: str = "CNN", encoder_depth: int = 4, learn_obstacles: bool = False,
                 const: float = None, use_differentiable_astar: bool = True):
        """Neural A*: an encoder predicts the cost map, then the same search runs (reference astar.py:105-152).

        Args mirror the reference: ``encoder_input`` "m+" = map + (start+goal) channel, "m" = map only;
        ``encoder_arch`` in {"CNN", "CNNDownSize", "Unet"}; ``learn_obstacles`` hides the obstacle map from the search;
        ``const`` = learnable scale on the predicted cost.
        """
        super().__init__()
        self.astar = DifferentiableAstar(g_ratio=g_ratio, Tmax=Tmax)
        self.encoder_input = encoder_input
        self.encoder = getattr(encoder, encoder_arch)(len(self.encoder_input), encoder_depth, const)
        self.learn_obstacles = learn_obstacles
        if self.learn_obstacles:
            print("WARNING: learn_obstacles has been set to True")
        self.g_ratio = g_ratio
        self.use_differentiable_astar = use_differentiable_astar
        # Which kernels predict the cost map.  "auto" (default): on a HIP device "hip_f16x3", on CPU tensors "torch" -- decided per
        # call from the input's device, before any BatchNorm state is touched, so that the reference's scripts/train.py:30-50
        # (which constructs NeuralAstar(...) with no such knob) trains and validates on the MFMA kernels unmodified.  "torch" = fp32
        # torch.nn (MIOpen), "hip_bf16" (bf16-MFMA inference kernels for the depth-4 CNN encoder, csrc/nastar_encoder.hip.h),
        # "hip_f16" (plain fp16 operands) or "hip_f16x3" (split fp16 operands: 3x the matrix work, cost maps within 1e-5 of the fp32
        # encoder -- the only hip_* mode that meets the reference's float tolerance, hence what "auto" picks).  With a hip_* backend:
        # eval mode under no_grad = the inference kernels (CNN of any depth / size, CNNDownSize, Unet); training mode with autograd on
        # = the training kernels of neural_astar/encoder_train.py (forward, input and weight gradients, batch-statistics BatchNorm,
        # max-pool, upsample-concat) for CNN of any depth / map size, CNNDownSize and the VggUnet definition of Unet; shapes those
        # kernels do not take, and eval mode with gradients, stay on torch.nn.  Not part of the reference's constructor.
        # "hip_strict" = "hip_f16x3" that RAISES instead of falling back to torch.nn for a shape / mode the kernels do not take.
        self.encoder_backend = "auto"
        self._hip_encoder = None
        # which code predicted the latest cost map: "hip:<kernel family>/<precision>" or "torch.nn" (+ why); a fall-back to torch.nn on a HIP
        # device under a hip_* backend also warns once per reason (a shape that misses must not train on MIOpen without a word)
        self.last_encoder_route: str = ""
        self._route_warned: set = set()

    def effective_encoder_backend(self, like: torch.Tensor) -> str:
        """``encoder_backend`` with "auto" resolved for a tensor on ``like``'s device."""
        if self.encoder_backend == "auto":
            return "hip_f16x3" if like.is_cuda else "torch"
        if self.encoder_backend == "hip_strict":
            return "hip_f16x3"
        return self.encoder_backend

    def _routed(self, route: str, cost: torch.Tensor) -> torch.Tensor:
        self.last_encoder_route = route
        return cost

    def encode(self, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor) -> torch.Tensor:
        """Predict cost maps (reference astar.py:154-180)."""
        backend = self.effective_encoder_backend(map_designs)
        if (backend.startswith("hip") and not self.training and not torch.is_grad_enabled()
                and isinstance(self.encoder, encoder.CNNDownSize)):
            # CNNDownSize (WarCraft): f32-input MFMA kernels at fp32 accuracy whatever the hip_* precision asked for
            convs = [m for m in self.encoder.model if isinstance(m, nn.Conv2d)]
            depth = len(convs) - 1
            plus = "+" in self.encoder_input
            if (1 <= depth <= 4 and [c.out_channels for c in convs] == [32, 64, 128, 256][:depth] + [1] and convs[0].in_channels <= 4
                    and map_designs.shape[1] + int(plus) == convs[0].in_channels
                    and map_designs.shape[-2] % (1 << depth) == 0 and map_designs.shape[-1] % (1 << depth) == 0):
                if not isinstance(self._hip_encoder, _downsize_cls()):
                    self._hip_encoder = _downsize_cls()(self.encoder)
                return self._routed("hip:CNNDownSize-infer/f32", self._hip_encoder(map_designs, start_maps, goal_maps, plus))
        if (backend.startswith("hip") and not self.training and not torch.is_grad_enabled()
                and isinstance(self.encoder, encoder.Unet) and isinstance(self.encoder.model, encoder.VggUnet)
                and map_designs.shape[1] == 1 and map_designs.shape[-2:] == start_maps.shape[-2:]
                and map_designs.shape[-2] % (1 << self.encoder.model.depth) == 0
                and map_designs.shape[-1] % (1 << self.encoder.model.depth) == 0):
            # Unet(vgg16_bn): generic fp16 MFMA convolution (csrc/nastar_conv_flat.hip.h); "hip_f16x3" = split operands, fp32-grade
            precision = "f16x3" if backend == "hip_f16x3" else "f16"
            from ..encoder_hip import HipUnetEncoder
            if type(self._hip_encoder) is not HipUnetEncoder or self._hip_encoder.precision != precision:
                self._hip_encoder = HipUnetEncoder(self.encoder, precision)
            return self._routed(f"hip:Unet-infer/{precision}", self._hip_encoder(map_designs, start_maps, goal_maps, "+" in self.encoder_input))
        # (training mode under no_grad -- a validation pass somebody forgot to switch to eval() -- is the same forward: batch statistics, running
        #  statistics updated; the autograd functions then simply record nothing)
        if (backend.startswith("hip") and map_designs.is_cuda
                and (self.encoder.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.encoder.parameters())))
                and isinstance(self.encoder, encoder.Unet) and map_designs.shape[1] == 1
                and map_designs.shape[-2:] == start_maps.shape[-2:]):
            from ..encoder_train import unet_supported, unet_train_forward
            first = next(m for m in self.encoder.model.modules() if isinstance(m, nn.Conv2d))
            if (unet_supported(self.encoder, map_designs.shape[-2], map_designs.shape[-1])
                    and first.in_channels == 1 + int("+" in self.encoder_input)):
                prec = "f16" if backend == "hip_f16" else "f16x3"
                return self._routed(f"hip:Unet-{'train' if self.encoder.training else 'evalgrad'}/{prec}", unet_train_forward(self.encoder, map_designs, start_maps, goal_maps,
                                                                                 "+" in self.encoder_input, prec))
        if (backend.startswith("hip") and map_designs.is_cuda and isinstance(self.encoder, encoder.CNN)
                and (self.encoder.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.encoder.parameters())))):
            # UNDER AUTOGRAD: convolutions, BatchNorm (batch statistics in training mode; the running statistics in eval mode -- round 6),
            # ReLU, max-pool and all their gradients on the MI355X kernels (neural_astar/encoder_train.py); "hip_f16" = plain fp16
            # operands, anything else = split operands (fp32-grade).  CNN (any depth) and CNNDownSize (WarCraft); shapes the kernels do
            # not take stay on torch.nn.
            from ..encoder_train import cnn_train_forward, supported
            convs = [m for m in self.encoder.model if isinstance(m, nn.Conv2d)]
            pool = isinstance(self.encoder, encoder.CNNDownSize)
            plus = "+" in self.encoder_input
            if (supported(self.encoder, map_designs.shape[-2], map_designs.shape[-1])
                    and map_designs.shape[1] + int(plus) == convs[0].in_channels
                    and (pool or map_designs.shape[-2:] == start_maps.shape[-2:])):
                prec = "f16" if backend == "hip_f16" else "f16x3"
                kind = "train" if self.encoder.training else "evalgrad"
                return self._routed(f"hip:{'CNNDownSize' if pool else 'CNN'}-{kind}/{prec}",
                                    cnn_train_forward(self.encoder, map_designs, start_maps, goal_maps, plus, prec))
        tile = 32 if backend in ("hip_f16", "hip_f16x3") else 16
        if (backend in ("hip_bf16", "hip_f16", "hip_f16x3") and not self.training and not torch.is_grad_enabled()
                and map_designs.shape[1] == 1 and map_designs.shape[-2:] == start_maps.shape[-2:]
                and map_designs.shape[-2] % tile == 0 and map_designs.shape[-1] % 32 == 0
                and isinstance(self.encoder, encoder.CNN) and not isinstance(self.encoder, encoder.CNNDownSize)
                and _is_depth4_cnn(self.encoder)):
            precision = backend[4:]
            from ..encoder_hip import HipCnnEncoder
            if type(self._hip_encoder) is not HipCnnEncoder or self._hip_encoder.precision != precision:
                self._hip_encoder = HipCnnEncoder(self.encoder, precision)
            return self._routed(f"hip:CNN-infer-img32/{precision}", self._hip_encoder(map_designs, start_maps, goal_maps, "+" in self.encoder_input))
        if (backend.startswith("hip") and not self.training and not torch.is_grad_enabled()
                and type(self.encoder) is encoder.CNN and map_designs.shape[1] == 1 and map_designs.shape[-2:] == start_maps.shape[-2:]
                and self.encoder.model[0].in_channels == 1 + int("+" in self.encoder_input)):
            # any other depth / map size: the generic fp16 MFMA convolution ("hip_f16x3" = split operands, otherwise plain fp16)
            precision = "f16x3" if backend == "hip_f16x3" else "f16"
            from ..encoder_hip import HipFlatCnnEncoder
            if type(self._hip_encoder) is not HipFlatCnnEncoder or self._hip_encoder.precision != precision:
                self._hip_encoder = HipFlatCnnEncoder(self.encoder, precision)
            return self._routed(f"hip:CNN-infer-flat/{precision}", self._hip_encoder(map_designs, start_maps, goal_maps, "+" in self.encoder_input))
        # torch.nn (MIOpen on a HIP device): the CPU path, the "torch" backend -- or a shape / mode the kernels do not take
        route = "torch.nn"
        if backend.startswith("hip"):
            why = (f"{type(self.encoder).__name__} on {tuple(map_designs.shape)} maps, training={self.encoder.training}, "
                   f"grad={torch.is_grad_enabled()}, device={map_designs.device.type}")
            route = f"torch.nn (fell through from {backend}: {why})"
            if self.encoder_backend == "hip_strict":
                raise RuntimeError(f"encoder_backend='hip_strict': no MI355X encoder kernel takes {why}; it would run on torch.nn")
            if map_designs.is_cuda and why not in self._route_warned:
                self._route_warned.add(why)
                import warnings
                warnings.warn(f"NeuralAstar.encode: {why} is not covered by the MI355X encoder kernels -- this call runs on torch.nn (MIOpen); "
                              "planner.last_encoder_route records the route, encoder_backend='hip_strict' turns this into an error, "
                              "'torch' silences it", RuntimeWarning, stacklevel=2)
        self.last_encoder_route = route
        inputs = map_designs
        if "+" in self.encoder_input:
            sg = start_maps + goal_maps
            if map_designs.shape[-1] != start_maps.shape[-1]:  # WarCraft: 96x96 image vs 12x12 grid
                sg = nn.functional.interpolate(sg, size=map_designs.shape[-2:], mode="nearest")
            inputs = torch.cat((inputs, sg), dim=1)
        return self.encoder(inputs)

    def forward(self, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor,
                store_intermediate_results: bool = False) -> AstarOutput:
        cost_maps = self.encode(map_designs, start_maps, goal_maps)
        obstacles_maps = map_designs if not self.learn_obstacles else torch.ones_like(start_maps)
        return self.perform_astar(cost_maps, start_maps, goal_maps, obstacles_maps, store_intermediate_results)
