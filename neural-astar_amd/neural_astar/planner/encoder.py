"""Cost-map encoders (reference ``planner/encoder.py``).

NOT part of the hot path (SURVEY.md section 8f "next #1"): plain torch.nn modules whose only job here is to keep
``NeuralAstar`` constructible and checkpoint-compatible (state-dict keys ``encoder.model.<n>.*`` of the shipped
``mazes_032_moore_c8`` checkpoint load strictly).  The convolutions run on whatever MIOpen/hipBLASLt give.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class EncoderBase(nn.Module):
    def __init__(self, input_dim: int, encoder_depth: int = 4, const: float = None):
        super().__init__()
        self.model = self.construct_encoder(input_dim, encoder_depth)
        # learnable scale on the sigmoid output, or the constant 1.0 (reference encoder.py:24-27)
        self.const = nn.Parameter(torch.ones(1) * const) if const is not None else 1.0

    def construct_encoder(self, input_dim: int, encoder_depth: int) -> nn.Module:
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sigmoid(self.model(x)) * self.const


def _conv_stack(widths, pool: bool) -> nn.Sequential:
    layers = []
    last = len(widths) - 2
    for k, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
        layers += [nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(cout)]
        if k == last:
            break  # final 1-channel block keeps its BatchNorm but has no ReLU / pooling (reference encoder.py:78,97)
        layers.append(nn.ReLU())
        if pool:
            layers.append(nn.MaxPool2d((2, 2)))
    return nn.Sequential(*layers)


class CNN(EncoderBase):
    """input -> 32 -> 64 -> 128 -> 256 -> 1 channels of 3x3 convs with BN+ReLU (reference encoder.py:60-78)."""

    CHANNELS = [32, 64, 128, 256]

    def construct_encoder(self, input_dim: int, encoder_depth: int) -> nn.Module:
        return _conv_stack([input_dim] + self.CHANNELS[:encoder_depth] + [1], pool=False)


class CNNDownSize(CNN):
    """Same stack with a 2x2 max-pool after every hidden block (reference encoder.py:81-97)."""

    def construct_encoder(self, input_dim: int, encoder_depth: int) -> nn.Module:
        return _conv_stack([input_dim] + self.CHANNELS[:encoder_depth] + [1], pool=True)


class Unet(EncoderBase):
    """``segmentation_models_pytorch`` U-Net with a vgg16_bn backbone (reference encoder.py:37-57).

    The third-party package is not vendored by the reference and is absent here: parity unpinned."""

    DECODER_CHANNELS = [256, 128, 64, 32, 16]

    def construct_encoder(self, input_dim: int, encoder_depth: int) -> nn.Module:
        try:
            import segmentation_models_pytorch as smp
        except ImportError as e:  # pragma: no cover
            raise ImportError("encoder_arch='Unet' needs segmentation_models_pytorch (not installed)") from e
        return smp.Unet(encoder_name="vgg16_bn", encoder_weights=None, classes=1, in_channels=input_dim,
                        encoder_depth=encoder_depth, decoder_channels=self.DECODER_CHANNELS[:encoder_depth])
