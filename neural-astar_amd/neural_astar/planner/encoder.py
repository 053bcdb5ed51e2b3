"""Cost-map encoders (reference ``planner/encoder.py``).

Plain torch.nn modules that keep ``NeuralAstar`` constructible and checkpoint-compatible (state-dict keys
``encoder.model.<n>.*`` of the shipped ``mazes_032_moore_c8`` checkpoint load strictly) and hold the parameters.  With the
``encoder_backend = "torch"`` (and on CPU tensors) their convolutions run on torch.nn / MIOpen; with a ``hip_*`` backend (the default "auto" = ``hip_f16x3`` on a HIP device) ``NeuralAstar.encode`` runs the
same parameters through this package's MFMA kernels instead (``encoder_hip.py`` for inference, ``encoder_train.py`` for
training: SURVEY.md section 8f "next #1").
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


def _conv_bn_relu(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _UnetDecoderBlock(nn.Module):
    """upsample x2 (nearest) -> concat skip -> 2 x [conv3x3, BN, ReLU]"""

    def __init__(self, cin: int, cskip: int, cout: int):
        super().__init__()
        self.conv1 = _conv_bn_relu(cin + cskip, cout)
        self.conv2 = _conv_bn_relu(cout, cout)

    def forward(self, x, skip=None):
        x = nn.functional.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class _VggUnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]  # the full-resolution feature map is not used as a skip; deepest first
        head = enc[0]
        cin = [head] + list(decoder_channels[:-1])
        cskip = enc[1:] + [0]
        self.center = nn.Sequential(_conv_bn_relu(head, head), _conv_bn_relu(head, head))  # vgg encoders get a centre block
        self.blocks = nn.ModuleList([_UnetDecoderBlock(a, b, c) for a, b, c in zip(cin, cskip, decoder_channels)])

    def forward(self, *features):
        feats = list(features[1:])[::-1]
        x = self.center(feats[0])
        skips = feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class VggUnet(nn.Module):
    """From-scratch ``Unet(encoder_name="vgg16_bn")`` with the structure the reference asks ``segmentation_models_pytorch`` 0.3.1
    for (encoder.py:37-57): VGG16-BN feature stages split at the max-pools (64, 128, 256, 512, 512 channels at strides 1..16),
    a centre block, ``encoder_depth`` decoder blocks (nearest x2 upsampling, skip concat, two conv-BN-ReLU) with
    ``decoder_channels`` outputs, and a 3x3 segmentation head.  Module names mirror smp's (``encoder.features``, ``decoder.center``,
    ``decoder.blocks.<i>.conv1/conv2``, ``segmentation_head.0``).  The third-party package is absent from the reference tree and
    from this image: **parity unpinned** -- nothing here can be checked against smp itself."""

    VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]

    def __init__(self, in_channels: int, encoder_depth: int, decoder_channels):
        super().__init__()
        assert 1 <= encoder_depth <= 5 and len(decoder_channels) == encoder_depth
        layers, c = [], in_channels
        for v in self.VGG16:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.encoder = nn.Module()
        self.encoder.features = nn.Sequential(*layers)
        self.depth = encoder_depth
        out_channels = (64, 128, 256, 512, 512, 512)[: encoder_depth + 1]
        self.decoder = _VggUnetDecoder(out_channels, list(decoder_channels))
        self.segmentation_head = nn.Sequential(nn.Conv2d(decoder_channels[-1], 1, kernel_size=3, padding=1))

    def _stages(self):
        stages, cur = [], []
        for m in self.encoder.features:
            if isinstance(m, nn.MaxPool2d):
                stages.append(cur)
                cur = []
            cur.append(m)
        stages.append(cur)
        return stages

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        feats = []
        for stage in self._stages()[: self.depth + 1]:
            for m in stage:
                x = m(x)
            feats.append(x)
        return self.segmentation_head(self.decoder(*feats))


class Unet(EncoderBase):
    """U-Net with a vgg16_bn backbone (reference encoder.py:37-57).  Uses ``segmentation_models_pytorch`` when it is installed
    (the reference's own dependency); otherwise the from-scratch :class:`VggUnet` of the same structure (parity unpinned).
    The ``VggUnet`` form has MFMA inference and training paths (``encoder_hip.HipUnetEncoder``, ``encoder_train.unet_train_forward``)."""

    DECODER_CHANNELS = [256, 128, 64, 32, 16]

    def construct_encoder(self, input_dim: int, encoder_depth: int) -> nn.Module:
        try:
            import segmentation_models_pytorch as smp
        except ImportError:
            return VggUnet(input_dim, encoder_depth, self.DECODER_CHANNELS[:encoder_depth])
        return smp.Unet(encoder_name="vgg16_bn", encoder_weights=None, classes=1, in_channels=input_dim,
                        encoder_depth=encoder_depth, decoder_channels=self.DECODER_CHANNELS[:encoder_depth])
