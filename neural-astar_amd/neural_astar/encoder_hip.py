"""MFMA inference paths of the reference's encoders: ``CNN`` on the fixed-shape kernels (``csrc/nastar_encoder.hip.h``: bf16, fp16,
f16x3 = fp32-grade), ``CNNDownSize`` on the f32-input MFMA, ``Unet`` and CNNs of any depth / map size on the generic convolution
(``csrc/nastar_conv_flat.hip.h``).  Training lives in ``encoder_train.py``.

``HipCnnEncoder`` wraps a ``planner.encoder.CNN`` module (depth 4: 2 -> 32 -> 64 -> 128 -> 256 -> 1): it folds the eval-mode
BatchNorm and the conv bias into per-channel scale/shift, packs the weights in the kernel's ``[tap][cin/8][cout][8]`` bf16
layout (zero padded), and runs ``nastar_encoder_cnn_forward``.  Inference only: gradients do not flow (training keeps the
torch path).  Numerics: bf16 operands, fp32 accumulation -> the cost map differs from the fp32 torch encoder by ~1e-3
(tests bound it); the search downstream is still exact for whatever cost map it is given.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _native

_CIN_P = (16, 32, 64, 128, 256)
_COUT_P = (32, 64, 128, 256, 32)


def pack_conv_weight(w: torch.Tensor, cin_p: int, cout_p: int, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """[Cout, Cin, 3, 3] fp32 -> [9, cin_p/8, cout_p, 8] bf16 / fp16 (as int16 bits), zero padded, tap = ky*3 + kx."""
    cout, cin = w.shape[:2]
    wp = torch.zeros((cout_p, cin_p, 3, 3), dtype=torch.float32, device=w.device)
    wp[:cout, :cin] = w
    wp = wp.permute(2, 3, 1, 0).reshape(9, cin_p // 8, 8, cout_p).permute(0, 1, 3, 2).contiguous()
    return wp.to(dtype).view(torch.int16)


def split_f16(w: torch.Tensor):
    """w = hi + lo with hi = fp16(w), lo = fp16(w - hi): 22 significant bits in two fp16 terms (both returned as fp32 tensors)."""
    hi = w.to(torch.float16).float()
    lo = (w - hi).to(torch.float16).float()
    return hi, lo


def pack_conv_weight_f16x3(w: torch.Tensor, cout_p: int) -> torch.Tensor:
    """Weights of a layer whose input arrives as [x_hi | x_lo]: packed over 3*Cin virtual channels [W_hi | W_hi | W_lo], so that
    the kernel's slices compute x_hi*W_hi + x_lo*W_hi + x_hi*W_lo (the lo*lo term, 2^-22 relative, is dropped)."""
    hi, lo = split_f16(w)
    return pack_conv_weight(torch.cat((hi, hi, lo), dim=1), 3 * w.shape[1], cout_p, torch.float16)


def fold_bn(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], cout_p: int):
    """y = BN_eval(conv(x) + bias) = acc * scale + shift  (per output channel; padded channels get scale 0, shift 0)."""
    cout = conv.out_channels
    dev = conv.weight.device
    bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(cout, device=dev)
    if bn is not None:
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        sh = bn.bias.detach().float() + (bias - bn.running_mean.detach().float()) * s
    else:
        s, sh = torch.ones(cout, device=dev), bias
    scale = torch.zeros(cout_p, device=dev)
    shift = torch.zeros(cout_p, device=dev)
    scale[:cout], shift[:cout] = s, sh
    return scale.contiguous(), shift.contiguous()


class HipCnnEncoder:
    """``precision``: ``"bf16"`` (default: bf16 operands, 2.8 ms per 4096 32x32 maps, cost maps within ~1e-3 of the fp32 encoder),
    ``"f16"`` (plain fp16 operands: 8x finer than bf16 at the same speed; activations must stay below 65504) or ``"f16x3"`` (split fp16 operands, ~3x the
    matrix work, cost maps within 1e-5 -- the north-star tolerance for float outputs)."""

    def __init__(self, cnn: nn.Module, precision: str = "bf16"):
        if precision not in ("bf16", "f16", "f16x3"):
            raise ValueError(precision)
        self.precision = precision
        layers = list(cnn.model)
        convs = [m for m in layers if isinstance(m, nn.Conv2d)]
        if [c.out_channels for c in convs] != [32, 64, 128, 256, 1] or convs[0].in_channels > 16:
            raise NotImplementedError("the HIP encoder implements the reference's depth-4 CNN (.. -> 32 -> 64 -> 128 -> 256 -> 1)")
        self.cnn = cnn
        self.in_channels = convs[0].in_channels
        self._key = None
        self._ws: Optional[torch.Tensor] = None
        self._refresh()

    def _refresh(self) -> None:
        layers = list(self.cnn.model)
        # version counters miss module.to(device) / .half() / .float() / `param.data = ...`: key on storage, device and dtype too
        key = tuple((int(t._version), t.data_ptr(), str(t.device), t.dtype)
                    for t in list(self.cnn.parameters()) + list(self.cnn.buffers()))
        if key == self._key:
            return
        self.wpack: List[torch.Tensor] = []
        self.scale: List[torch.Tensor] = []
        self.shift: List[torch.Tensor] = []
        idx = 0
        for li in range(5):
            conv = layers[idx]
            bn = layers[idx + 1] if idx + 1 < len(layers) and isinstance(layers[idx + 1], nn.BatchNorm2d) else None
            self.wpack.append(pack_conv_weight(conv.weight.detach().float(), _CIN_P[li], _COUT_P[li]))
            sc, sh = fold_bn(conv, bn, _COUT_P[li])
            self.scale.append(sc)
            self.shift.append(sh)
            idx += 3 if li < 4 else 2  # conv, bn, relu  |  conv, bn
        if self.precision == "f16":
            convs = [m for m in layers if isinstance(m, nn.Conv2d)]
            self.w1 = convs[0].weight.detach().float().contiguous()
            self.wsplit = [pack_conv_weight(convs[li].weight.detach().float(), _CIN_P[li], _COUT_P[li], torch.float16) for li in range(5)]
        if self.precision == "f16x3":
            convs = [m for m in layers if isinstance(m, nn.Conv2d)]
            self.w1 = convs[0].weight.detach().float().contiguous()
            hi5, lo5 = split_f16(convs[4].weight.detach().float())
            self.wsplit = [pack_conv_weight_f16x3(convs[li].weight.detach().float(), _COUT_P[li]) for li in (1, 2, 3)]
            self.wsplit += [pack_conv_weight(hi5, 256, 32, torch.float16), pack_conv_weight(lo5, 256, 32, torch.float16)]
        const = self.cnn.const  # read back once per weight version (a per-call .item() would sync the stream)
        self._mul = float(const.detach().item()) if isinstance(const, torch.Tensor) else float(const)
        self._key = key

    def __call__(self, map_designs: torch.Tensor, start_maps: Optional[torch.Tensor], goal_maps: Optional[torch.Tensor],
                 plus: bool) -> torch.Tensor:
        """map/start/goal [B,1,H,W] fp32 on the device -> cost [B,1,H,W] fp32 = sigmoid(model(x)) * const."""
        if self.cnn.training:
            raise RuntimeError("HipCnnEncoder is inference only (eval-mode BatchNorm is folded into the kernel)")
        self._refresh()
        lib = _native.load()
        m = map_designs[:, 0].contiguous()
        B, H, W = m.shape
        dev = m.device
        s = start_maps[:, 0].contiguous() if plus else None
        g = goal_maps[:, 0].contiguous() if plus else None
        cost = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        split = self.precision in ("f16", "f16x3")
        fwd16 = lib.nastar_encoder_cnn_forward_f16x3 if self.precision == "f16x3" else lib.nastar_encoder_cnn_forward_f16
        ws_bytes = int({"bf16": lib.nastar_encoder_workspace_bytes, "f16": lib.nastar_encoder_workspace_bytes_f16,
                        "f16x3": lib.nastar_encoder_workspace_bytes_f16x3}[self.precision](B, H, W))
        ws = self._ws  # activation slabs, kept across calls (grown on demand) instead of re-allocated per batch
        if ws is None or ws.device != dev or ws.numel() < ws_bytes:
            ws = self._ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        arr = ctypes.c_void_p * 5
        mul = self._mul
        if split:
            with torch.cuda.device(dev):
                rc = fwd16(
                    m.data_ptr(), s.data_ptr() if plus else None, g.data_ptr() if plus else None, int(plus), B, H, W,
                    self.w1.data_ptr(), arr(*[t.data_ptr() for t in self.wsplit]), arr(*[t.data_ptr() for t in self.scale]),
                    arr(*[t.data_ptr() for t in self.shift]), mul, cost.data_ptr(), ws.data_ptr(), ws.numel(),
                    torch.cuda.current_stream(dev).cuda_stream)
            _native.check(rc, "nastar_encoder_cnn_forward_" + self.precision)
            return cost.unsqueeze(1)
        with torch.cuda.device(dev):
            rc = lib.nastar_encoder_cnn_forward(
                m.data_ptr(), s.data_ptr() if plus else None, g.data_ptr() if plus else None, int(plus), B, H, W,
                arr(*[t.data_ptr() for t in self.wpack]), arr(*[t.data_ptr() for t in self.scale]),
                arr(*[t.data_ptr() for t in self.shift]), mul, cost.data_ptr(), ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "nastar_encoder_cnn_forward")
        return cost.unsqueeze(1)


def pack_conv_weight_f32(weight: torch.Tensor, cin_p: int, cout_p: int) -> torch.Tensor:
    """torch conv weight [cout, cin, 3, 3] -> fp32 [9][cin_p][cout_p] (tap = ky*3+kx, output channel contiguous), zero padded:
    the B-operand order of nastar_conv3x3_f32mfma_kernel."""
    cout, cin = weight.shape[:2]
    w = torch.zeros((9, cin_p, cout_p), dtype=torch.float32, device=weight.device)
    w[:, :cin, :cout] = weight.detach().float().permute(2, 3, 1, 0).reshape(9, cin, cout)
    return w.contiguous()


class HipCnnDownSizeEncoder:
    """Eval-mode ``CNNDownSize`` (reference planner/encoder.py:81-97; the WarCraft encoder) through
    ``nastar_encoder_cnn_downsize_forward``: f32-input MFMA, fp32 accuracy (csrc/nastar_encoder_downsize.hip.h)."""

    def __init__(self, cnn: nn.Module):
        convs = [m for m in cnn.model if isinstance(m, nn.Conv2d)]
        self.depth = len(convs) - 1
        if not (1 <= self.depth <= 4) or [c.out_channels for c in convs] != [32, 64, 128, 256][:self.depth] + [1] \
                or convs[0].in_channels > 4:
            raise NotImplementedError("the HIP CNNDownSize encoder implements input (<= 4 ch) -> 32 -> 64 -> 128 -> 256 (depth 1..4) -> 1")
        self.cnn = cnn
        self.in_channels = convs[0].in_channels
        self._key = None
        self._ws: Optional[torch.Tensor] = None
        self._refresh()

    def _refresh(self) -> None:
        key = tuple((int(t._version), t.data_ptr(), str(t.device), t.dtype)
                    for t in list(self.cnn.parameters()) + list(self.cnn.buffers()))
        if key == self._key:
            return
        layers = list(self.cnn.model)
        convs = [i for i, m in enumerate(layers) if isinstance(m, nn.Conv2d)]
        self.w, self.scale, self.shift = [], [], []
        for k, i in enumerate(convs):
            conv = layers[i]
            bn = layers[i + 1] if i + 1 < len(layers) and isinstance(layers[i + 1], nn.BatchNorm2d) else None
            cin_p = (2 if conv.in_channels <= 2 else 4) if k == 0 else conv.in_channels
            cout_p = conv.out_channels if k < self.depth else 32
            self.w.append(pack_conv_weight_f32(conv.weight, cin_p, cout_p))
            sc, sh = fold_bn(conv, bn, cout_p)
            self.scale.append(sc)
            self.shift.append(sh)
        const = self.cnn.const
        self._mul = float(const.detach().item()) if isinstance(const, torch.Tensor) else float(const)
        self._key = key

    def __call__(self, images: torch.Tensor, start_maps: Optional[torch.Tensor], goal_maps: Optional[torch.Tensor],
                 plus: bool) -> torch.Tensor:
        """images [B,C,H,W] fp32, start/goal [B,1,h,w] -> cost [B,1,H>>depth,W>>depth] fp32 = sigmoid(model(x)) * const."""
        if self.cnn.training:
            raise RuntimeError("HipCnnDownSizeEncoder is inference only (eval-mode BatchNorm is folded into the kernel)")
        self._refresh()
        lib = _native.load()
        x = images.contiguous()
        B, C, H, W = x.shape
        if C + (1 if plus else 0) != self.in_channels:
            raise ValueError(f"encoder expects {self.in_channels} input channels, got {C} + {int(plus)}")
        dev = x.device
        s = start_maps[:, 0].contiguous() if plus else None
        g = goal_maps[:, 0].contiguous() if plus else None
        h, w = (s.shape[-2], s.shape[-1]) if plus else (1, 1)
        Ho, Wo = H >> self.depth, W >> self.depth
        cost = torch.empty((B, Ho, Wo), dtype=torch.float32, device=dev)
        ws_bytes = int(lib.nastar_encoder_downsize_workspace_bytes(B, self.in_channels, H, W, self.depth))
        ws = self._ws
        if ws is None or ws.device != dev or ws.numel() < ws_bytes:
            ws = self._ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        arr = ctypes.c_void_p * (self.depth + 1)
        with torch.cuda.device(dev):
            rc = lib.nastar_encoder_cnn_downsize_forward(
                x.data_ptr(), s.data_ptr() if plus else None, g.data_ptr() if plus else None, int(plus), B, C, H, W, h, w,
                self.depth, arr(*[t.data_ptr() for t in self.w]), arr(*[t.data_ptr() for t in self.scale]),
                arr(*[t.data_ptr() for t in self.shift]), self._mul, cost.data_ptr(), ws.data_ptr(), ws.numel(),
                torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "nastar_encoder_cnn_downsize_forward")
        return cost.unsqueeze(1)


# ---- U-Net (vgg16_bn) encoder on the generic fp16 / f16x3 MFMA convolution (csrc/nastar_conv_flat.hip.h) ---------------------------------
CONV_RELU, CONV_FINAL, CONV_UPSAMPLE, CONV_SPLIT = 1, 2, 4, 8  # include/nastar.h


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def pack_flat_conv(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], split: bool, in_scale: float = 1.0, out_scale: float = 1.0):
    """(wpack, scale, shift, cin_p, cout_p) of one conv(+BN) layer for ``nastar_conv3x3_f16``: input / output channels zero padded to
    multiples of 32, weights ``[9][cin_v/8][cout_p][8]`` fp16 with cin_v = cin_p, or 3*cin_p virtual channels [W_hi | W_hi | W_lo]
    in the split ("f16x3") form; eval-mode BatchNorm and bias folded into per-channel scale / shift.

    Split form only: the weights are multiplied by a power of two 2^s that brings max|w| to ~2^14 before they are split, and 2^-s is
    folded into ``scale`` (exact).  Conv weights are ~1e-2, so their lo terms (~2^-12 |w|) would otherwise be fp16 SUBNORMALS
    (< 6.1e-5, absolute quantum 6e-8): 19 significant bits instead of 22.  ``in_scale`` / ``out_scale`` (powers of two) say that the
    layer's input activations arrive multiplied by ``in_scale`` and that its output must leave multiplied by ``out_scale`` -- the same
    cure for the activations' lo terms (ReLU and max-pool commute with a positive factor)."""
    cout, cin = conv.weight.shape[:2]
    cin_p, cout_p = _pad32(cin), _pad32(cout)
    w = torch.zeros((cout, cin_p, 3, 3), dtype=torch.float32, device=conv.weight.device)
    w[:, :cin] = conv.weight.detach().float()
    scale, shift = fold_bn(conv, bn, cout_p)
    if split:
        wmax = float(w.abs().max())
        s = 0 if wmax == 0.0 else max(0, min(24, int(np.floor(np.log2(16384.0 / wmax)))))
        wpack = pack_conv_weight_f16x3(w * (2.0 ** s), cout_p)
        scale = scale * (2.0 ** -s)
    else:
        wpack = pack_conv_weight(w, cin_p, cout_p, torch.float16)
    scale = (scale * (out_scale / in_scale)).contiguous()
    shift = (shift * out_scale).contiguous()
    return wpack, scale, shift, cin_p, cout_p


def unet_layer_plan(model: nn.Module):
    """The launch sequence of a ``planner.encoder.VggUnet`` (reference encoder.py:37-57 via this package's from-scratch definition) as a
    list of steps over named activation buffers ("x0" = the assembled input; ``div`` = resolution divisor of the step's OUTPUT):
        ("conv", dst, src, skip|None, conv, bn|None, flags, div)     flags: CONV_RELU | CONV_UPSAMPLE | CONV_FINAL
        ("pool", dst, src, None, div_in)"""
    steps = []
    depth = model.depth
    x, div, n = "x0", 1, 0
    feats = []
    for si, stage in enumerate(model._stages()[: depth + 1]):
        mods = list(stage)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.MaxPool2d):
                steps.append(("pool", f"p{si}", x, None, div))
                x, div = f"p{si}", div * 2
                i += 1
            elif isinstance(m, nn.Conv2d):
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
                n += 1
                steps.append(("conv", f"e{n}", x, None, m, bn, CONV_RELU, div))
                x = f"e{n}"
                i += 3 if bn is not None else 2
            else:
                i += 1
        feats.append((x, div))
    skips = feats[1:][::-1]  # the full-resolution stage is not a skip (VggUnet.decoder)
    x, div = skips[0]
    for k, blk in enumerate(model.decoder.center):
        steps.append(("conv", f"c{k}", x, None, blk[0], blk[1], CONV_RELU, div))
        x = f"c{k}"
    for k, blk in enumerate(model.decoder.blocks):
        skip = skips[k + 1][0] if k + 1 < len(skips) else None
        div //= 2
        steps.append(("conv", f"d{k}a", x, skip, blk.conv1[0], blk.conv1[1], CONV_RELU | CONV_UPSAMPLE, div))
        steps.append(("conv", f"d{k}b", f"d{k}a", None, blk.conv2[0], blk.conv2[1], CONV_RELU, div))
        x = f"d{k}b"
    assert div == 1, "decoder must return to the input resolution (encoder_depth decoder blocks)"
    steps.append(("conv", "cost", x, None, model.segmentation_head[0], None, CONV_FINAL, 1))
    return steps


def sequential_layer_plan(seq: nn.Module):
    """Launch plan (same step format as :func:`unet_layer_plan`) of a plain ``nn.Sequential`` of Conv2d / BatchNorm2d / ReLU /
    MaxPool2d blocks whose last convolution (+ BatchNorm, no ReLU) has one output channel: the reference's ``CNN`` at any depth
    (encoder.py:60-78) and any other stack of that shape."""
    mods = list(seq)
    steps, x, div, n, i = [], "x0", 1, 0, 0
    last_conv = max(k for k, m in enumerate(mods) if isinstance(m, nn.Conv2d))
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.MaxPool2d):
            steps.append(("pool", f"p{n}", x, None, div))
            x, div = f"p{n}", div * 2
            i += 1
        elif isinstance(m, nn.Conv2d):
            if m.kernel_size != (3, 3) or m.padding != (1, 1) or m.stride != (1, 1):
                raise NotImplementedError("3x3 convolutions with padding 1 only")
            j = i + 1
            bn = mods[j] if j < len(mods) and isinstance(mods[j], nn.BatchNorm2d) else None
            j += bn is not None
            relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
            j += relu
            n += 1
            if i == last_conv:
                if m.out_channels != 1 or relu or div != 1:
                    raise NotImplementedError("the last block must be a 1-channel conv (+ BatchNorm) at the input resolution")
                steps.append(("conv", "cost", x, None, m, bn, CONV_FINAL, div))
            else:
                steps.append(("conv", f"e{n}", x, None, m, bn, CONV_RELU if relu else 0, div))
            x = f"e{n}"
            i = j
        else:
            raise NotImplementedError(type(m).__name__)
    return steps


class HipUnetEncoder:
    """Eval-mode ``planner.encoder.Unet`` (from-scratch VggUnet definition; reference planner/encoder.py:37-57) on the MFMA through
    the layer-level C ABI ``nastar_conv3x3_f16`` / ``nastar_maxpool2x2_f16`` / ``nastar_encoder_prep_f16``.

    ``precision``: ``"f16"`` (plain fp16 operands, fp32 accumulation -- BASELINE config 3) or ``"f16x3"`` (split operands, results
    within ~1e-5 of the fp32 torch module).  Inference only; upsampling, skip concatenation, BatchNorm, ReLU, bias, sigmoid * const are
    all fused into the convolution launches (26 + 4 pooling launches per forward at depth 4)."""

    def _plan(self):
        return unet_layer_plan(self.unet.model)

    def _check(self, unet):
        from .planner.encoder import VggUnet
        if not isinstance(unet.model, VggUnet):
            raise NotImplementedError("HipUnetEncoder implements this package's VggUnet definition of Unet(vgg16_bn)")
        self.size_multiple = 1 << unet.model.depth

    def __init__(self, unet: nn.Module, precision: str = "f16"):
        if precision not in ("f16", "f16x3"):
            raise ValueError(precision)
        self._check(unet)
        self.unet = unet
        self.precision = precision
        self.split = precision == "f16x3"
        self._key = None
        self._bufs = {}
        self._refresh()

    def _refresh(self) -> None:
        key = tuple((int(t._version), t.data_ptr(), str(t.device), t.dtype)
                    for t in list(self.unet.parameters()) + list(self.unet.buffers()))
        if key == self._key:
            return
        self.steps = []
        for st in self._plan():
            if st[0] == "conv":
                _, dst, src, skip, conv, bn, flags, div = st
                # split form: hidden activations travel multiplied by 16 (their lo terms stay out of the fp16 subnormal range)
                a = 16.0 if self.split else 1.0
                wpack, scale, shift, cin_p, cout_p = pack_flat_conv(conv, bn, self.split, 1.0 if src == "x0" else a,
                                                                    1.0 if flags & CONV_FINAL else a)
                self.steps.append(("conv", dst, src, skip, wpack, scale, shift, cin_p, cout_p, flags, div))
            else:
                self.steps.append(st)
        const = self.unet.const
        self._mul = float(const.detach().item()) if isinstance(const, torch.Tensor) else float(const)
        self._key = key

    def flops(self, H: int, W: int) -> float:
        """useful multiply-add FLOPs of the conv layers per image (real channel counts)"""
        total = 0.0
        for st in self._plan():
            if st[0] == "conv":
                conv, div = st[4], st[7]
                total += 2.0 * 9 * conv.in_channels * conv.out_channels * (H // div) * (W // div)
        return total

    def _buf(self, name: str, numel: int, dev) -> torch.Tensor:
        t = self._bufs.get(name)
        if t is None or t.device != dev or t.numel() < numel:
            t = self._bufs[name] = torch.empty((numel,), dtype=torch.int16, device=dev)
        return t

    def __call__(self, map_designs: torch.Tensor, start_maps: Optional[torch.Tensor], goal_maps: Optional[torch.Tensor],
                 plus: bool) -> torch.Tensor:
        """map/start/goal [B,1,H,W] fp32 on the device -> cost [B,1,H,W] fp32 = sigmoid(model(x)) * const."""
        if self.unet.training:
            raise RuntimeError("HipUnetEncoder is inference only (eval-mode BatchNorm is folded into the kernels)")
        self._refresh()
        lib = _native.load()
        m = map_designs[:, 0].contiguous()
        B, H, W = m.shape
        if H % self.size_multiple or W % self.size_multiple:
            raise NotImplementedError(f"H, W must be multiples of {self.size_multiple}")
        dev = m.device
        s = start_maps[:, 0].contiguous() if plus else None
        g = goal_maps[:, 0].contiguous() if plus else None
        cost = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        mult = 2 if self.split else 1
        # 32-bit offsets in 16-byte units inside a launch: chunk the batch so that the widest tensor stays below 2^34 fp16 elements
        per_image = max(st[8] * (H // st[10]) * (W // st[10]) for st in self.steps if st[0] == "conv") * mult
        chunk = max(1, min(B, ((1 << 34) - 1) // per_image))
        stream = torch.cuda.current_stream(dev).cuda_stream
        sflag = CONV_SPLIT if self.split else 0
        with torch.cuda.device(dev):
            for b0 in range(0, B, chunk):
                nb = min(chunk, B - b0)
                chans = {"x0": 32}
                x0 = self._buf("x0", nb * H * W * 32 * mult, dev)
                rc = lib.nastar_encoder_prep_f16(m[b0:].data_ptr(), s[b0:].data_ptr() if plus else None,
                                                 g[b0:].data_ptr() if plus else None, int(plus), nb * H * W, 32, int(self.split),
                                                 x0.data_ptr(), stream)
                _native.check(rc, "nastar_encoder_prep_f16")
                for st in self.steps:
                    if st[0] == "pool":
                        _, dst, src, _, div = st
                        c = chans[src]
                        h, w = H // div, W // div
                        out = self._buf(dst, nb * (h // 2) * (w // 2) * c * mult, dev)
                        rc = lib.nastar_maxpool2x2_f16(self._bufs[src].data_ptr(), out.data_ptr(), nb, h, w, c, int(self.split), stream)
                        _native.check(rc, "nastar_maxpool2x2_f16")
                        chans[dst] = c
                        continue
                    _, dst, src, skip, wpack, scale, shift, cin_p, cout_p, flags, div = st
                    h, w = H // div, W // div
                    c1 = chans[src]
                    c2 = chans[skip] if skip is not None else 0
                    assert c1 + c2 == cin_p, (dst, c1, c2, cin_p)
                    final = bool(flags & CONV_FINAL)
                    out = None if final else self._buf(dst, nb * h * w * cout_p * mult, dev)
                    rc = lib.nastar_conv3x3_f16(
                        self._bufs[src].data_ptr(), self._bufs[skip].data_ptr() if skip is not None else None, wpack.data_ptr(),
                        scale.data_ptr(), shift.data_ptr(), None if final else out.data_ptr(), cost[b0:].data_ptr() if final else None,
                        nb, h, w, c1, c2, cout_p, flags | sflag, self._mul, stream)
                    _native.check(rc, f"nastar_conv3x3_f16({dst})")
                    chans[dst] = cout_p
        return cost.unsqueeze(1)


class HipFlatCnnEncoder(HipUnetEncoder):
    """Eval-mode ``planner.encoder.CNN`` of ANY depth on maps of ANY size through the same generic fp16 / f16x3 MFMA
    convolution: what ``NeuralAstar.encode`` uses when the fixed-shape kernels of ``HipCnnEncoder`` (depth 4, H and W multiples of
    32 / 16) do not apply -- e.g. 20x45 or 24x40 maps, ``encoder_depth=3``."""

    def _plan(self):
        return sequential_layer_plan(self.unet.model)

    def _check(self, cnn):
        sequential_layer_plan(cnn.model)  # raises on anything but conv3x3 / BatchNorm / ReLU / MaxPool stacks ending in 1 channel
        if any(isinstance(m, nn.MaxPool2d) for m in cnn.model):
            raise NotImplementedError("pooling stacks change the output resolution: use HipCnnDownSizeEncoder")
        self.size_multiple = 1
