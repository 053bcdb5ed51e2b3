"""MFMA inference paths of the reference's ``CNN`` encoder (``csrc/nastar_encoder.hip.h``): bf16 (fast) and f16x3 (fp32-grade).

``HipCnnEncoder`` wraps a ``planner.encoder.CNN`` module (depth 4: 2 -> 32 -> 64 -> 128 -> 256 -> 1): it folds the eval-mode
BatchNorm and the conv bias into per-channel scale/shift, packs the weights in the kernel's ``[tap][cin/8][cout][8]`` bf16
layout (zero padded), and runs ``nastar_encoder_cnn_forward``.  Inference only: gradients do not flow (training keeps the
torch path).  Numerics: bf16 operands, fp32 accumulation -> the cost map differs from the fp32 torch encoder by ~1e-3
(tests bound it); the search downstream is still exact for whatever cost map it is given.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

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
