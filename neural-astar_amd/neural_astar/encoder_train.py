"""Training path of the ``CNN`` cost-map encoder on the MI355X kernels (SURVEY.md section 8f "next #1", backward part).

The reference trains ``NeuralAstar`` by autograd through ``encoder.py:60-78`` (conv3x3 -> BatchNorm2d with BATCH statistics -> ReLU,
four times, then conv3x3 -> BatchNorm2d -> sigmoid * const, ``encoder.py:32-34``) with ``utils/training.py:55-61``.  Here the whole
trunk -- the five convolutions, the four hidden BatchNorm + ReLU blocks and every one of their gradients -- is ONE autograd node
(``_CnnTrunk``) made of launches of this package's kernels through the layer-level C ABI:

  forward    z_l = conv(a_{l-1})                     nastar_conv3x3_f16          (fp16 MFMA; split "f16x3" operands = fp32-grade)
             (sum z, sum z^2) per channel            nastar_chan_stats_f16       -> batch mean / variance, running statistics
             a_l = relu(gamma (z - mean)/std + beta) nastar_chan_affine_f16
  backward   (sum dy, sum dy z), dy = da [a > 0]     nastar_chan_stats_f16       -> dgamma, dbeta, BatchNorm-backward means
             dz_l = k1 dy + k2 z + k3                nastar_chan_affine_f16      (closed-form BatchNorm backward)
             dW_l = dz_l (*) a_{l-1}                  nastar_conv3x3_wgrad_f16    (fp16 MFMA, LDS transpose reads)
             da_{l-1} = conv(dz_l, W_l^T flipped)    nastar_conv3x3_f16

The last block's 1-channel BatchNorm + sigmoid * const stay ordinary torch ops on a [B,1,H,W] tensor (negligible, and torch's
autograd then also serves ``const`` and that BatchNorm's parameters).  Gradients travel multiplied by a power of two chosen on the
device from max|dL/dz5| (no host sync) so that fp16 never under- or overflows; it is divided out of every result.

Conv biases sit in front of a BatchNorm, so their true gradient is exactly zero (the batch mean removes them); zeros are returned
where torch's autograd returns ~1e-10 of rounding noise.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.nn as nn

from . import _native
from .encoder_hip import CONV_FINAL, CONV_RELU, CONV_SPLIT, _pad32, pack_conv_weight, pack_conv_weight_f16x3

CONV_RAW = 16  # include/nastar.h


def pack_flat_weight(w: torch.Tensor, split: bool):
    """[cout, cin, 3, 3] fp32 -> (wpack, unscale): channels zero padded to multiples of 32, the kernel's ``[9][cin_v/8][cout][8]`` fp16
    order; split form: weights times 2^s (max|w| -> ~2^14, keeps the lo terms normal fp16 numbers), ``unscale`` = 2^-s as a DEVICE
    scalar (computed without a host sync: weights change every optimiser step)."""
    cout, cin = w.shape[:2]
    cin_p, cout_p = _pad32(cin), _pad32(cout)
    wp = torch.zeros((cout, cin_p, 3, 3), dtype=torch.float32, device=w.device)
    wp[:, :cin] = w.detach().float()
    if not split:
        return pack_conv_weight(wp, cin_p, cout_p, torch.float16), torch.ones((), device=w.device)
    wmax = wp.abs().max().clamp_min(1e-30)
    s = torch.floor(torch.log2(16384.0 / wmax)).clamp(0, 24)
    scale = torch.exp2(s)
    return pack_conv_weight_f16x3(wp * scale, cout_p), 1.0 / scale


def _split_view(buf: torch.Tensor, npix: int, C: int, split: bool) -> torch.Tensor:
    return buf.view(npix, (2 if split else 1) * C)


class _Lib:
    """thin typed wrappers over the C ABI (all tensors on the current device, launches on torch's current stream)"""

    def __init__(self, dev):
        self.lib = _native.load()
        self.dev = dev
        self.stream = torch.cuda.current_stream(dev).cuda_stream

    def conv(self, src, wpack, scale, shift, B, H, W, cin, cout, flags, out=None, out_f32=None):
        rc = self.lib.nastar_conv3x3_f16(src.data_ptr(), None, wpack.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                         out.data_ptr() if out is not None else None,
                                         out_f32.data_ptr() if out_f32 is not None else None, B, H, W, cin, 0, cout, flags, 1.0, self.stream)
        _native.check(rc, "nastar_conv3x3_f16")

    def stats(self, u, v, ms, mt, npix, C, split):
        sums = torch.empty((C, 2), dtype=torch.float64, device=self.dev)
        rc = self.lib.nastar_chan_stats_f16(u.data_ptr() if u is not None else None, v.data_ptr(),
                                            ms.data_ptr() if ms is not None else None, mt.data_ptr() if mt is not None else None,
                                            sums.data_ptr(), npix, C, int(split), self.stream)
        _native.check(rc, "nastar_chan_stats_f16")
        return sums

    def affine(self, u, v, k1, k2, k3, ms, mt, out, npix, C, relu, split):
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        rc = self.lib.nastar_chan_affine_f16(p(u), p(v), p(k1), p(k2), p(k3), p(ms), p(mt), out.data_ptr(), npix, C, int(relu),
                                             int(split), self.stream)
        _native.check(rc, "nastar_chan_affine_f16")

    def wgrad(self, dz, a, B, H, W, co, ci, split):
        dw = torch.empty((9, ci, co), dtype=torch.float32, device=self.dev)
        rc = self.lib.nastar_conv3x3_wgrad_f16(dz.data_ptr(), a.data_ptr(), dw.data_ptr(), B, H, W, co, ci, int(split), 1.0, self.stream)
        _native.check(rc, "nastar_conv3x3_wgrad_f16")
        return dw.view(3, 3, ci, co).permute(3, 2, 0, 1)  # -> [co, ci, 3, 3]


def supported_shape(H: int, W: int) -> bool:
    """nastar_conv3x3_wgrad_f16 works on chunks of 64 pixels = whole image rows"""
    return 2 <= W <= 64 and 64 % W == 0 and H % (64 // W) == 0


class _CnnTrunk(torch.autograd.Function):
    """(map, start+goal inputs, conv / BatchNorm parameters of the 4 hidden blocks, last conv) -> z5 [B,1,H,W] fp32 (raw output of the
    last convolution, bias included).  ``cfg``: dict(split, plus, eps[4], bns[4] for the running statistics, training momentum)."""

    @staticmethod
    def forward(ctx, cfg, m, s, g, *params):
        split = cfg["split"]
        dev = m.device
        B, H, W = m.shape
        npix = B * H * W
        L = _Lib(dev)
        mult = 2 if split else 1
        sflag = CONV_SPLIT if split else 0
        ws = list(params[0:20:4]) + []      # conv weights of blocks 1..5
        bs = list(params[1:20:4])           # conv biases
        gammas = list(params[2:16:4])       # BatchNorm weights of blocks 1..4
        betas = list(params[3:16:4])
        with torch.cuda.device(dev):
            x0 = torch.empty((npix * 32 * mult,), dtype=torch.int16, device=dev)
            rc = L.lib.nastar_encoder_prep_f16(m.data_ptr(), s.data_ptr() if cfg["plus"] else None, g.data_ptr() if cfg["plus"] else None,
                                               int(cfg["plus"]), npix, 32, int(split), x0.data_ptr(), L.stream)
            _native.check(rc, "nastar_encoder_prep_f16")
            acts, zs, coef = [x0], [], []
            for l in range(4):
                w = ws[l]
                cout, cin_p = w.shape[0], _pad32(w.shape[1])
                wpack, unscale = pack_flat_weight(w, split)
                scale = unscale.expand(cout).contiguous().float()
                z = torch.empty((npix * cout * mult,), dtype=torch.int16, device=dev)
                L.conv(acts[-1], wpack, scale, bs[l].detach().float().contiguous(), B, H, W, cin_p, cout, sflag, out=z)
                sums = L.stats(None, z, None, None, npix, cout, split)
                mean = sums[:, 0] / npix
                var = (sums[:, 1] / npix - mean * mean).clamp_min(0.0)
                invstd = torch.rsqrt(var + cfg["eps"][l])
                k2 = (gammas[l].detach().double() * invstd).float().contiguous()
                k3 = (betas[l].detach().double() - mean * gammas[l].detach().double() * invstd).float().contiguous()
                a = torch.empty_like(z)
                L.affine(None, z, None, k2, k3, None, None, a, npix, cout, True, split)
                bn = cfg["bns"][l]
                if bn is not None and bn.track_running_stats:  # nn.BatchNorm2d's training-mode side effect (unbiased variance)
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean.float() * mom)
                    bn.running_var.mul_(1 - mom).add_((var * (npix / max(npix - 1, 1))).float() * mom)
                    bn.num_batches_tracked += 1
                zs.append(z)
                acts.append(a)
                coef.append((mean, invstd, k2, k3))
            w5 = ws[4]
            wpack5, unscale5 = pack_flat_weight(w5, split)  # cout 1 -> 32
            scale5 = torch.zeros(32, device=dev)
            scale5[0] = unscale5
            shift5 = torch.zeros(32, device=dev)
            shift5[0] = bs[4].detach().float()[0]
            z5 = torch.empty((B, H, W), dtype=torch.float32, device=dev)
            L.conv(acts[-1], wpack5, scale5, shift5, B, H, W, w5.shape[1], 32, sflag | CONV_FINAL | CONV_RAW, out_f32=z5)
        ctx.cfg = cfg
        ctx.shape = (B, H, W)
        ctx.acts, ctx.zs, ctx.coef = acts, zs, coef
        ctx.save_for_backward(*params)
        return z5.unsqueeze(1)

    @staticmethod
    def backward(ctx, dz5):
        cfg = ctx.cfg
        split = cfg["split"]
        params = ctx.saved_tensors
        ws = list(params[0:20:4])
        gammas = list(params[2:16:4])
        B, H, W = ctx.shape
        npix = B * H * W
        dev = dz5.device
        L = _Lib(dev)
        mult = 2 if split else 1
        sflag = CONV_SPLIT if split else 0
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        with torch.cuda.device(dev):
            d = dz5.reshape(npix).float()
            # power-of-two gradient scale from the device-side maximum: scaled gradients peak near 2^10 (fp16: no overflow, and 2^-24
            # of the peak is still a representable hi term)
            amax = d.abs().max().clamp_min(1e-30)
            S = torch.exp2(torch.floor(torch.log2(1024.0 / amax)).clamp(-60, 60))
            ds = d * S
            dzb = torch.zeros((npix, 32 * mult), dtype=torch.float16, device=dev)
            hi = ds.to(torch.float16)
            dzb[:, 0] = hi
            if split:
                dzb[:, 32] = (ds - hi.float()).to(torch.float16)
            dzb = dzb.view(torch.int16).reshape(-1)
            cur_co = 32  # padded channel count of the current dz
            for l in range(4, -1, -1):
                w = ws[l]
                cout, cin = w.shape[:2]
                cin_p = _pad32(cin)
                a_prev = ctx.acts[l]
                dw = L.wgrad(dzb, a_prev, B, H, W, cur_co, cin_p, split)            # [cur_co, cin_p, 3, 3] * S
                grads[4 * l] = (dw[:cout, :cin] / S).contiguous()
                grads[4 * l + 1] = torch.zeros_like(params[4 * l + 1])             # conv bias in front of a BatchNorm: exactly 0
                if l == 0:
                    break
                # input gradient: the same convolution with W^T flipped (cin <-> cout)
                wd = w.detach().float().transpose(0, 1).flip(2, 3)                  # [cin, cout, 3, 3]
                if cur_co != cout:                                                  # last block: cout 1 padded to 32 input channels
                    wd = torch.cat((wd, torch.zeros((cin, cur_co - cout, 3, 3), device=dev)), dim=1)
                wpack, unscale = pack_flat_weight(wd, split)
                da = torch.empty((npix * cin_p * mult,), dtype=torch.int16, device=dev)
                L.conv(dzb, wpack, unscale.expand(cin_p).contiguous().float(), torch.zeros(cin_p, device=dev), B, H, W, cur_co, cin_p,
                       sflag, out=da)
                # ReLU mask + BatchNorm backward of block l (its output is a_l = acts[l], pre-activation zs[l-1])
                z = ctx.zs[l - 1]
                mean, invstd, k2f, k3f = ctx.coef[l - 1]
                C = cin_p
                sums = L.stats(da, z, k2f, k3f, npix, C, split)                     # (sum dy, sum dy z) * S
                sdy, sdyz = sums[:, 0], sums[:, 1]
                sdyx = (sdyz - mean * sdy) * invstd                                  # sum dy * xhat
                gam = gammas[l - 1].detach().double()
                grads[4 * (l - 1) + 2] = (sdyx / S).float()
                grads[4 * (l - 1) + 3] = (sdy / S).float()
                k1 = gam * invstd
                m1, m2 = sdy / npix, sdyx / npix
                # re-centre the gradient scale for the next block: |dz| <~ 2 max|k1| max|da| (BatchNorm backward amplifies by
                # gamma / std, which can be far from 1); one reduction over da, all on the device
                amax_da = da.view(torch.float16).abs().max().double().clamp_min(1e-30)
                r = torch.exp2(torch.floor(torch.log2(1024.0 / (2.0 * k1.abs().max().clamp_min(1e-30) * amax_da))).clamp(-40, 40))
                S = S * r
                c1 = (k1 * r).float().contiguous()
                c2 = (-k1 * m2 * invstd * r).float().contiguous()
                c3 = ((-k1 * m1 + k1 * m2 * mean * invstd) * r).float().contiguous()
                dzb = torch.empty_like(da)
                L.affine(da, z, c1, c2, c3, k2f, k3f, dzb, npix, C, False, split)
                cur_co = C
        return (None, None, None, None) + tuple(grads)


def cnn_train_forward(cnn: nn.Module, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor, plus: bool,
                      precision: str = "f16x3") -> torch.Tensor:
    """``cnn(cat(map, start + goal))`` for a ``planner.encoder.CNN`` of depth 4 in TRAINING mode (batch-statistics BatchNorm, running
    statistics updated), differentiable w.r.t. every encoder parameter, on the MI355X kernels.  Returns the cost map [B,1,H,W] fp32."""
    layers = list(cnn.model)
    convs = [m for m in layers if isinstance(m, nn.Conv2d)]
    bns = [m for m in layers if isinstance(m, nn.BatchNorm2d)]
    if [c.out_channels for c in convs] != [32, 64, 128, 256, 1] or len(bns) != 5:
        raise NotImplementedError("cnn_train_forward implements the reference's depth-4 CNN (.. -> 32 -> 64 -> 128 -> 256 -> 1)")
    B, _, H, W = map_designs.shape
    if not supported_shape(H, W):
        raise NotImplementedError("W must divide 64 and H must be a multiple of 64 / W")
    params = []
    for l in range(5):
        params += [convs[l].weight, convs[l].bias, bns[l].weight, bns[l].bias]
    cfg = {"split": precision == "f16x3", "plus": bool(plus), "eps": [bn.eps for bn in bns[:4]], "bns": bns[:4]}
    m = map_designs[:, 0].contiguous()
    s = start_maps[:, 0].contiguous() if plus else m
    g = goal_maps[:, 0].contiguous() if plus else m
    z5 = _CnnTrunk.apply(cfg, m, s, g, *params[:18])
    bn5 = bns[4]
    y = nn.functional.batch_norm(z5, bn5.running_mean, bn5.running_var, bn5.weight, bn5.bias, True,
                                 bn5.momentum if bn5.momentum is not None else 0.1, bn5.eps)
    if bn5.track_running_stats:
        bn5.num_batches_tracked += 1
    return torch.sigmoid(y) * cnn.const
