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

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _native
from .encoder_hip import CONV_FINAL, CONV_RELU, CONV_SPLIT, _pad32, pack_conv_weight, pack_conv_weight_f16x3

CONV_RAW = 16  # include/nastar.h


class SyncBatchNorm:
    """Process-wide switch for data-parallel training (set by ``utils.distributed.DataParallelTrainer(sync_bn=True)``).

    The reference's step runs on ONE device, so its BatchNorm layers (encoder.py:60-97, training mode) normalise with the statistics
    of the WHOLE batch and their backward couples all rows of it.  With the batch sharded over ranks the per-channel sums that the
    statistics kernels produce -- ``(sum z, sum z^2)`` forward, ``(sum dy, sum dy z)`` backward, [C][2] doubles -- are all-reduced
    before the coefficient kernels turn them into scale / shift (forward) and the closed-form backward coefficients: one tiny
    collective per BatchNorm layer and direction, after which a sharded step is the single-device step on the concatenated batch
    (parameters AND running statistics; equal shard sizes assumed, as everywhere in ``neural_astar.parallel``)."""
    enabled = False
    group = None
    force = False  # dev / bench: run the collectives even in a 1-rank group (RCCL smoke on a single GPU)

    @classmethod
    def world(cls) -> int:
        import torch.distributed as dist
        if not cls.enabled or not dist.is_available() or not dist.is_initialized():
            return 1
        return dist.get_world_size(cls.group)

    @classmethod
    def active(cls) -> bool:
        return cls.world() > 1 or (cls.force and cls.world() == 1 and cls.enabled and torch.distributed.is_initialized())

    @classmethod
    def snapshot(cls) -> Tuple[bool, object, int]:
        """(active, group, world) as a forward pass sees them.  The autograd nodes keep this in ``ctx`` and their backward uses IT,
        not the process-wide switch: a backward that runs after the trainer restored the switch (a retained graph, a caller that
        runs ``loss.backward()`` itself) must all-reduce exactly as its forward did, or dz / dgamma / dbeta silently mix global
        pixel counts with local sums and the ranks diverge."""
        return cls.active(), cls.group, cls.world()


def _sync_sums(sums: torch.Tensor, scale: Optional[torch.Tensor] = None, state: Optional[Tuple[bool, object, int]] = None) -> int:
    """all-reduce (SUM) per-channel double sums over the ranks when SyncBatchNorm is on; returns the world size (1 = untouched).
    ``scale``: device scalar S the sums are multiplied by on THIS rank (the fp16 gradient scale differs per rank): they travel
    unscaled and come back in this rank's scale (S is a power of two: exact).  ``state``: a ``SyncBatchNorm.snapshot()`` taken by the
    forward pass (backward passes hand in the one their forward recorded); None = the process-wide switch as it is now."""
    return _sync_sums_end(_sync_sums_begin(sums, scale, state), sums, scale, state)


def _sync_sums_begin(sums: torch.Tensor, scale: Optional[torch.Tensor], state: Optional[Tuple[bool, object, int]]):
    """first half of ``_sync_sums``: ISSUE the all-reduce (on the backend's own stream, behind everything queued so far) and return
    at once, so that launches which do not need the sums run beside the collective; None when SyncBatchNorm is off"""
    active, group, _ = state if state is not None else SyncBatchNorm.snapshot()
    if not active:
        return None
    import torch.distributed as dist
    if scale is not None:
        sums.div_(scale.double())
    return dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group, async_op=True)


def _sync_sums_end(work, sums: torch.Tensor, scale: Optional[torch.Tensor], state: Optional[Tuple[bool, object, int]]) -> int:
    """second half: make the current stream wait for the collective (no host sync), restore the rank's scale; returns the world size"""
    if work is None:
        return 1
    work.wait()
    if scale is not None:
        sums.mul_(scale.double())
    return (state if state is not None else SyncBatchNorm.snapshot())[2]


class _SyncBatchNorm1(torch.autograd.Function):
    """the last block's 1-channel BatchNorm with GLOBAL batch statistics: y = (z - mean) * invstd (affine applied by the caller)"""

    @staticmethod
    def forward(ctx, z, eps):
        import torch.distributed as dist
        zd = z.double()
        st = torch.stack((zd.sum(), (zd * zd).sum()))
        ctx.group = SyncBatchNorm.group
        dist.all_reduce(st, group=ctx.group)
        n = z.numel() * SyncBatchNorm.world()
        mean = st[0] / n
        var = (st[1] / n - mean * mean).clamp_min(0.0)
        invstd = torch.rsqrt(var + eps)
        xhat = ((zd - mean) * invstd).float()
        ctx.save_for_backward(xhat, invstd.float())
        ctx.n = n
        return xhat, mean.float(), var.float()

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        import torch.distributed as dist
        xhat, invstd = ctx.saved_tensors
        st = torch.stack((dy.double().sum(), (dy.double() * xhat.double()).sum()))
        dist.all_reduce(st, group=ctx.group)
        m1, m2 = (st[0] / ctx.n).float(), (st[1] / ctx.n).float()
        return invstd * (dy - m1 - xhat * m2), None


def pack_flat_weight(w: torch.Tensor, split: bool):  # torch-op reference of nastar_pack_conv_weight_f16 (tests compare the two)
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


class _Lib:
    """thin typed wrappers over the C ABI (all tensors on the current device, launches on torch's current stream)"""

    def __init__(self, dev):
        self.lib = _native.load()
        self.dev = dev
        self.stream = torch.cuda.current_stream(dev).cuda_stream

    def f32(self, n):
        return torch.empty((n,), dtype=torch.float32, device=self.dev)

    def pack(self, w, transpose_flip, split, bias=None, scal=None):
        """device-side weight pack: (wpack, scale[cout_p], shift[cout_p], scal); ``scal``: the scalars of an earlier pack of the same
        weight (its max|w| is reused instead of reduced again)"""
        co, ci = w.shape[:2]
        cout_l, cin_l = (ci, co) if transpose_flip else (co, ci)
        cin_p, cout_p = _pad32(cin_l), _pad32(cout_l)
        wpack = torch.empty((9 * (3 if split else 1) * cin_p * cout_p,), dtype=torch.int16, device=self.dev)
        scale, shift = self.f32(cout_p), self.f32(cout_p)
        reuse = scal is not None
        scal = scal if reuse else self.f32(3)
        wc = w.detach()
        wc = wc if wc.is_contiguous() and wc.dtype == torch.float32 else wc.float().contiguous()
        bc = None
        if bias is not None:
            bc = bias.detach()
            bc = bc if bc.is_contiguous() and bc.dtype == torch.float32 else bc.float().contiguous()
        rc = self.lib.nastar_pack_conv_weight_f16(wc.data_ptr(), co, ci, int(transpose_flip), int(split),
                                                  bc.data_ptr() if bc is not None else None, wpack.data_ptr(), scale.data_ptr(),
                                                  shift.data_ptr(), scal.data_ptr(), int(reuse), self.stream)
        _native.check(rc, "nastar_pack_conv_weight_f16")
        return wpack, scale, shift, scal

    _TABLES: dict = {}  # (device, data pointers, sizes) -> device table of nastar_absmax_multi_f32 (weights keep their storage across steps)

    def weight_maxima(self, weights):
        """max|w| of every convolution weight in ONE launch: float [n, 3] whose rows are the ``scal`` triples ``pack`` expects with
        ``reuse`` (row[2] = max|w|).  Split operands only -- the plain fp16 pack needs no scale."""
        ws = [w.detach() for w in weights]
        if any(w.dtype != torch.float32 or not w.is_contiguous() for w in ws):
            return None
        key = (str(self.dev),) + tuple((w.data_ptr(), w.numel()) for w in ws)
        table = self._TABLES.get(key)
        if table is None:
            if len(self._TABLES) > 64:
                self._TABLES.clear()
            table = torch.tensor([[w.data_ptr(), w.numel()] for w in ws], dtype=torch.int64).to(self.dev)
            self._TABLES[key] = table
        scal = torch.empty((len(ws), 3), dtype=torch.float32, device=self.dev)
        rc = self.lib.nastar_absmax_multi_f32(table.data_ptr(), len(ws), scal.data_ptr(), self.stream)
        _native.check(rc, "nastar_absmax_multi_f32")
        return scal

    def pack_all(self, specs, split, scal=None):
        """every weight pack of a step in ONE launch.  specs: [(w, bias or None, transpose_flip, row of `scal`)]; ``scal``: the
        [rows, 3] table of ``weight_maxima`` (split operands) or None.  Returns [(wpack, scale, shift, scal_row)] (views of two flat
        buffers), or None when a tensor is not plain contiguous fp32 (the caller then packs one by one)."""
        ws = [sp[0].detach() for sp in specs]
        bs = [sp[1].detach() if sp[1] is not None else None for sp in specs]
        if any(t is not None and (t.dtype != torch.float32 or not t.is_contiguous()) for t in ws + bs):
            return None
        rows = 1 + max(sp[3] for sp in specs)
        if scal is None:
            scal = torch.empty((rows, 3), dtype=torch.float32, device=self.dev)
        key = ("pack", str(self.dev), bool(split)) + tuple((w.data_ptr(), b.data_ptr() if b is not None else 0, w.shape[0], w.shape[1],
                                                             int(sp[2]), int(sp[3])) for w, b, sp in zip(ws, bs, specs))
        ent = self._TABLES.get(key)
        if ent is None:
            if len(self._TABLES) > 64:
                self._TABLES.clear()
            rows_t, lay, o16, of = [], [], 0, 0
            for w, b, sp in zip(ws, bs, specs):
                co, ci = w.shape[:2]
                cout_l, cin_l = (ci, co) if sp[2] else (co, ci)
                cin_p, cout_p = _pad32(cin_l), _pad32(cout_l)
                n16 = 9 * (3 if split else 1) * cin_p * cout_p
                rows_t.append([w.data_ptr(), b.data_ptr() if b is not None else 0, co, ci, int(sp[2]), o16, of, int(sp[3])])
                lay.append((o16, n16, of, cout_p))
                o16 += n16
                of += 2 * cout_p
            tiles = max(((w.shape[0] + 31) // 32) * ((w.shape[1] + 31) // 32) for w in ws)
            ent = (torch.tensor(rows_t, dtype=torch.int64).to(self.dev), lay, o16, of, tiles)
            self._TABLES[key] = ent
        table, lay, n16_all, nf_all, tiles = ent
        flat16 = torch.empty((n16_all,), dtype=torch.int16, device=self.dev)
        flatf = torch.empty((nf_all,), dtype=torch.float32, device=self.dev)
        rc = self.lib.nastar_pack_conv_weights_multi_f16(table.data_ptr(), len(specs), tiles, int(split), scal.data_ptr(), flat16.data_ptr(),
                                                         flatf.data_ptr(), self.stream)
        _native.check(rc, "nastar_pack_conv_weights_multi_f16")
        return [(flat16[o:o + n], flatf[f:f + c], flatf[f + c:f + 2 * c], scal[sp[3]]) for (o, n, f, c), sp in zip(lay, specs)]

    def bn_fwd(self, z, npix, C, split, gamma, beta, eps, mom, rm, rv):
        """batch statistics of z and the forward BatchNorm coefficients in two launches (partial rows; finish + coefficients):
        (mean, invstd, k2, k3).  Not for SyncBatchNorm (the sums must be all-reduced between the halves)."""
        k2, k3 = self.f32(C), self.f32(C)
        mean = torch.empty((C,), dtype=torch.float64, device=self.dev)
        invstd = torch.empty((C,), dtype=torch.float64, device=self.dev)
        nbytes = int(self.lib.nastar_chan_stats_workspace_bytes(npix, C))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        rc = self.lib.nastar_bn_stats_coef_fwd_f16(z.data_ptr(), npix, C, int(split), gamma.data_ptr(), beta.data_ptr(), float(eps), float(mom),
                                                   rm.data_ptr() if rm is not None else None, rv.data_ptr() if rv is not None else None,
                                                   k2.data_ptr(), k3.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, ws.data_ptr(), nbytes,
                                                   self.stream)
        _native.check(rc, "nastar_bn_stats_coef_fwd_f16")
        return mean, invstd, k2, k3

    def bn_bwd(self, da, z, k2f, k3f, npix, C, split, mean, invstd, gamma, gscale_in, gscale_out, sums_out=None):
        """(sum dy, sum dy z), max|dy| and the backward BatchNorm coefficients in two launches: (dgamma, dbeta, c1, c2, c3); the re-centred
        gradient scale goes to ``gscale_out`` (a DIFFERENT tensor than ``gscale_in``: every workgroup of the finishing kernel reads the latter)"""
        dgamma, dbeta, c1, c2, c3 = (self.f32(C) for _ in range(5))
        nbytes = int(self.lib.nastar_chan_stats_workspace_bytes(npix, C))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        rc = self.lib.nastar_bn_stats_coef_bwd_f16(da.data_ptr(), z.data_ptr(), k2f.data_ptr(), k3f.data_ptr(), npix, C, int(split), mean.data_ptr(),
                                                   invstd.data_ptr(), gamma.data_ptr(), gscale_in.data_ptr(), gscale_out.data_ptr(),
                                                   dgamma.data_ptr(), dbeta.data_ptr(), c1.data_ptr(), c2.data_ptr(), c3.data_ptr(),
                                                   sums_out.data_ptr() if sums_out is not None else None, ws.data_ptr(), nbytes, self.stream)
        _native.check(rc, "nastar_bn_stats_coef_bwd_f16")
        return dgamma, dbeta, c1, c2, c3

    IMG32 = {(32, 64), (64, 128), (128, 256), (256, 128), (128, 64)}

    def conv(self, src, wpack, scale, shift, B, H, W, cin, cout, flags, out=None, out_f32=None, src2=None, c2=0):
        if (H == 32 and W == 32 and (cin, cout) in self.IMG32 and src2 is None and out is not None
                and not (flags & ~(CONV_RELU | CONV_SPLIT)) and B >= 256):
            # large batches of 32x32 maps: the CNN encoder's persistent whole-image kernel (same layouts, ~1.4x the generic kernel;
            # below ~256 maps its one-workgroup-per-CU grid has nothing to amortise)
            rc = self.lib.nastar_conv3x3_img32_f16(src.data_ptr(), wpack.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), B,
                                                   cin, cout, flags, self.stream)
            _native.check(rc, "nastar_conv3x3_img32_f16")
            return
        rc = self.lib.nastar_conv3x3_f16(src.data_ptr(), src2.data_ptr() if src2 is not None else None, wpack.data_ptr(),
                                         scale.data_ptr(), shift.data_ptr(), out.data_ptr() if out is not None else None,
                                         out_f32.data_ptr() if out_f32 is not None else None, B, H, W, cin, c2, cout, flags, 1.0, self.stream)
        _native.check(rc, "nastar_conv3x3_f16")

    def i16(self, n):
        return torch.empty((n,), dtype=torch.int16, device=self.dev)

    def stats(self, u, v, ms, mt, npix, C, split, amax=None):
        """per-channel double sums [C,2]: two-stage form (per-workgroup partials + a fixed-order finishing launch: deterministic)"""
        sums = torch.empty((C, 2), dtype=torch.float64, device=self.dev)
        nbytes = int(self.lib.nastar_chan_stats_workspace_bytes(npix, C))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        rc = self.lib.nastar_chan_stats_f16_ws(u.data_ptr() if u is not None else None, v.data_ptr(),
                                               ms.data_ptr() if ms is not None else None, mt.data_ptr() if mt is not None else None,
                                               sums.data_ptr(), amax.data_ptr() if amax is not None else None, npix, C, int(split),
                                               ws.data_ptr(), nbytes, self.stream)
        _native.check(rc, "nastar_chan_stats_f16_ws")
        return sums

    def affine(self, u, v, k1, k2, k3, ms, mt, out, npix, C, relu, split):
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        rc = self.lib.nastar_chan_affine_f16(p(u), p(v), p(k1), p(k2), p(k3), p(ms), p(mt), out.data_ptr(), npix, C, int(relu),
                                             int(split), self.stream)
        _native.check(rc, "nastar_chan_affine_f16")

    # ---- the 1-channel closing convolution as streams (csrc/nastar_encoder_co1.hip.h) ----
    @staticmethod
    def co1_ok(C: int, c_real: int) -> bool:
        return C == c_real and 8 <= C <= 512 and (C & (C - 1)) == 0

    def co1_ws(self, B, H, W, C):
        n = int(self.lib.nastar_conv3x3_co1_workspace_bytes(B, H, W, C))
        return torch.empty((n,), dtype=torch.uint8, device=self.dev), n

    def conv_co1(self, a, w, bias, B, H, W, C, split, k2=None, k3=None):
        """z = conv(a, w) + bias for the 1-channel closing convolution; k2 / k3: ``a`` is the pre-activation of the block in front and
        the layer's input relu(k2 a + k3) is formed while loading"""
        z = torch.empty((B, H, W), dtype=torch.float32, device=self.dev)
        ws, n = self.co1_ws(B, H, W, C)
        wc, bc = _f32c(w), (_f32c(bias) if bias is not None else None)
        rc = self.lib.nastar_conv3x3_co1_f16(a.data_ptr(), wc.data_ptr(), bc.data_ptr() if bc is not None else None, B, H, W, C, int(split),
                                             k2.data_ptr() if k2 is not None else None, k3.data_ptr() if k3 is not None else None,
                                             z.data_ptr(), ws.data_ptr(), n, self.stream)
        _native.check(rc, "nastar_conv3x3_co1_f16")
        return z

    def wgrad_co1(self, d, a, B, H, W, C, split, k2=None, k3=None):
        dw = torch.empty((1, C, 3, 3), dtype=torch.float32, device=self.dev)
        ws, n = self.co1_ws(B, H, W, C)
        rc = self.lib.nastar_conv3x3_co1_wgrad_f16(d.data_ptr(), a.data_ptr(), B, H, W, C, int(split),
                                                   k2.data_ptr() if k2 is not None else None, k3.data_ptr() if k3 is not None else None,
                                                   dw.data_ptr(), ws.data_ptr(), n, self.stream)
        _native.check(rc, "nastar_conv3x3_co1_wgrad_f16")
        return dw

    def bn_bwd_u1(self, d, wlast, B, H, W, z, k2f, k3f, C, split, mean, invstd, gamma, gscale_in, gscale_out):
        """``bn_bwd`` for the block in front of the closing convolution: da = gscale_in * (its input gradient of d), never stored"""
        dgamma, dbeta, c1, c2, c3 = (self.f32(C) for _ in range(5))
        nbytes = int(self.lib.nastar_chan_stats_workspace_bytes(B * H * W, C))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        rc = self.lib.nastar_bn_stats_coef_bwd_u1_f16(d.data_ptr(), wlast.data_ptr(), B, H, W, z.data_ptr(), k2f.data_ptr(), k3f.data_ptr(), C, int(split),
                                                      mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), gscale_in.data_ptr(), gscale_out.data_ptr(),
                                                      dgamma.data_ptr(), dbeta.data_ptr(), c1.data_ptr(), c2.data_ptr(), c3.data_ptr(), None,
                                                      ws.data_ptr(), nbytes, self.stream)
        _native.check(rc, "nastar_bn_stats_coef_bwd_u1_f16")
        return dgamma, dbeta, c1, c2, c3

    def stats_u1(self, d, wlast, gscale, B, H, W, z, ms, mt, C, split, amax):
        """``stats`` (backward form) with da formed on the fly: double sums [C,2] for the sync-BN all-reduce"""
        sums = torch.empty((C, 2), dtype=torch.float64, device=self.dev)
        nbytes = int(self.lib.nastar_chan_stats_workspace_bytes(B * H * W, C))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        rc = self.lib.nastar_chan_stats_u1_f16_ws(d.data_ptr(), wlast.data_ptr(), gscale.data_ptr(), B, H, W, z.data_ptr(), ms.data_ptr(), mt.data_ptr(),
                                                  sums.data_ptr(), amax.data_ptr(), C, int(split), ws.data_ptr(), nbytes, self.stream)
        _native.check(rc, "nastar_chan_stats_u1_f16_ws")
        return sums

    def affine_u1(self, d, wlast, gscale, B, H, W, z, k1, k2, k3, ms, mt, out, C, split):
        rc = self.lib.nastar_chan_affine_u1_f16(d.data_ptr(), wlast.data_ptr(), gscale.data_ptr(), B, H, W, z.data_ptr(), k1.data_ptr(), k2.data_ptr(),
                                                k3.data_ptr(), ms.data_ptr(), mt.data_ptr(), out.data_ptr(), C, int(split), self.stream)
        _native.check(rc, "nastar_chan_affine_u1_f16")

    def wgrad(self, dz, a, B, H, W, co, ci, co_real, ci_real, split, gscale):
        """dW in torch's [co_real, ci_real, 3, 3] layout, already divided by the device-side gradient scale"""
        dw = torch.empty((co_real, ci_real, 3, 3), dtype=torch.float32, device=self.dev)
        nbytes = int(self.lib.nastar_conv3x3_wgrad_workspace_bytes(B, H, W, co, ci))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        rc = self.lib.nastar_conv3x3_wgrad_f16(dz.data_ptr(), a.data_ptr(), dw.data_ptr(), B, H, W, co, ci, co_real, ci_real, int(split),
                                               1.0, gscale.data_ptr(), ws.data_ptr(), nbytes, self.stream)
        _native.check(rc, "nastar_conv3x3_wgrad_f16")
        return dw


CO1_STREAMS = True  # the 1-channel closing convolution as streams (csrc/nastar_encoder_co1.hip.h); False: padded to 32 channels on the MFMA (A/B, tests)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    return t if t.is_contiguous() and t.dtype == torch.float32 else t.float().contiguous()


def wgrad_segment(W: int) -> int:
    """width of a weight-gradient chunk row (csrc/nastar_conv_wgrad.hip.h: nastar_wgrad_segment): the image width up to 96 pixels, else its
    widest divisor <= 96 when that is at least 64 pixels, else equal RAGGED segments of ceil(W / ceil(W / 96)) pixels (prime widths, 2 x 97, ...:
    the last segment of a row is shorter, its columns beyond the image are staged as zeros)"""
    if W <= 96:
        return W
    for d in range(96, 63, -1):
        if W % d == 0:
            return d
    nseg = (W + 95) // 96
    return (W + nseg - 1) // nseg


def chunk_rows(H: int, W: int) -> int:
    """rows per weight-gradient chunk (csrc/nastar_conv_wgrad.hip.h: nastar_wgrad_chunk_rows); 0 = unsupported shape"""
    W = wgrad_segment(W)
    if W < 2 or W > 96 or H <= 0:
        return 0
    if 64 % W == 0 and H % (64 // W) == 0:
        return 64 // W
    for r in range(96 // W, 0, -1):
        if H % r == 0:
            return r
    return 0


def supported_shape(H: int, W: int, depth: int = 4, pool: bool = False) -> bool:
    """every resolution the stack visits must suit the weight-gradient kernel (<= 96-pixel chunks of whole image rows, or of row segments for
    images wider than 96 pixels: the width then needs a divisor in [2, 96]); pooling stacks halve the resolution after every hidden block.
    (The generic convolution takes any width since round 6: 2-D tiles beyond 126 pixels.)"""
    for l in range(depth + 1):
        h, w = (H >> l, W >> l) if pool else (H, W)
        if pool and l < depth and ((h | w) & 1):
            return False
        if chunk_rows(h, w) == 0:
            return False
    return True


class _CnnTrunk(torch.autograd.Function):
    """(assembled input x0, conv / BatchNorm parameters of the D hidden blocks, last conv) -> z [B,1,h,w] fp32 (raw output of the
    last convolution, bias included).  ``cfg``: dict(split, depth D, pool, shape (B, H, W), eps[D], bns[D] for the running
    statistics).  Hidden block: conv3x3 -> BatchNorm (batch statistics) -> ReLU [-> 2x2 max-pool]."""

    @staticmethod
    def forward(ctx, cfg, x0, *params):
        split, D, pool = cfg["split"], cfg["depth"], cfg["pool"]
        dev = x0.device
        B, H, W = cfg["shape"]
        L = _Lib(dev)
        mult = 2 if split else 1
        sflag = CONV_SPLIT if split else 0
        ws = list(params[0:4 * D + 1:4])             # conv weights of blocks 1..D+1
        bs = list(params[1:4 * D + 2:4])             # conv biases
        gammas = list(params[2:4 * D:4])             # BatchNorm weights of the hidden blocks
        betas = list(params[3:4 * D:4])
        with torch.cuda.device(dev):
            acts, zs, rs, coef, scals = [x0], [], [], [], []
            tracked = []  # the BatchNorm step counters: ONE multi-tensor increment instead of a launch per layer
            ctx.sync_state = SyncBatchNorm.snapshot()  # the backward all-reduces iff this forward did
            sync = ctx.sync_state[0]
            h, w = H, W
            wmax = L.weight_maxima(ws) if split else None  # one launch for all D + 1 weight maxima
            # the 1-channel closing convolution: a stream over its input (31/32 of a padded matrix product would be zeros); where nothing
            # else reads the activations of the block in front of it (no pooling, no test probe),
            # that block's BatchNorm + ReLU is applied while the stream loads its pre-activations and the activation tensor never exists
            co1 = cfg.get("co1", CO1_STREAMS) and ws[D].shape[0] == 1 and L.co1_ok(_pad32(ws[D].shape[1]), ws[D].shape[1])
            fuse_act = co1 and D >= 1 and not pool and cfg.get("debug") is None
            # ... and one for every weight pack of the step: the D + 1 forward forms, then the D input-gradient forms the backward needs --
            # minus the ones the streams make unnecessary: the padded forward pack of the closing convolution, and its input-gradient pack
            # where the BatchNorm backward of the block in front forms that gradient on the fly (no pooling)
            specs = [(ws[l], bs[l], False, l) for l in range(D + 1)] + [(ws[l], None, True, l) for l in range(1, D + 1)]
            unused = ({D} if co1 else set()) | ({2 * D} if (co1 and D >= 1 and not pool) else set())
            got = L.pack_all([sp for i, sp in enumerate(specs) if i not in unused], split, wmax)
            packs = None
            if got is not None:
                it = iter(got)
                packs = [None if i in unused else next(it) for i in range(len(specs))]
            for l in range(D):
                wt = ws[l]
                cout, cin_p = wt.shape[0], _pad32(wt.shape[1])
                npix = B * h * w
                wpack, scale, shift, scal = packs[l] if packs is not None else L.pack(wt, False, split, bs[l], scal=wmax[l] if wmax is not None else None)
                scals.append(scal)
                z = torch.empty((npix * cout * mult,), dtype=torch.int16, device=dev)
                L.conv(acts[-1], wpack, scale, shift, B, h, w, cin_p, cout, sflag, out=z)
                bn = cfg["bns"][l]
                track = bn is not None and bn.track_running_stats and bn.running_mean is not None and not cfg.get("eval_bn")
                mom = 0.0
                if track:  # nn.BatchNorm2d's training-mode side effect (unbiased variance), done inside the coefficient kernel
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
                    tracked.append(bn.num_batches_tracked)
                gam, bet = gammas[l].detach(), betas[l].detach()
                if cfg.get("eval_bn"):
                    # EVAL-mode BatchNorm under autograd (module.eval() with gradients on): the coefficients come from the RUNNING statistics,
                    # nothing is updated; [C]-sized host-side tensor math (a handful of tiny launches on a rare path)
                    invstd = torch.rsqrt(bn.running_var.double() + float(cfg["eps"][l]))
                    mean = bn.running_mean.double().clone()
                    k2 = (gam.double() * invstd).float()
                    k3 = (bet.double() - mean * gam.double() * invstd).float()
                elif not sync:  # partial rows, then finish + coefficients in one kernel
                    mean, invstd, k2, k3 = L.bn_fwd(z, npix, cout, split, gam, bet, cfg["eps"][l], mom, bn.running_mean if track else None,
                                                    bn.running_var if track else None)
                else:  # data parallel: the sums of the GLOBAL batch go through an all-reduce between the halves
                    sums = L.stats(None, z, None, None, npix, cout, split)
                    npix_bn = npix * _sync_sums(sums, state=ctx.sync_state)
                    k2, k3 = L.f32(cout), L.f32(cout)
                    mean = torch.empty((cout,), dtype=torch.float64, device=dev)
                    invstd = torch.empty((cout,), dtype=torch.float64, device=dev)
                    rc = L.lib.nastar_bn_coef_fwd(sums.data_ptr(), gam.data_ptr(), bet.data_ptr(), float(cfg["eps"][l]), npix_bn, float(mom),
                                                  bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                                  k2.data_ptr(), k3.data_ptr(), mean.data_ptr(), invstd.data_ptr(), cout, L.stream)
                    _native.check(rc, "nastar_bn_coef_fwd")
                zs.append(z)
                coef.append((mean, invstd, k2, k3))
                if fuse_act and l == D - 1:
                    acts.append(None)  # relu(k2 z + k3) is formed inside the closing convolution's streams
                    continue
                r = torch.empty_like(z)
                L.affine(None, z, None, k2, k3, None, None, r, npix, cout, True, split)
                if cfg.get("debug") is not None:  # test probe: this block's ReLU mask is [k2 z + k3 > 0], its pool sees r
                    cfg["debug"][f"fwd:{l}"] = (z, k2, k3, r, (B, h, w, cout))
                if pool:
                    a = torch.empty((B * (h // 2) * (w // 2) * cout * mult,), dtype=torch.int16, device=dev)
                    rc = L.lib.nastar_maxpool2x2_f16(r.data_ptr(), a.data_ptr(), B, h, w, cout, int(split), L.stream)
                    _native.check(rc, "nastar_maxpool2x2_f16")
                    rs.append(r)  # the pool's input: its backward needs the arg-max
                    acts.append(a)
                    h, w = h // 2, w // 2
                else:
                    acts.append(r)
            wl = ws[D]
            if co1:  # streams: no padded pack of the closing convolution
                wpackl = scalel = shiftl = None
                scal = wmax[D] if wmax is not None else None
            else:
                wpackl, scalel, shiftl, scal = (packs[D] if packs is not None else  # cout 1 -> 32 (padded channels: zero weights, zero shift)
                                                L.pack(wl, False, split, bs[D], scal=wmax[D] if wmax is not None else None))
            scals.append(scal)
            # the 1-channel closing convolution: a stream over its input (31/32 of a padded matrix product would be zeros)
            if co1 and fuse_act:
                zl = L.conv_co1(zs[D - 1], wl, bs[D], B, h, w, wl.shape[1], split, coef[D - 1][2], coef[D - 1][3])
            elif co1:
                zl = L.conv_co1(acts[-1], wl, bs[D], B, h, w, wl.shape[1], split)
            else:
                zl = torch.empty((B, h, w), dtype=torch.float32, device=dev)
                L.conv(acts[-1], wpackl, scalel, shiftl, B, h, w, _pad32(wl.shape[1]), 32, sflag | CONV_FINAL | CONV_RAW, out_f32=zl)
            ctx.co1, ctx.fuse_act = co1, fuse_act
            if tracked:
                torch._foreach_add_(tracked, 1)
        ctx.cfg = cfg
        ctx.acts, ctx.zs, ctx.rs, ctx.coef, ctx.scals = acts, zs, rs, coef, scals
        ctx.tpacks = packs[D + 1:] if packs is not None else None  # input-gradient packs of blocks 1..D (the weights do not change in between)
        ctx.save_for_backward(*params)
        return zl.unsqueeze(1)

    @staticmethod
    def backward(ctx, dzl):
        cfg = ctx.cfg
        split, D, pool = cfg["split"], cfg["depth"], cfg["pool"]
        params = ctx.saved_tensors
        ws = list(params[0:4 * D + 1:4])
        gammas = list(params[2:4 * D:4])
        B, H, W = cfg["shape"]
        dev = dzl.device
        L = _Lib(dev)
        mult = 2 if split else 1
        sflag = CONV_SPLIT if split else 0
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        with torch.cuda.device(dev):
            h, w = (H >> D, W >> D) if pool else (H, W)   # resolution of the last convolution
            npix = B * h * w
            d = dzl.reshape(npix)
            d = d if d.is_contiguous() and d.dtype == torch.float32 else d.float().contiguous()
            # gradients travel multiplied by a power of two S (device scalar `gscale`, re-centred per block): scaled values peak near
            # 2^10, so fp16 neither overflows nor loses the small terms; S is divided out inside the weight-gradient / coefficient kernels
            gscale, amax = L.f32(1), L.f32(1)
            bias_grads = {}  # eval-mode BatchNorm only: conv-bias gradients of the hidden blocks
            top = D
            # closing convolution as streams: its weight gradient from d itself, its input gradient never stored -- the BatchNorm backward
            # of block D forms it on the fly (not for pooling stacks, whose gradient passes through the max-pool first)
            if getattr(ctx, "co1", False) and D >= 1 and not pool:
                wl = ws[D]
                C = wl.shape[1]
                rc = L.lib.nastar_grad_scale_f32(d.data_ptr(), npix, gscale.data_ptr(), amax.data_ptr(), L.stream)
                _native.check(rc, "nastar_grad_scale_f32")
                def closing_wgrad():
                    if ctx.fuse_act:
                        return L.wgrad_co1(d, ctx.zs[D - 1], B, h, w, C, split, ctx.coef[D - 1][2], ctx.coef[D - 1][3])
                    return L.wgrad_co1(d, ctx.acts[D], B, h, w, C, split)
                # eval-mode BatchNorm = the batch-statistics closed form in the limit of infinitely many pixels (the mean terms vanish): the
                # unfused path below with npix -> 1e30 and the running statistics as mean / invstd; dgamma / dbeta need the same sums
                eval_bn = bool(cfg.get("eval_bn"))
                unfused = ctx.sync_state[0] or eval_bn
                if not unfused:
                    grads[4 * D] = closing_wgrad()
                grads[4 * D + 1] = torch.empty_like(params[4 * D + 1])
                z = ctx.zs[D - 1]
                mean, invstd, k2f, k3f = ctx.coef[D - 1]
                wlc = _f32c(wl)
                gs_new = L.f32(1)
                if not unfused:
                    dgamma, dbeta, c1, c2, c3 = L.bn_bwd_u1(d, wlc, B, h, w, z, k2f, k3f, C, split, mean, invstd, gammas[D - 1].detach(), gscale, gs_new)
                else:  # data parallel: the sums of the GLOBAL batch (all-reduced between the statistics and the coefficients); or eval mode
                    sums = L.stats_u1(d, wlc, gscale, B, h, w, z, k2f, k3f, C, split, amax)
                    work = _sync_sums_begin(sums, gscale, ctx.sync_state)
                    grads[4 * D] = closing_wgrad()  # beside the collective, which it does not need
                    world = _sync_sums_end(work, sums, gscale, ctx.sync_state)
                    dgamma, dbeta, c1, c2, c3 = (L.f32(C) for _ in range(5))
                    rc = L.lib.nastar_bn_coef_bwd_io(sums.data_ptr(), amax.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                     gammas[D - 1].detach().data_ptr(), (1 << 62) if eval_bn else npix * world, gscale.data_ptr(), gs_new.data_ptr(),
                                                     dgamma.data_ptr(), dbeta.data_ptr(), c1.data_ptr(), c2.data_ptr(), c3.data_ptr(), C, L.stream)
                    _native.check(rc, "nastar_bn_coef_bwd_io")
                    if world > 1:  # the flat gradient all-reduce AVERAGES over the ranks
                        dgamma /= world
                        dbeta /= world
                grads[4 * (D - 1) + 2] = dgamma
                grads[4 * (D - 1) + 3] = dbeta
                if eval_bn:  # the conv bias in front of an EVAL-mode BatchNorm has a gradient: sum_p dz = gamma invstd sum_p dy
                    nb = params[4 * (D - 1) + 1].numel()
                    bias_grads[D - 1] = (dbeta[:nb].double() * gammas[D - 1].detach().double()[:nb] * invstd[:nb]).float()
                dzb = torch.empty((npix * C * mult,), dtype=torch.int16, device=dev)
                L.affine_u1(d, wlc, gscale, B, h, w, z, c1, c2, c3, k2f, k3f, dzb, C, split)
                gscale = gs_new
                cur_co = C
                top = D - 1
            else:
                dzb = torch.empty((npix * 32 * mult,), dtype=torch.int16, device=dev)
                rc = L.lib.nastar_grad_seed_f16(d.data_ptr(), npix, int(split), dzb.data_ptr(), gscale.data_ptr(), amax.data_ptr(), L.stream)
                _native.check(rc, "nastar_grad_seed_f16")
                cur_co = 32  # padded channel count of the current dz
            for l in range(top, -1, -1):
                wt = ws[l]
                cout, cin = wt.shape[:2]
                cin_p = _pad32(cin)
                # data-parallel (sync) steps launch this layer's weight gradient BESIDE the all-reduce of the next BatchNorm backward's
                # sums (below): it needs neither, and a small collective costs ~80 us of latency even in a 1-rank group
                eval_bn = bool(cfg.get("eval_bn"))
                unfused = ctx.sync_state[0] or eval_bn
                late_wgrad = unfused and l > 0
                wg_h, wg_w = h, w  # (the pooling stacks change h, w before the deferred launch)
                if not late_wgrad:
                    grads[4 * l] = L.wgrad(dzb, ctx.acts[l], B, h, w, cur_co, cin_p, cout, cin, split, gscale)
                grads[4 * l + 1] = torch.empty_like(params[4 * l + 1])             # conv bias in front of a BatchNorm: exactly 0 (zeroed below)
                if l == 0:
                    break
                # input gradient: the same convolution with W^T flipped (cin <-> cout; cout 1 of the last block padded to 32 inputs)
                wpack, scale, shift, _ = ctx.tpacks[l - 1] if (ctx.tpacks is not None and ctx.tpacks[l - 1] is not None) else L.pack(wt, True, split, scal=ctx.scals[l])
                da = torch.empty((npix * cin_p * mult,), dtype=torch.int16, device=dev)
                L.conv(dzb, wpack, scale, shift, B, h, w, cur_co, cin_p, sflag, out=da)
                C = cin_p
                if pool:  # da is the gradient w.r.t. the pooled activations: route it to each window's arg-max at the finer resolution
                    h, w = h * 2, w * 2
                    npix = B * h * w
                    dr = torch.empty((npix * C * mult,), dtype=torch.int16, device=dev)
                    rc = L.lib.nastar_maxpool2x2_bwd_f16(ctx.rs[l - 1].data_ptr(), da.data_ptr(), dr.data_ptr(), B, h, w, C, int(split), L.stream)
                    _native.check(rc, "nastar_maxpool2x2_bwd_f16")
                    da = dr
                # ReLU mask + BatchNorm backward of hidden block l (pre-activation zs[l-1])
                z = ctx.zs[l - 1]
                mean, invstd, k2f, k3f = ctx.coef[l - 1]
                if not unfused:  # (sum dy, sum dy z) * S, max|dy| * S: partial rows, then finish + coefficients in one kernel
                    gs_new = L.f32(1)  # NOT in place: the finishing kernel has many workgroups, all of which read the incoming scale
                    dgamma, dbeta, c1, c2, c3 = L.bn_bwd(da, z, k2f, k3f, npix, C, split, mean, invstd, gammas[l - 1].detach(), gscale, gs_new)
                    gscale = gs_new
                else:
                    sums = L.stats(da, z, k2f, k3f, npix, C, split, amax=amax)
                    work = _sync_sums_begin(sums, gscale, ctx.sync_state)
                    # ... the collective is in flight: this layer's weight gradient (dz of block l+1 x activations of block l) runs now
                    grads[4 * l] = L.wgrad(dzb, ctx.acts[l], B, wg_h, wg_w, cur_co, cin_p, cout, cin, split, gscale)
                    world = _sync_sums_end(work, sums, gscale, ctx.sync_state)
                    dgamma, dbeta, c1, c2, c3 = (L.f32(C) for _ in range(5))
                    rc = L.lib.nastar_bn_coef_bwd(sums.data_ptr(), amax.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                  gammas[l - 1].detach().data_ptr(), (1 << 62) if eval_bn else npix * world, gscale.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                  c1.data_ptr(), c2.data_ptr(), c3.data_ptr(), C, L.stream)
                    _native.check(rc, "nastar_bn_coef_bwd")
                    if world > 1:  # the kernel formed dgamma / dbeta from the GLOBAL sums; the flat gradient all-reduce AVERAGES over ranks
                        dgamma /= world
                        dbeta /= world
                grads[4 * (l - 1) + 2] = dgamma
                grads[4 * (l - 1) + 3] = dbeta
                if eval_bn:
                    nb = params[4 * (l - 1) + 1].numel()
                    bias_grads[l - 1] = (dbeta[:nb].double() * gammas[l - 1].detach().double()[:nb] * invstd[:nb]).float()
                dzb = torch.empty_like(da)
                L.affine(da, z, c1, c2, c3, k2f, k3f, dzb, npix, C, False, split)
                cur_co = C
            torch._foreach_zero_([grads[4 * l + 1] for l in range(D + 1)])  # one launch for all of them
            if cfg.get("eval_bn"):  # ... except in eval mode, where the biases in front of a BatchNorm on running statistics do get gradients
                for l, bg in bias_grads.items():
                    grads[4 * l + 1] = bg
                grads[4 * D + 1] = dzl.float().sum().reshape(1)
        return (None, None) + tuple(grads)


_CONSTS: dict = {}


def _const_tensor(value: float, device: torch.device) -> torch.Tensor:
    """a cached [1] fp32 device tensor for an encoder whose ``const`` is the plain float 1.0 (reference encoder.py:24-27)"""
    key = (value, device.type, device.index)
    if key not in _CONSTS:
        _CONSTS[key] = torch.full((1,), value, dtype=torch.float32, device=device)
    return _CONSTS[key]


class _LastBlock(torch.autograd.Function):
    """cost = const * sigmoid(BatchNorm1(z)) for the closing 1-channel block (reference encoder.py:60-97 last block + :32-34), batch
    statistics, on two launches each way (``nastar_bn1_*``) instead of ~35 framework launches.  Inputs: z [B,1,h,w] fp32 (raw output of
    the last convolution), the BatchNorm's weight / bias ([1] each), ``const`` as a [1] tensor (parameter or plain).  Data-parallel
    training (``SyncBatchNorm``): the partial sums are reduced to one row, all-reduced, and the kernels take the global sums."""

    @staticmethod
    def forward(ctx, z, gamma, beta, cmul, eps, momentum, running_mean, running_var):
        dev = z.device
        lib = _native.load()
        st = torch.cuda.current_stream(dev).cuda_stream
        zc = z if z.is_contiguous() and z.dtype == torch.float32 else z.float().contiguous()
        n = zc.numel()
        nparts = int(lib.nastar_bn1_parts(n))
        part = torch.empty((nparts, 2), dtype=torch.float64, device=dev)
        cost = torch.empty_like(zc)
        stat = torch.empty((2,), dtype=torch.float64, device=dev)
        g, b, c = gamma.detach(), beta.detach(), cmul.detach()
        eval_bn = momentum is None  # (cnn_train_forward's signal: module.eval() with gradients on -- BatchNorm on its running statistics)
        ctx.eval_bn = eval_bn
        with torch.cuda.device(dev):
            world = 1
            ctx.sync_state = SyncBatchNorm.snapshot()
            if eval_bn:
                # one fabricated row of "sums" over n elements whose mean / variance are the running statistics: the kernel's own arithmetic
                # then normalises with exactly those; nothing is updated (momentum 0, no running pointers)
                rm, rv = running_mean.double().reshape(1), running_var.double().reshape(1)
                part = (torch.stack((rm, rv + rm * rm), dim=1) * float(n)).contiguous()
                _native.check(lib.nastar_bn1_sigmoid_fwd(zc.data_ptr(), n, part.data_ptr(), 1, float(n), g.data_ptr(), b.data_ptr(), float(eps), c.data_ptr(), 0.0,
                                                         None, None, cost.data_ptr(), stat.data_ptr(), st), "nastar_bn1_sigmoid_fwd")
                ctx.save_for_backward(zc, g, b, c, stat)
                ctx.n_total = 1e30  # backward: the batch-mean terms vanish -- eval-mode BatchNorm's dz = gamma invstd dy
                ctx.world = 1
                ctx.sync_state = (False, None, 1)
                ctx.set_materialize_grads(False)
                return cost
            _native.check(lib.nastar_bn1_fwd_partial(zc.data_ptr(), n, part.data_ptr(), st), "nastar_bn1_fwd_partial")
            if ctx.sync_state[0]:
                part = part.sum(0, keepdim=True)
                world = _sync_sums(part, state=ctx.sync_state)
                nparts = 1
            _native.check(lib.nastar_bn1_sigmoid_fwd(zc.data_ptr(), n, part.data_ptr(), nparts, float(n * world), g.data_ptr(), b.data_ptr(),
                                                     float(eps), c.data_ptr(), float(momentum),
                                                     running_mean.data_ptr() if running_mean is not None else None,
                                                     running_var.data_ptr() if running_var is not None else None, cost.data_ptr(),
                                                     stat.data_ptr(), st), "nastar_bn1_sigmoid_fwd")
        ctx.save_for_backward(zc, g, b, c, stat)
        ctx.n_total = float(n * world)
        ctx.world = world
        ctx.set_materialize_grads(False)
        return cost

    @staticmethod
    def backward(ctx, dcost):
        if dcost is None:
            return (None,) * 8
        zc, g, b, c, stat = ctx.saved_tensors
        dev = zc.device
        lib = _native.load()
        st = torch.cuda.current_stream(dev).cuda_stream
        d = dcost if dcost.is_contiguous() and dcost.dtype == torch.float32 else dcost.float().contiguous()
        n = zc.numel()
        nparts = int(lib.nastar_bn1_parts(n))
        part = torch.empty((nparts, 3), dtype=torch.float64, device=dev)
        dz = torch.empty_like(zc)
        dgamma, dbeta, dconst = (torch.empty((1,), dtype=torch.float32, device=dev) for _ in range(3))
        with torch.cuda.device(dev):
            _native.check(lib.nastar_bn1_sigmoid_bwd_partial(zc.data_ptr(), d.data_ptr(), n, stat.data_ptr(), g.data_ptr(), b.data_ptr(),
                                                             c.data_ptr(), part.data_ptr(), st), "nastar_bn1_sigmoid_bwd_partial")
            if ctx.sync_state[0]:  # as the forward did (ctx.n_total counts the GLOBAL pixels exactly then)
                part = part.sum(0, keepdim=True)
                _sync_sums(part, state=ctx.sync_state)
                nparts = 1
            _native.check(lib.nastar_bn1_sigmoid_bwd(zc.data_ptr(), d.data_ptr(), n, stat.data_ptr(), g.data_ptr(), b.data_ptr(), c.data_ptr(),
                                                     part.data_ptr(), nparts, ctx.n_total, dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                     dconst.data_ptr(), st), "nastar_bn1_sigmoid_bwd")
        if ctx.world > 1:  # formed from the GLOBAL sums; the flat gradient all-reduce averages over ranks
            dgamma /= ctx.world
            dbeta /= ctx.world
            dconst /= ctx.world
        return dz, dgamma, dbeta, dconst, None, None, None, None


def _structure(cnn: nn.Module):
    """(convs, bns, pool, depth) of a conv3x3 -> BatchNorm -> ReLU [-> max-pool] stack closed by a 1-channel conv3x3 + BatchNorm, or None
    if the module is anything else (channel counts must be 32, 64, ... powers of two times 32: the streaming kernels' lane layout)"""
    layers = list(cnn.model)
    convs = [m for m in layers if isinstance(m, nn.Conv2d)]
    bns = [m for m in layers if isinstance(m, nn.BatchNorm2d)]
    pool = any(isinstance(m, nn.MaxPool2d) for m in layers)
    D = len(convs) - 1
    ok_c = lambda c: c >= 32 and c <= 2048 and (c & (c - 1)) == 0  # noqa: E731
    if (D < 1 or len(bns) != D + 1 or convs[-1].out_channels != 1 or not all(ok_c(c.out_channels) for c in convs[:-1])
            or any(c.kernel_size != (3, 3) or c.padding != (1, 1) or c.stride != (1, 1) or c.bias is None for c in convs)
            or convs[0].in_channels > 32
            or any(not isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.MaxPool2d)) for m in layers)):
        return None
    return convs, bns, pool, D


def supported(cnn: nn.Module, H: int, W: int) -> bool:
    """can ``cnn_train_forward`` run this encoder on H x W inputs?  (``NeuralAstar.encode`` asks before leaving torch.nn)"""
    st = _structure(cnn)
    return st is not None and supported_shape(H, W, st[3], st[2])


def _assemble_input(map_designs, start_maps, goal_maps, plus, split, L) -> torch.Tensor:
    """x0 [B,H,W,32 (x2)] fp16 NHWC of NeuralAstar.encode's input (reference astar.py:171-177)"""
    B, C, H, W = map_designs.shape
    dev = map_designs.device
    mult = 2 if split else 1
    if C == 1 and (not plus or start_maps.shape[-2:] == map_designs.shape[-2:]):
        m = map_designs[:, 0].contiguous()
        s = start_maps[:, 0].contiguous() if plus else m
        g = goal_maps[:, 0].contiguous() if plus else m
        x0 = torch.empty((B * H * W * 32 * mult,), dtype=torch.int16, device=dev)
        rc = L.lib.nastar_encoder_prep_f16(m.data_ptr(), s.data_ptr() if plus else None, g.data_ptr() if plus else None, int(plus),
                                           B * H * W, 32, int(split), x0.data_ptr(), L.stream)
        _native.check(rc, "nastar_encoder_prep_f16")
        return x0
    # multi-channel images (WarCraft: RGB + nearest-upsampled start + goal): a handful of tensor ops on the small input
    x = map_designs
    if plus:
        sg = start_maps + goal_maps
        if sg.shape[-2:] != x.shape[-2:]:
            sg = nn.functional.interpolate(sg, size=x.shape[-2:], mode="nearest")
        x = torch.cat((x, sg), dim=1)
    x = x.permute(0, 2, 3, 1).float()
    xp = torch.zeros((B, H, W, 32), dtype=torch.float32, device=dev)
    xp[..., :x.shape[-1]] = x
    hi = xp.to(torch.float16)
    if split:
        hi = torch.cat((hi, (xp - hi.float()).to(torch.float16)), dim=-1)
    return hi.contiguous().view(torch.int16).reshape(-1)


def cnn_train_forward(cnn: nn.Module, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor, plus: bool,
                      precision: str = "f16x3") -> torch.Tensor:
    """``cnn(cat(map, start + goal))`` for a ``planner.encoder.CNN`` / ``CNNDownSize`` under autograd, differentiable w.r.t. every encoder
    parameter, on the MI355X kernels: in TRAINING mode (batch-statistics BatchNorm, running statistics updated) or -- round 6 -- in EVAL mode
    (``module.eval()`` with gradients on: BatchNorm on its running statistics, nothing updated).  Returns the cost map [B,1,h,w] fp32
    (h, w = H, W >> depth for the pooling stack)."""
    st = _structure(cnn)
    if st is None:
        raise NotImplementedError("conv3x3 -> BatchNorm -> ReLU [-> max-pool] blocks with 32 * 2^k channels, closed by a 1-channel "
                                  "conv3x3 + BatchNorm")
    convs, bns, pool, D = st
    B, _, H, W = map_designs.shape
    if not supported_shape(H, W, D, pool):
        raise NotImplementedError("map size not supported by the training kernels (see encoder_train.supported_shape)")
    params = []
    for l in range(D + 1):
        params += [convs[l].weight, convs[l].bias, bns[l].weight, bns[l].bias]
    if any(p.dtype != torch.float32 or not p.is_contiguous() for p in params):
        raise NotImplementedError("fp32 contiguous parameters expected")
    split = precision == "f16x3"
    eval_bn = not cnn.training  # module.eval() with gradients on: BatchNorm on its running statistics (no update), still differentiable
    if eval_bn and any(bn.running_mean is None or bn.weight is None for bn in bns):
        raise NotImplementedError("eval-mode BatchNorm without running statistics / affine parameters")
    cfg = {"split": split, "depth": D, "pool": pool, "shape": (B, H, W), "eps": [bn.eps for bn in bns[:D]], "bns": bns[:D],
           "debug": getattr(cnn, "_nastar_debug", None), "eval_bn": eval_bn}
    with torch.cuda.device(map_designs.device):
        x0 = _assemble_input(map_designs, start_maps, goal_maps, plus, split, _Lib(map_designs.device))
    zl = _CnnTrunk.apply(cfg, x0, *params[:4 * D + 2])
    # last block: 1-channel BatchNorm (batch statistics) + sigmoid * const as ONE autograd node on two launches each way (the plain
    # tensor form was ~35 framework launches per step; MIOpen's spatial BatchNorm kernels are slow on a single channel)
    bnl = bns[D]
    if eval_bn:
        const = cnn.const if isinstance(cnn.const, torch.Tensor) else _const_tensor(float(cnn.const), zl.device)
        if const.numel() != 1 or const.dtype != torch.float32:
            y = (zl - bnl.running_mean) * torch.rsqrt(bnl.running_var + bnl.eps) * bnl.weight + bnl.bias
            return torch.sigmoid(y) * cnn.const
        return _LastBlock.apply(zl, bnl.weight, bnl.bias, const.reshape(1), bnl.eps, None, bnl.running_mean, bnl.running_var)  # momentum None = eval
    track = bnl.track_running_stats and bnl.running_mean is not None
    mom = 0.0
    if track:
        mom = bnl.momentum if bnl.momentum is not None else 1.0 / float(int(bnl.num_batches_tracked) + 1)
        bnl.num_batches_tracked += 1
    const = cnn.const if isinstance(cnn.const, torch.Tensor) else _const_tensor(float(cnn.const), zl.device)
    if bnl.weight is None or const.numel() != 1 or const.dtype != torch.float32:  # BatchNorm without affine / exotic const: tensor ops
        if SyncBatchNorm.active():
            y, mean, var = _SyncBatchNorm1.apply(zl, bnl.eps)
        else:
            var, mean = torch.var_mean(zl, unbiased=False)
            y = (zl - mean) * torch.rsqrt(var + bnl.eps)
        if bnl.weight is not None:
            y = y * bnl.weight + bnl.bias
        if track:
            with torch.no_grad():
                n = zl.numel() * SyncBatchNorm.world()
                bnl.running_mean.mul_(1 - mom).add_(mean.detach() * mom)
                bnl.running_var.mul_(1 - mom).add_(var.detach() * (n / max(n - 1, 1)) * mom)
        return torch.sigmoid(y) * cnn.const
    return _LastBlock.apply(zl, bnl.weight, bnl.bias, const.reshape(1), bnl.eps, mom, bnl.running_mean if track else None,
                            bnl.running_var if track else None)


# ---- U-Net (vgg16_bn) training: the same kernels over the launch plan of encoder_hip.unet_layer_plan ------------------------------------
class _UnetTrunk(torch.autograd.Function):
    """(assembled input x0, parameters in plan order) -> raw head output [B,1,H,W] fp32 of a ``planner.encoder.VggUnet`` in training mode.
    ``cfg``: dict(split, shape, plan) where plan = list of dict steps (kind conv / pool, names, channel counts, flags, div, parameter
    slots).  Gradients of tensors with several consumers (skip features) are summed with ``nastar_grad_add_f16``: every gradient tensor
    carries its own power-of-two scale."""

    @staticmethod
    def forward(ctx, cfg, x0, *params):
        from .encoder_hip import CONV_UPSAMPLE
        split = cfg["split"]
        B, H, W = cfg["shape"]
        dev = x0.device
        L = _Lib(dev)
        mult = 2 if split else 1
        sflag = CONV_SPLIT if split else 0
        acts = {"x0": (x0, 32)}
        saved = []
        tracked = []  # the BatchNorm step counters: ONE multi-tensor increment instead of a launch per layer
        ctx.sync_state = SyncBatchNorm.snapshot()  # the backward all-reduces iff this forward did
        sync = ctx.sync_state[0]
        out = None
        with torch.cuda.device(dev):
            conv_steps = [st for st in cfg["plan"] if st["kind"] != "pool"]
            wmax = L.weight_maxima([params[st["w"]] for st in conv_steps]) if split else None  # one launch for all weight maxima
            wrow = {id(st): k for k, st in enumerate(conv_steps)}
            # every weight pack of the step in one launch: the forward forms in plan order, then the input-gradient forms
            tsteps = [st for st in conv_steps if st["src"] != "x0"]
            packs = L.pack_all([(params[st["w"]], params[st["b"]] if st["b"] is not None else None, False, wrow[id(st)]) for st in conv_steps]
                               + [(params[st["w"]], None, True, wrow[id(st)]) for st in tsteps], split, wmax)
            tpack = {id(st): packs[len(conv_steps) + k] for k, st in enumerate(tsteps)} if packs is not None else None
            for st in cfg["plan"]:
                h, w = H // st["div"], W // st["div"]
                if st["kind"] == "pool":
                    src, C = acts[st["src"]]
                    hi, wi = H // st["div_in"], W // st["div_in"]
                    a = L.i16(B * (hi // 2) * (wi // 2) * C * mult)
                    _native.check(L.lib.nastar_maxpool2x2_f16(src.data_ptr(), a.data_ptr(), B, hi, wi, C, int(split), L.stream), "nastar_maxpool2x2_f16")
                    acts[st["dst"]] = (a, C)
                    saved.append(None)
                    if cfg.get("debug") is not None:  # dev / test probe: what this pool saw (its arg-max decisions)
                        cfg["debug"]["fwd:" + st["dst"]] = (src, (B, hi, wi, C))
                    continue
                wt = params[st["w"]]
                bias = params[st["b"]] if st["b"] is not None else None
                src, c1 = acts[st["src"]]
                src2, c2 = acts[st["skip"]] if st["skip"] is not None else (None, 0)
                ups = CONV_UPSAMPLE if st["ups"] else 0
                npix = B * h * w
                wpack, scale, shift, scal = (packs[wrow[id(st)]] if packs is not None else
                                             L.pack(wt, False, split, bias, scal=wmax[wrow[id(st)]] if wmax is not None else None))
                if st["final"]:
                    out = torch.empty((B, h, w), dtype=torch.float32, device=dev)
                    L.conv(src, wpack, scale, shift, B, h, w, c1, 32, sflag | CONV_FINAL | CONV_RAW, out_f32=out)
                    saved.append({"scal": scal})
                    continue
                cout_r = wt.shape[0]
                cout = _pad32(cout_r)  # (encoder_depth = 5 ends in a 16-channel decoder block: the packed weights, z and the BatchNorm vectors carry
                #                         16 zero channels more -- gamma = beta = 0 there, so they stay exactly zero through the ReLU)
                z = L.i16(npix * cout * mult)
                L.conv(src, wpack, scale, shift, B, h, w, c1, cout, sflag | ups, out=z, src2=src2, c2=c2)
                bn = st["bn"]
                eval_bn = bool(cfg.get("eval_bn"))
                track = bn.track_running_stats and bn.running_mean is not None and not eval_bn
                mom = 0.0
                if track:
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(int(bn.num_batches_tracked) + 1)
                    tracked.append(bn.num_batches_tracked)
                padc = cout - cout_r
                gam_v, bet_v = params[st["g"]].detach(), params[st["be"]].detach()
                rm_v, rv_v = (bn.running_mean, bn.running_var) if (track or eval_bn) else (None, None)
                if padc:  # padded copies of the per-channel vectors (the running statistics are copied back below)
                    gam_v, bet_v = torch.nn.functional.pad(gam_v, (0, padc)), torch.nn.functional.pad(bet_v, (0, padc))
                    if rm_v is not None:
                        rm_v, rv_v = torch.nn.functional.pad(rm_v, (0, padc)), torch.nn.functional.pad(rv_v, (0, padc), value=1.0)
                if eval_bn:
                    # EVAL-mode BatchNorm under autograd (module.eval() with gradients on): coefficients from the RUNNING statistics, nothing is
                    # updated; [C]-sized host-side tensor math (as in _CnnTrunk)
                    gam, bet = gam_v.double(), bet_v.double()
                    invstd = torch.rsqrt(rv_v.double() + float(bn.eps))
                    mean = rm_v.double().clone()
                    k2 = (gam * invstd).float()
                    k3 = (bet - mean * gam * invstd).float()
                elif not sync:  # partial rows, then finish + coefficients in one kernel
                    mean, invstd, k2, k3 = L.bn_fwd(z, npix, cout, split, gam_v, bet_v, bn.eps, mom, rm_v if track else None, rv_v if track else None)
                else:  # data parallel: statistics of the GLOBAL batch (all-reduce between the halves)
                    sums = L.stats(None, z, None, None, npix, cout, split)
                    npix_bn = npix * _sync_sums(sums, state=ctx.sync_state)
                    k2, k3 = L.f32(cout), L.f32(cout)
                    mean = torch.empty((cout,), dtype=torch.float64, device=dev)
                    invstd = torch.empty((cout,), dtype=torch.float64, device=dev)
                    rc = L.lib.nastar_bn_coef_fwd(sums.data_ptr(), gam_v.data_ptr(), bet_v.data_ptr(),
                                                  float(bn.eps), npix_bn, float(mom), rm_v.data_ptr() if track else None,
                                                  rv_v.data_ptr() if track else None, k2.data_ptr(), k3.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), cout, L.stream)
                    _native.check(rc, "nastar_bn_coef_fwd")
                if padc and track:  # the kernels updated the padded copies
                    bn.running_mean.copy_(rm_v[:cout_r])
                    bn.running_var.copy_(rv_v[:cout_r])
                a = L.i16(npix * cout * mult)
                L.affine(None, z, None, k2, k3, None, None, a, npix, cout, True, split)
                acts[st["dst"]] = (a, cout)
                saved.append({"z": z, "coef": (mean, invstd, k2, k3), "scal": scal, "gam": gam_v})
                if cfg.get("debug") is not None:  # ... and the ReLU mask of this block: [k2 z + k3 > 0]
                    cfg["debug"]["fwd:" + st["dst"]] = (z, k2, k3, (B, h, w, cout))
            if tracked:
                torch._foreach_add_(tracked, 1)
        ctx.cfg, ctx.acts, ctx.saved, ctx.tpack = cfg, acts, saved, tpack
        ctx.save_for_backward(*params)
        return out.unsqueeze(1)

    @staticmethod
    def backward(ctx, dz):
        cfg = ctx.cfg
        split = cfg["split"]
        B, H, W = cfg["shape"]
        params = ctx.saved_tensors
        acts, saved = ctx.acts, ctx.saved
        dev = dz.device
        L = _Lib(dev)
        mult = 2 if split else 1
        sflag = CONV_SPLIT if split else 0
        grads_p: List[Optional[torch.Tensor]] = [None] * len(params)
        grads = {}  # activation name -> (gradient buffer, device scale)

        def accumulate(name, buf, S, npix, C):
            if name not in grads:
                grads[name] = (buf, S)
                return
            b0, S0 = grads[name]
            out, So = torch.empty_like(buf), L.f32(1)
            rc = L.lib.nastar_grad_add_f16(b0.data_ptr(), S0.data_ptr(), buf.data_ptr(), S.data_ptr(), out.data_ptr(), So.data_ptr(), npix, C,
                                           int(split), L.stream)
            _native.check(rc, "nastar_grad_add_f16")
            grads[name] = (out, So)

        zero_bias = []
        with torch.cuda.device(dev):
            amax = L.f32(1)
            for idx in range(len(cfg["plan"]) - 1, -1, -1):
                st, sv = cfg["plan"][idx], saved[idx]
                h, w = H // st["div"], W // st["div"]
                npix = B * h * w
                if st["kind"] == "pool":
                    g, S = grads.pop(st["dst"])
                    src, C = acts[st["src"]]
                    hi, wi = H // st["div_in"], W // st["div_in"]
                    dr = L.i16(B * hi * wi * C * mult)
                    _native.check(L.lib.nastar_maxpool2x2_bwd_f16(src.data_ptr(), g.data_ptr(), dr.data_ptr(), B, hi, wi, C, int(split), L.stream),
                                  "nastar_maxpool2x2_bwd_f16")
                    accumulate(st["src"], dr, S, B * hi * wi, C)
                    continue
                wt = params[st["w"]]
                cout_r, cin = wt.shape[:2]
                cout = cout_r if st["final"] else _pad32(cout_r)  # buffers of a hidden block carry whole 32-channel groups (16 -> 32: zeros)
                src, c1 = acts[st["src"]]
                src2, c2 = acts[st["skip"]] if st["skip"] is not None else (None, 0)
                if st["final"]:
                    d = dz.reshape(npix)
                    d = d if d.is_contiguous() and d.dtype == torch.float32 else d.float().contiguous()
                    S = L.f32(1)
                    dzb = L.i16(npix * 32 * mult)
                    _native.check(L.lib.nastar_grad_seed_f16(d.data_ptr(), npix, int(split), dzb.data_ptr(), S.data_ptr(), amax.data_ptr(), L.stream),
                                  "nastar_grad_seed_f16")
                    cur_co = 32
                    if st["b"] is not None:
                        grads_p[st["b"]] = d.sum().reshape(1)  # no BatchNorm behind the head: its bias has a real gradient
                else:
                    g, S_in = grads.pop(st["dst"])
                    if cfg.get("debug") is not None:  # dev probe: the gradient w.r.t. this block's output, as it arrives
                        cfg["debug"][st["dst"]] = (g.clone(), S_in.clone(), (B, h, w, cout))
                    S = L.f32(1)  # the BatchNorm backward re-centres the scale: S_in (possibly shared with a skip branch) -> S
                    z = sv["z"]
                    mean, invstd, k2f, k3f = sv["coef"]
                    sums = None
                    eval_bn = bool(cfg.get("eval_bn"))
                    if not ctx.sync_state[0] and not eval_bn:  # partial rows, then finish + coefficients in one kernel
                        if cfg.get("debug") is not None:
                            sums = torch.empty((cout, 2), dtype=torch.float64, device=dev)
                        dgamma, dbeta, c1v, c2v, c3v = L.bn_bwd(g, z, k2f, k3f, npix, cout, split, mean, invstd, sv["gam"], S_in, S, sums_out=sums)
                    else:
                        sums = L.stats(g, z, k2f, k3f, npix, cout, split, amax=amax)
                        world = _sync_sums(sums, S_in, ctx.sync_state)
                        dgamma, dbeta, c1v, c2v, c3v = (L.f32(cout) for _ in range(5))
                        # (eval mode: BatchNorm on its running statistics = the batch-statistics closed form with infinitely many pixels)
                        rc = L.lib.nastar_bn_coef_bwd_io(sums.data_ptr(), amax.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                         sv["gam"].data_ptr(), (1 << 62) if eval_bn else npix * world, S_in.data_ptr(), S.data_ptr(),
                                                         dgamma.data_ptr(), dbeta.data_ptr(), c1v.data_ptr(), c2v.data_ptr(), c3v.data_ptr(), cout,
                                                         L.stream)
                        _native.check(rc, "nastar_bn_coef_bwd_io")
                        if world > 1:  # formed from the GLOBAL sums; the flat gradient all-reduce averages over ranks
                            dgamma /= world
                            dbeta /= world
                    if cfg.get("debug") is not None:
                        cfg["debug"][st["dst"] + ":bn"] = (z, k2f, k3f, dbeta.clone(), dgamma.clone(), sums.clone(), S_in.clone())
                    grads_p[st["g"]], grads_p[st["be"]] = (dgamma, dbeta) if cout == cout_r else (dgamma[:cout_r].contiguous(), dbeta[:cout_r].contiguous())
                    dzb = L.i16(npix * cout * mult)
                    L.affine(g, z, c1v, c2v, c3v, k2f, k3f, dzb, npix, cout, False, split)
                    cur_co = cout
                    if st["b"] is not None and eval_bn:
                        # ... whose conv bias DOES have a gradient then: sum_p dz = gamma invstd sum_p dy
                        nb = params[st["b"]].numel()
                        grads_p[st["b"]] = (dbeta[:nb].double() * sv["gam"].double()[:nb] * invstd[:nb]).float()
                    elif st["b"] is not None:
                        grads_p[st["b"]] = torch.empty_like(params[st["b"]])  # conv bias in front of a BatchNorm: exactly 0 (zeroed below)
                        zero_bias.append(grads_p[st["b"]])
                # the convolution's input as ONE tensor (the decoder's upsample + concat is materialised for the weight gradient)
                cin_p = c1 + c2
                if st["ups"]:
                    a_in = L.i16(npix * cin_p * mult)
                    _native.check(L.lib.nastar_upcat_f16(src.data_ptr(), src2.data_ptr() if src2 is not None else None, a_in.data_ptr(), B, h, w,
                                                         c1, c2, int(split), L.stream), "nastar_upcat_f16")
                else:
                    a_in = src
                grads_p[st["w"]] = L.wgrad(dzb, a_in, B, h, w, cur_co, cin_p, cout_r, cin, split, S)
                if st["src"] == "x0":
                    continue
                wpack, scale, shift, _ = ctx.tpack[id(st)] if ctx.tpack is not None else L.pack(wt, True, split, scal=sv["scal"])
                da = L.i16(npix * cin_p * mult)
                L.conv(dzb, wpack, scale, shift, B, h, w, cur_co, cin_p, sflag, out=da)
                if st["ups"]:
                    dx = L.i16(B * (h // 2) * (w // 2) * c1 * mult)
                    dsk = L.i16(npix * c2 * mult) if c2 else None
                    _native.check(L.lib.nastar_upcat_bwd_f16(da.data_ptr(), dx.data_ptr(), dsk.data_ptr() if dsk is not None else None, B, h, w,
                                                             c1, c2, int(split), L.stream), "nastar_upcat_bwd_f16")
                    accumulate(st["src"], dx, S, B * (h // 2) * (w // 2), c1)
                    if c2:
                        accumulate(st["skip"], dsk, S, npix, c2)
                else:
                    accumulate(st["src"], da, S, npix, cin_p)
            if zero_bias:
                torch._foreach_zero_(zero_bias)  # one launch for all of them
        return (None, None) + tuple(grads_p)


def unet_supported(unet: nn.Module, H: int, W: int) -> bool:
    from .planner.encoder import VggUnet
    model = getattr(unet, "model", None)
    if not isinstance(model, VggUnet):
        return False
    depth = model.depth
    if H % (1 << depth) or W % (1 << depth):
        return False
    # every convolution but the 1-channel head produces a multiple of 32 channels -- or 16 (encoder_depth = 5 ends in a 16-channel decoder
    # block: _UnetTrunk pads it to 32 with zero channels) -- decided HERE, before any BatchNorm running statistic has been touched
    convs = [m for m in model.modules() if isinstance(m, nn.Conv2d)]
    if not convs or any(c.out_channels % 32 and c.out_channels != 16 for c in convs if c.out_channels != 1) or convs[0].in_channels > 32:
        return False
    return all(chunk_rows(H >> l, W >> l) > 0 and (W >> l) >= 2 for l in range(depth + 1))


def unet_training_plan(model: nn.Module):
    """(plan, params): the launch plan of encoder_hip.unet_layer_plan as dict steps for ``_UnetTrunk`` -- kind, buffer names, resolution
    divisors, flags, the BatchNorm module (running statistics) and the SLOTS of the step's parameters in the flat ``params`` list
    (conv weight, conv bias or None, BatchNorm weight / bias or None for the head).  Pure host logic."""
    from .encoder_hip import CONV_FINAL as _F, CONV_UPSAMPLE as _U, unet_layer_plan
    params, plan = [], []

    def slot(t):
        if t is None:
            return None
        params.append(t)
        return len(params) - 1
    for st in unet_layer_plan(model):
        if st[0] == "pool":
            plan.append({"kind": "pool", "dst": st[1], "src": st[2], "div_in": st[4], "div": st[4] * 2})
            continue
        _, dst, src, skip, conv, bn, flags, div = st
        plan.append({"kind": "conv", "dst": dst, "src": src, "skip": skip, "div": div, "ups": bool(flags & _U), "final": bool(flags & _F),
                     "bn": bn, "w": slot(conv.weight), "b": slot(conv.bias), "g": slot(bn.weight if bn is not None else None),
                     "be": slot(bn.bias if bn is not None else None)})
    return plan, params


def unet_train_forward(unet: nn.Module, map_designs: torch.Tensor, start_maps: torch.Tensor, goal_maps: torch.Tensor, plus: bool,
                       precision: str = "f16x3") -> torch.Tensor:
    """``unet(cat(map, start + goal))`` for this package's ``VggUnet`` definition of the reference's Unet(vgg16_bn) (encoder.py:37-57)
    in TRAINING mode -- or in eval mode under autograd --, differentiable w.r.t. every parameter, on the MI355X kernels.  Returns the cost map
    [B,1,H,W] fp32."""
    B, _, H, W = map_designs.shape
    if not unet_supported(unet, H, W):
        raise NotImplementedError("VggUnet on maps whose size is a multiple of 2^depth (and whose widths suit the weight-gradient chunks)")
    plan, params = unet_training_plan(unet.model)
    if any(p.dtype != torch.float32 or not p.is_contiguous() for p in params):
        raise NotImplementedError("fp32 contiguous parameters expected")
    split = precision == "f16x3"
    # module.eval() with gradients on: BatchNorm on its running statistics (no update), still differentiable (round 6, as cnn_train_forward)
    cfg = {"split": split, "shape": (B, H, W), "plan": plan, "debug": getattr(unet, "_nastar_debug", None), "eval_bn": not unet.training}
    with torch.cuda.device(map_designs.device):
        x0 = _assemble_input(map_designs, start_maps, goal_maps, plus, split, _Lib(map_designs.device))
    z = _UnetTrunk.apply(cfg, x0, *params)
    return torch.sigmoid(z) * unet.const
