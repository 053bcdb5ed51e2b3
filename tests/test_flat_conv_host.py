"""CPU checks of the generic MFMA convolution's host side and addressing logic (csrc/nastar_conv_flat.hip.h):
the lane-level emulator (tests/flat_conv_emulator.py) must reproduce torch's conv2d for the kernel's weight pack, flat-halo staging,
upsample + concat gather, split-precision segments and final-layer epilogue; plus the U-Net launch plan."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.dirname(os.path.abspath(__file__))]

from neural_astar import encoder_hip as E  # noqa: E402
from neural_astar.planner.encoder import Unet, VggUnet  # noqa: E402

import flat_conv_emulator as emu  # noqa: E402


def nhwc_f16(x: torch.Tensor, split: bool) -> np.ndarray:
    """[B,C,H,W] fp32 -> flat fp16 NHWC, split: [hi(C) | lo(C)] per pixel"""
    x = x.permute(0, 2, 3, 1).contiguous()
    hi = x.to(torch.float16)
    if split:
        lo = (x - hi.float()).to(torch.float16)
        return torch.cat((hi, lo), dim=-1).reshape(-1).numpy()
    return hi.reshape(-1).numpy()


def from_nhwc(flat: np.ndarray, B, H, W, C, split: bool) -> torch.Tensor:
    t = torch.from_numpy(flat.astype(np.float32)).reshape(B, H, W, (2 if split else 1) * C)
    v = t[..., :C] + (t[..., C:] if split else 0)
    return v.permute(0, 3, 1, 2)


CASES = [
    # B, H, W, c1, c2, cout, relu, final, ups, split
    (3, 4, 4, 32, 0, 32, True, False, False, False),
    (2, 4, 6, 32, 32, 64, True, False, True, False),
    (2, 2, 2, 32, 0, 32, False, False, False, True),
    (2, 4, 4, 32, 0, 32, False, True, False, False),
    (5, 8, 8, 32, 0, 32, True, False, False, False),
    (1, 4, 4, 32, 32, 32, True, False, True, True),
    (1, 3, 5, 64, 0, 64, True, False, False, False),   # two slices, odd sizes
    (2, 2, 2, 64, 0, 128, False, False, False, True),  # 2x2 images, two 64-channel blocks, split: six virtual slices
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dx%d_c%d+%d_o%d_r%d_f%d_u%d_s%d" % tuple(int(v) for v in c))
def test_flat_conv_emulation_matches_torch_conv2d(case):
    B, H, W, c1, c2, cout, relu, final, ups, split = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    xa = torch.randn((B, c1, H // 2 if ups else H, W // 2 if ups else W), generator=g)
    xb = torch.randn((B, c2, H, W), generator=g) if c2 else None
    cout_real = 1 if final else cout
    conv = nn.Conv2d(c1 + c2, cout_real, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.1)
        conv.bias.copy_(torch.randn(conv.bias.shape, generator=g))
    wpack, scale, shift, cin_p, cout_p = E.pack_flat_conv(conv, None, split)
    assert (cin_p, cout_p) == (c1 + c2, cout)
    scale = scale * 1.5  # a non-trivial folded BatchNorm scale
    out = emu.run(nhwc_f16(xa, split), nhwc_f16(xb, split) if c2 else None, wpack.numpy().view(np.float16).reshape(-1), scale.numpy(),
                  shift.numpy(), B, H, W, c1, c2, cout, relu, final, ups, split, final_mul=2.0)
    # reference: conv2d of the operands the kernel sees (fp16-rounded, or the 22-bit hi+lo pairs), in float64
    def seen(t):
        hi = t.to(torch.float16).float()
        return (hi + (t - hi).to(torch.float16).float()) if split else hi
    xin = seen(xa)
    if ups:
        xin = nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    if c2:
        xin = torch.cat((xin, seen(xb)), dim=1)
    y = nn.functional.conv2d(xin.double(), seen(conv.weight.detach()).double(), None, padding=1)
    rscale, rshift = E.fold_bn(conv, None, cout)  # the module's own scale / shift (the split pack folds 2^-s into the kernel's)
    y = y * (1.5 * rscale[:cout_real]).double().view(1, -1, 1, 1) + rshift[:cout_real].double().view(1, -1, 1, 1)
    if relu:
        y = y.clamp_min(0)
    if final:
        ref = 2.0 * torch.sigmoid(y[:, 0])
        assert np.allclose(out.reshape(B, H, W), ref.numpy(), atol=1e-5)
        return
    got = from_nhwc(out, B, H, W, cout, split).double()
    tol = 2e-5 if split else 2e-3  # output rounding: 22 or 11 significant bits
    err = (got[:, :cout_real] - y).abs().max().item() / max(1.0, y.abs().max().item())
    assert err <= tol, err
    assert float(got[:, cout_real:].abs().max()) == 0.0 if cout_real < cout else True


def test_unet_launch_plan_and_packing():
    torch.manual_seed(0)
    u = Unet(2, 4, None).eval()
    assert isinstance(u.model, VggUnet)
    plan = E.unet_layer_plan(u.model)
    convs = [s for s in plan if s[0] == "conv"]
    pools = [s for s in plan if s[0] == "pool"]
    assert len(convs) == 13 + 2 + 8 + 1 and len(pools) == 4
    # decoder blocks: upsample (+ skip) -> conv; the last block has no skip; resolutions return to 1
    ups = [s for s in convs if s[6] & E.CONV_UPSAMPLE]
    assert [(s[4].in_channels, s[4].out_channels, s[3] is not None, s[7]) for s in ups] == \
        [(1024, 256, True, 8), (512, 128, True, 4), (256, 64, True, 2), (64, 32, False, 1)]
    assert convs[-1][6] == E.CONV_FINAL and convs[-1][4].out_channels == 1
    # every buffer is produced before it is consumed
    have = {"x0"}
    for s in plan:
        assert s[2] in have and (s[3] is None or s[3] in have)
        have.add(s[1])
    # split pack = [W_hi | W_hi | W_lo] over 3 * cin_p virtual channels; hi + lo reproduces the weight to 2^-21
    conv = convs[1][4]
    wp, sc, sh, cin_p, cout_p = E.pack_flat_conv(conv, convs[1][5], True)
    assert tuple(wp.shape) == (9, 3 * cin_p // 8, cout_p, 8)
    w = wp.view(torch.float16).float().permute(0, 1, 3, 2).reshape(9, 3 * cin_p, cout_p)
    hi, hi2, lo = w[:, :cin_p], w[:, cin_p:2 * cin_p], w[:, 2 * cin_p:]
    assert torch.equal(hi, hi2)
    ref = conv.weight.detach().permute(2, 3, 1, 0).reshape(9, conv.in_channels, conv.out_channels)
    # the pack holds w * 2^s (max|w| brought to ~2^14 so that the lo terms are normal fp16 numbers), 2^-s sits in `scale`
    k = float(hi.abs().max() / ref.abs().max())
    s_exp = round(np.log2(k))
    assert abs(k / 2.0 ** s_exp - 1) < 1e-3 and 8192 <= float(hi.abs().max()) <= 16384
    assert float(((hi + lo)[:, :conv.in_channels, :conv.out_channels] * 2.0 ** -s_exp - ref).abs().max()) <= 2.0 ** -22 * float(ref.abs().max())
    sc_plain = E.pack_flat_conv(conv, convs[1][5], False)[1]
    assert torch.allclose(sc * 2.0 ** s_exp, sc_plain, rtol=1e-6)


WG_CASES = [
    # B, H, W, co, ci, split
    (2, 8, 8, 32, 32, False),     # one 64-pixel chunk per image
    (2, 4, 16, 64, 32, True),     # split operands, two output blocks
    (1, 6, 12, 32, 64, True),     # 12-wide rows: 6 rows = 72 pixels, a partial fifth k-step
    (1, 4, 45, 32, 32, False),    # 2 x 45 = 90 pixels
    (7, 4, 4, 32, 32, True),      # tiny images, five per chunk (the last chunk holds two)
    (13, 2, 2, 32, 64, False),    # twelve 2x2 images per chunk + one
]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "B%d_%dx%d_co%d_ci%d_s%d" % tuple(int(v) for v in c))
def test_wgrad_emulation_matches_torch_autograd(case):
    """the weight-gradient kernel's addressing (chunks, framed LDS tiles, measured ds_read_b64_tr_b16 semantics, MFMA layout, partial
    sums) reproduced on the CPU must give conv2d's weight gradient"""
    import wgrad_emulator as wemu
    B, H, W, co, ci, split = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case))
    a = torch.randn((B, ci, H, W), generator=g)
    dz = torch.randn((B, co, H, W), generator=g)
    co_r, ci_r = co - 1, ci - 2
    got = wemu.run(nhwc_f16(dz, split), nhwc_f16(a, split), B, H, W, co, ci, co_r, ci_r, split, nsplit=3, out_scale=0.5)

    def seen(t):
        hi = t.to(torch.float16).float()
        return (hi + (t - hi).to(torch.float16).float()) if split else hi
    w = torch.zeros((co, ci, 3, 3), dtype=torch.float64, requires_grad=True)
    nn.functional.conv2d(seen(a).double(), w, None, padding=1).backward(seen(dz).double())
    ref = 0.5 * w.grad[:co_r, :ci_r]
    if split:  # the kernel drops the lo * lo product (2^-22 relative)
        assert float((torch.from_numpy(got).double() - ref).abs().max() / ref.abs().max()) <= 1e-6
    else:
        assert np.allclose(got, ref.numpy(), rtol=0, atol=1e-6 * float(ref.abs().max()))
    assert wemu.chunk_rows(H, W) == E_chunk_rows(H, W)


def E_chunk_rows(H, W):
    from neural_astar import encoder_train as ET
    return ET.chunk_rows(H, W)
