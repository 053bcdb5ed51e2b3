"""GPU parity of the generic fp16 / f16x3 MFMA convolution (csrc/nastar_conv_flat.hip.h) and of the U-Net encoder built from it
(reference planner/encoder.py:37-57; BASELINE config 3), through the C ABI (nastar_conv3x3_f16 / nastar_maxpool2x2_f16 /
nastar_encoder_prep_f16), against a torch fp32 / fp64 reference of the same op.

Tolerances: plain fp16 operands -- the kernel must agree with a float64 conv2d of the SAME fp16-rounded operands to output rounding
(2^-11 relative); f16x3 (split operands) -- 1e-5 relative against the float64 conv of the unrounded fp32 operands.  Whole 26-layer
U-Net: the truth is the torch module evaluated in float64 on the CPU (the fp32 torch module itself is only ~1e-5 away from it after
26 layers, MIOpen's fp32 algorithms more); f16x3 must be within 1e-5 absolute of that truth on the (0,1) cost map (north-star float
tolerance) and no further from the fp32 torch module than 4e-5; plain fp16 within 3e-2."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    return torch.device("cuda:0")


def _nhwc(x, split):
    x = x.permute(0, 2, 3, 1).contiguous()
    hi = x.to(torch.float16)
    if split:
        return torch.cat((hi, (x - hi.float()).to(torch.float16)), dim=-1).contiguous()
    return hi.contiguous()


def _seen(t, split):
    hi = t.to(torch.float16).float()
    return hi + (t - hi).to(torch.float16).float() if split else hi


def _run_conv(xa, xb, conv, bn, B, H, W, flags, mul=1.0):
    from neural_astar import _native, encoder_hip as E
    lib = _native.load()
    dev = _dev()
    split = bool(flags & E.CONV_SPLIT)
    wpack, scale, shift, cin_p, cout_p = E.pack_flat_conv(conv, bn, split)
    ia = _nhwc(xa, split).to(dev)
    ib = _nhwc(xb, split).to(dev) if xb is not None else None
    wpack, scale, shift = wpack.to(dev), scale.to(dev), shift.to(dev)  # named: a temporary would be freed before the launch reads it
    c1, c2 = xa.shape[1], (xb.shape[1] if xb is not None else 0)
    assert c1 + c2 == cin_p
    final = bool(flags & E.CONV_FINAL)
    out = torch.full((B, H, W, cout_p * (2 if split else 1)), 7.0, dtype=torch.float16, device=dev)
    out32 = torch.full((B, H, W), -1.0, dtype=torch.float32, device=dev)
    rc = lib.nastar_conv3x3_f16(ia.data_ptr(), ib.data_ptr() if ib is not None else None, wpack.data_ptr(),
                                scale.data_ptr(), shift.data_ptr(), None if final else out.data_ptr(),
                                out32.data_ptr() if final else None, B, H, W, c1, c2, cout_p, flags, mul,
                                torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "nastar_conv3x3_f16")
    torch.cuda.synchronize()
    if final:
        return out32.cpu()
    o = out.float().cpu()
    v = o[..., :cout_p] + (o[..., cout_p:] if split else 0)
    return v.permute(0, 3, 1, 2)


CASES = [
    # B, H, W, c1, c2, cout(real), relu, ups, split
    (3, 32, 32, 32, 0, 64, True, False, False),
    (3, 32, 32, 64, 0, 64, True, False, True),
    (70, 2, 2, 64, 0, 64, True, False, False),      # 280 flat pixels: partial second tile, 2x2 images
    (33, 4, 4, 128, 0, 96, False, False, False),    # cout 96 -> 32-channel workgroups
    (5, 8, 8, 64, 64, 128, True, True, False),      # decoder block: upsample + concat
    (5, 8, 8, 32, 64, 64, True, True, True),
    (3, 16, 12, 64, 0, 32, True, True, False),      # upsample only (last decoder block), non-square
    (2, 6, 126, 32, 0, 32, False, False, False),    # widest supported row
    (6, 48, 48, 64, 0, 32, False, False, True),     # 32-channel workgroups, split operands, two slices per segment
    (6, 96, 96, 32, 0, 32, True, False, True),
    (1, 40, 56, 96, 0, 64, True, False, True),
    (40, 2, 2, 512, 512, 256, True, True, False),   # deepest decoder block: 1024 input channels
    (2, 6, 127, 32, 0, 32, False, False, False),    # round 6: rows wider than the flat tiles' halo -- 2-D tiles (64 x 4 pixels of one image);
    (3, 64, 128, 32, 0, 64, True, False, True),     # the reference's rectangle scenario (tests/astar_test.py:45-53)
    (2, 10, 200, 64, 0, 32, True, False, True),     # partial tiles both ways (200 = 3 x 64 + 8, 10 = 2 x 4 + 2)
    (2, 8, 256, 32, 64, 64, True, True, False),     # upsample + concat at 256 pixels per row
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dx%d_c%d+%d_o%d_r%d_u%d_s%d" % tuple(int(v) for v in c))
def test_flat_conv_matches_torch(case):
    from neural_astar import encoder_hip as E
    B, H, W, c1, c2, cout, relu, ups, split = case
    g = torch.Generator().manual_seed(1 + sum(int(v) for v in case))
    xa = torch.randn((B, c1, H // 2 if ups else H, W // 2 if ups else W), generator=g)
    xb = torch.randn((B, c2, H, W), generator=g) if c2 else None
    conv = nn.Conv2d(c1 + c2, cout, 3, padding=1)
    bn = nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * (c1 + c2))) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    flags = (E.CONV_RELU if relu else 0) | (E.CONV_UPSAMPLE if ups else 0) | (E.CONV_SPLIT if split else 0)
    got = _run_conv(xa, xb, conv, bn, B, H, W, flags).double()
    xin = _seen(xa, split)
    if ups:
        xin = nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    if c2:
        xin = torch.cat((xin, _seen(xb, split)), dim=1)
    # split form: the kernel's weight operands are 22-bit accurate (scaled into the normal fp16 range): compare with the exact weights
    wref = conv.weight.detach() if split else _seen(conv.weight.detach(), False)
    y = nn.functional.conv2d(xin.double(), wref.double(), None, padding=1)
    cout_p = E._pad32(cout)
    scale, shift = E.fold_bn(conv, bn, cout_p)  # the module's own eval-mode BatchNorm (the split pack folds a 2^-s into the kernel's)
    y = y * scale[:cout].double().view(1, -1, 1, 1) + shift[:cout].double().view(1, -1, 1, 1)
    if relu:
        y = y.clamp_min(0)
    tol = 1e-5 if split else 1.5e-3
    err = (got[:, :cout] - y).abs().max().item() / max(1.0, y.abs().max().item())
    assert err <= tol, err
    if cout_p > cout:
        assert float(got[:, cout:].abs().max()) == 0.0  # padded output channels stay zero


def test_flat_conv_final_layer_sigmoid():
    from neural_astar import encoder_hip as E
    g = torch.Generator().manual_seed(5)
    B, H, W, c = 9, 16, 16, 32
    x = torch.randn((B, c, H, W), generator=g)
    conv = nn.Conv2d(c, 1, 3, padding=1)
    for split in (False, True):
        got = _run_conv(x, None, conv, None, B, H, W, E.CONV_FINAL | (E.CONV_SPLIT if split else 0), mul=3.0)
        with torch.no_grad():
            ref = 3.0 * torch.sigmoid(conv(x))[:, 0]
        assert float((got - ref).abs().max()) <= (1e-5 if split else 5e-3)


@pytest.mark.parametrize("split", [False, True])
def test_maxpool2x2_f16(split):
    from neural_astar import _native
    lib = _native.load()
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, H, W, C = 7, 6, 10, 40
    x = torch.randn((B, C, H, W), generator=g)
    xin = _nhwc(x, split).to(dev)
    out = torch.empty((B, H // 2, W // 2, C * (2 if split else 1)), dtype=torch.float16, device=dev)
    _native.check(lib.nastar_maxpool2x2_f16(xin.data_ptr(), out.data_ptr(), B, H, W, C, int(split),
                                            torch.cuda.current_stream(dev).cuda_stream), "nastar_maxpool2x2_f16")
    o = out.float().cpu()
    got = (o[..., :C] + (o[..., C:] if split else 0)).permute(0, 3, 1, 2)
    ref = nn.functional.max_pool2d(_seen(x, split), 2)
    assert torch.equal(got, ref)


def _calibrated_unet(depth=4, seed=0):
    """Random-init Unet whose BatchNorm running statistics are the batch statistics of a random input (activations stay O(1)
    through all 26 layers, so every layer's arithmetic matters in the output)."""
    from neural_astar.planner.encoder import Unet
    torch.manual_seed(seed)
    u = Unet(2, depth, None)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in u.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.momentum = 1.0
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        u.train()
        x = torch.cat(((torch.rand((16, 1, 32, 32), generator=g) > 0.3).float(), torch.zeros(16, 1, 32, 32)), dim=1)
        x[:, 1, 3, 4] = 1
        x[:, 1, 20, 9] = 1
        u(x)
    return u.eval()


def _truth64(unet, m, s, g):
    """the torch module in float64 on the CPU"""
    import copy
    u64 = copy.deepcopy(unet).cpu().double().eval()
    with torch.no_grad():
        return u64(torch.cat((m.cpu().double(), (s + g).cpu().double()), dim=1)).float()


@pytest.mark.parametrize("precision,tol_truth,tol_torch", [("f16x3", 1.5e-5, 4e-5), ("f16", 3e-2, 3e-2)])
def test_unet_encoder_matches_torch(precision, tol_truth, tol_torch):
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import synthetic as syn
    dev = _dev()
    planner = NeuralAstar(encoder_arch="Unet", encoder_depth=4)
    planner.encoder = _calibrated_unet()
    planner = planner.to(dev).eval()
    pr = syn.random_obstacle_maps(24, 32, 32, 0.25, seed=2)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    truth = _truth64(planner.encoder, m, s, g)
    with torch.no_grad():
        ref = planner.encode(m, s, g)  # fp32 torch module on the device (MIOpen)
        planner.encoder_backend = "hip_" + precision
        got = planner.encode(m, s, g)
        assert type(planner._hip_encoder).__name__ == "HipUnetEncoder"
        assert float(truth.std()) > 1e-3, "degenerate cost map: the test would not see the layers"
        err_truth = float((got.cpu() - truth).abs().max())
        err_torch = float((got - ref).abs().max())
        print(f"unet {precision}: |hip - float64 truth| = {err_truth:.2e}, |hip - torch fp32 (device)| = {err_torch:.2e}, "
              f"|torch fp32 (device) - truth| = {float((ref.cpu() - truth).abs().max()):.2e}")
        assert err_truth <= tol_truth, err_truth
        assert err_torch <= tol_torch, err_torch
        if precision == "f16x3":
            # the yardstick for a 26-layer stack is the float64 truth: torch's OWN fp32 module is ~7e-6 from it, so two fp32-grade
            # implementations with different summation orders sit ~1e-5 from EACH OTHER however good they are; the split-operand
            # path (22-bit operands) must stay within 2x of the fp32 module's own distance to the truth
            assert err_truth <= 2.0 * float((ref.cpu() - truth).abs().max()) + 1e-6
        # the search consumes it (non-square maps as well: 16 x 48)
        out = planner(m, s, g)
        assert out.histories.shape == (24, 1, 32, 32) and float(out.paths.sum()) > 0
        m2 = torch.ones((3, 1, 16, 48), device=dev)
        s2 = torch.zeros_like(m2)
        g2 = torch.zeros_like(m2)
        s2[:, 0, 1, 1] = 1
        g2[:, 0, 14, 40] = 1
        t2 = _truth64(planner.encoder, m2, s2, g2)
        assert float((planner.encode(m2, s2, g2).cpu() - t2).abs().max()) <= tol_truth


@pytest.mark.parametrize("depth,H,W,enc_in,precision,tol", [(3, 20, 45, "m+", "f16x3", 1e-5), (2, 24, 40, "m", "f16x3", 1e-5),
                                                             (4, 20, 45, "m+", "f16", 2e-2), (4, 12, 12, "m+", "bf16", 2e-2),
                                                             (4, 24, 200, "m+", "f16x3", 1e-5), (3, 150, 130, "m", "f16x3", 1e-5)])
def test_cnn_of_any_depth_and_size_takes_the_generic_kernel(depth, H, W, enc_in, precision, tol):
    """CNN encoders the fixed-shape kernels do not cover (depth != 4, H / W not multiples of 32 or 16) run on the generic fp16 MFMA
    convolution instead of falling back to torch.nn (reference encoder.py:60-78 at any encoder_depth)."""
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import synthetic as syn
    dev = _dev()
    torch.manual_seed(depth * 100 + H)
    na = NeuralAstar(encoder_input=enc_in, encoder_arch="CNN", encoder_depth=depth, const=2.5).to(dev)
    with torch.no_grad():
        for mod in na.encoder.model:
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.2)
    na.eval()
    pr = syn.random_obstacle_maps(9, H, W, 0.2, seed=H + W)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        ref = na.encode(m, s, g)
        na.encoder_backend = "hip_" + precision
        got = na.encode(m, s, g)
        assert type(na._hip_encoder).__name__ == "HipFlatCnnEncoder"
        assert float((got - ref).abs().max()) <= tol * 2.5
        out = na(m, s, g)
        assert out.paths.shape == (9, 1, H, W) and int((na.astar.last_status != 0).sum()) == 0


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128), (128, 256), (256, 128), (128, 64)])
@pytest.mark.parametrize("split", [False, True])
def test_img32_layer_agrees_with_the_generic_convolution(cin, cout, split):
    """nastar_conv3x3_img32_f16 (the CNN encoder's persistent 32x32 kernel as a stand-alone layer, incl. the input-gradient shapes
    256 -> 128 and 128 -> 64 and the raw no-ReLU epilogue the training path uses) against nastar_conv3x3_f16 on the same operands, and
    against torch for one of them"""
    from neural_astar import _native, encoder_hip as E
    lib = _native.load()
    dev = _dev()
    g = torch.Generator().manual_seed(cin + cout + int(split))
    B = 41
    x = torch.randn((B, cin, 32, 32), generator=g)
    conv = nn.Conv2d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        conv.bias.copy_(torch.randn(cout, generator=g))
    wpack, scale, shift, _, _ = E.pack_flat_conv(conv, None, split)
    xin = _nhwc(x, split).to(dev)
    wpack, scale, shift = wpack.to(dev), scale.to(dev), shift.to(dev)
    M = 2 if split else 1
    st = torch.cuda.current_stream(dev).cuda_stream
    for relu in (False, True):
        flags = (E.CONV_RELU if relu else 0) | (E.CONV_SPLIT if split else 0)
        o1 = torch.full((B, 32, 32, cout * M), 3.0, dtype=torch.float16, device=dev)
        o2 = torch.full((B, 32, 32, cout * M), 5.0, dtype=torch.float16, device=dev)
        _native.check(lib.nastar_conv3x3_img32_f16(xin.data_ptr(), wpack.data_ptr(), scale.data_ptr(), shift.data_ptr(), o1.data_ptr(), B, cin,
                                                   cout, flags, st), "img32")
        _native.check(lib.nastar_conv3x3_f16(xin.data_ptr(), None, wpack.data_ptr(), scale.data_ptr(), shift.data_ptr(), o2.data_ptr(), None, B,
                                             32, 32, cin, 0, cout, flags, 1.0, st), "flat")
        torch.cuda.synchronize()
        a, b = o1.float(), o2.float()
        va = a[..., :cout] + (a[..., cout:] if split else 0)
        vb = b[..., :cout] + (b[..., cout:] if split else 0)
        tol = (2e-6 if split else 1e-3) * float(vb.abs().max())  # accumulation order only (plain: one fp16 ulp of the output)
        assert float((va - vb).abs().max()) <= tol, (relu, float((va - vb).abs().max()))
    ref = nn.functional.conv2d(_seen(x, split).double(), (conv.weight.detach() if split else _seen(conv.weight.detach(), False)).double(),
                               conv.bias.detach().double(), padding=1).clamp_min(0)
    got = va.permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max()) <= (1e-5 if split else 1.5e-3) * float(ref.abs().max())
