"""GPU parity of the encoder TRAINING kernels (csrc/nastar_conv_wgrad.hip.h, nastar_encoder_train.hip.h, neural_astar/encoder_train.py)
against torch autograd: the weight-gradient MFMA kernel, the per-channel statistics / affine streaming kernels, and the whole
CNN encoder (reference planner/encoder.py:60-78 in training mode, the shipped mazes_032 checkpoint's weights) forward + backward.

Truth = the torch module in float64 on the CPU.  Bar: f16x3 (split operands) within 1e-4 relative (max|diff| / max|ref| per tensor) on every
parameter gradient, 1e-5 absolute on the (0,1) cost map; plain fp16 operands 2e-2 relative."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import golden_util as G

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


WG_CASES = [
    # B, H, W, co, ci, split
    (8, 32, 32, 64, 32, True),
    (4, 32, 32, 128, 64, False),
    (16, 16, 16, 32, 32, True),
    (3, 64, 64, 256, 128, True),
    (32, 8, 8, 64, 64, False),
    (5, 32, 32, 32, 256, True),    # last block: co = 1 padded to 32, ci = 256
    (12, 4, 16, 96, 32, True),
    (3, 96, 96, 32, 32, True),     # WarCraft image resolution: one 96-pixel row per chunk (6 k-steps), the wide staging variant
    (5, 48, 48, 64, 32, False),    # two rows per chunk
    (6, 48, 48, 64, 32, True),
    (7, 24, 24, 128, 64, True),    # four rows per chunk
    (9, 12, 12, 32, 128, True),    # six rows = 72 pixels: a partial fifth k-step
    (4, 20, 45, 64, 64, True),     # 2 x 45 = 90 pixels
    (37, 4, 4, 64, 128, True),     # tiny images (U-Net 4x4 level): five per chunk, the last chunk partial
    (50, 2, 2, 128, 64, False),    # twelve 2x2 images per chunk
    (3, 8, 128, 64, 32, True),     # round 6: images wider than 96 pixels -- chunk rows are SEGMENTS of an image row (128 = 2 x 64)
    (2, 6, 200, 32, 64, False),    # ... 200 = 4 x 50: one 50-pixel segment per chunk, frame columns hold the neighbouring segments' pixels
    (2, 4, 160, 96, 32, True),     # ... 160 = 2 x 80 (the wide staging variant)
    (2, 5, 127, 64, 32, True),     # ... a prime width: two ragged segments (64 + 63 pixels; the last column of the second lies beyond the image)
    (1, 3, 197, 32, 64, False),    # ... 197 (prime) = 66 + 66 + 65
]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "B%d_%dx%d_co%d_ci%d_s%d" % tuple(int(v) for v in c))
def test_wgrad_matches_torch_autograd(case):
    from neural_astar import _native
    B, H, W, co, ci, split = case
    lib = _native.load()
    dev = _dev()
    g = torch.Generator().manual_seed(sum(int(v) for v in case))
    a = torch.randn((B, ci, H, W), generator=g)
    dz = torch.randn((B, co, H, W), generator=g)
    if co == 32 and ci == 256:
        dz[:, 1:] = 0
    da_, dz_ = _nhwc(a, split).to(dev), _nhwc(dz, split).to(dev)
    co_r, ci_r = (1, ci) if (co == 32 and ci == 256) else (co - 3, ci - 5)  # cropped outputs (padded channels are not written)
    dw = torch.full((co_r, ci_r, 3, 3), 7.0, dtype=torch.float32, device=dev)
    nbytes = int(lib.nastar_conv3x3_wgrad_workspace_bytes(B, H, W, co, ci))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    gscale = torch.full((1,), 4.0, device=dev)
    for _ in range(2):  # twice: deterministic (fixed-order partial sums), bitwise equal
        rc = lib.nastar_conv3x3_wgrad_f16(dz_.data_ptr(), da_.data_ptr(), dw.data_ptr(), B, H, W, co, ci, co_r, ci_r, int(split), 2.0,
                                          gscale.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc, "nastar_conv3x3_wgrad_f16")
        first = dw.clone() if _ == 0 else first
    assert torch.equal(first, dw)
    got = dw.cpu().double() * 2.0  # out_scale / gscale = 0.5
    w = torch.zeros((co, ci, 3, 3), dtype=torch.float64, requires_grad=True)
    nn.functional.conv2d(_seen(a, split).double(), w, None, padding=1).backward(_seen(dz, split).double())
    ref = w.grad[:co_r, :ci_r]
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= (2e-6 if split else 1e-5), err  # same operands, fp32 accumulation over B*H*W pixels vs float64


@pytest.mark.parametrize("split", [False, True])
def test_chan_stats_and_affine(split):
    from neural_astar import _native
    lib = _native.load()
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    B, H, W, C = 6, 16, 8, 64
    npix = B * H * W
    z = torch.randn((B, C, H, W), generator=g) * 2 + 0.3
    da = torch.randn((B, C, H, W), generator=g)
    ms = (torch.rand(C, generator=g) + 0.5).to(dev)
    mt = (torch.randn(C, generator=g) * 0.3).to(dev)
    k1, k2, k3 = ((torch.randn(C, generator=g)).to(dev) for _ in range(3))
    z_, da_ = _nhwc(z, split).to(dev), _nhwc(da, split).to(dev)
    zs, das = _seen(z, split).double(), _seen(da, split).double()
    st = torch.cuda.current_stream(dev).cuda_stream
    sums = torch.empty((C, 2), dtype=torch.float64, device=dev)
    _native.check(lib.nastar_chan_stats_f16(None, z_.data_ptr(), None, None, sums.data_ptr(), None, npix, C, int(split), st), "stats")
    assert torch.allclose(sums[:, 0].cpu(), zs.sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9)
    assert torch.allclose(sums[:, 1].cpu(), (zs * zs).sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9)
    mask = (ms.cpu().float().view(1, -1, 1, 1) * zs.float() + mt.cpu().float().view(1, -1, 1, 1)) > 0
    amax = torch.full((1,), -1.0, device=dev)
    _native.check(lib.nastar_chan_stats_f16(da_.data_ptr(), z_.data_ptr(), ms.data_ptr(), mt.data_ptr(), sums.data_ptr(), amax.data_ptr(),
                                            npix, C, int(split), st), "stats")
    dy = das * mask
    assert float(amax) == float(dy.abs().max().float())
    assert torch.allclose(sums[:, 0].cpu(), dy.sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9)
    assert torch.allclose(sums[:, 1].cpu(), (dy * zs).sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9)
    # two-stage form (what the training path launches): per-workgroup partials + a fixed-order finishing kernel -- same sums, no
    # zero-filled outputs needed, bitwise reproducible
    nb = int(lib.nastar_chan_stats_workspace_bytes(npix, C))
    assert nb > 0
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    runs = []
    for _ in range(2):
        s2 = torch.full((C, 2), 7.0, dtype=torch.float64, device=dev)
        a2 = torch.full((1,), -1.0, device=dev)
        _native.check(lib.nastar_chan_stats_f16_ws(da_.data_ptr(), z_.data_ptr(), ms.data_ptr(), mt.data_ptr(), s2.data_ptr(), a2.data_ptr(),
                                                   npix, C, int(split), ws.data_ptr(), nb, st), "stats_ws")
        runs.append((s2.clone(), a2.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][1]) == float(amax)
    assert torch.allclose(runs[0][0].cpu(), sums.cpu(), rtol=1e-13, atol=1e-10)
    s3 = torch.full((C, 2), 7.0, dtype=torch.float64, device=dev)
    _native.check(lib.nastar_chan_stats_f16_ws(None, z_.data_ptr(), None, None, s3.data_ptr(), None, npix, C, int(split), ws.data_ptr(), nb, st), "stats_ws")
    assert torch.allclose(s3[:, 0].cpu(), zs.sum(dim=(0, 2, 3)), rtol=1e-12, atol=1e-9)
    assert lib.nastar_chan_stats_f16_ws(None, z_.data_ptr(), None, None, s3.data_ptr(), None, npix, C, int(split), ws.data_ptr(), 8, st) == _native.NASTAR_ERR_WORKSPACE
    out = torch.empty_like(z_)
    v = lambda t: t.cpu().float().view(1, -1, 1, 1)  # noqa: E731
    for u, relu in ((None, True), (da_, False)):
        _native.check(lib.nastar_chan_affine_f16(u.data_ptr() if u is not None else None, z_.data_ptr(), k1.data_ptr(), k2.data_ptr(),
                                                 k3.data_ptr(), ms.data_ptr(), mt.data_ptr(), out.data_ptr(), npix, C, int(relu),
                                                 int(split), st), "affine")
        o = out.float().cpu()
        got = (o[..., :C] + (o[..., C:] if split else 0)).permute(0, 3, 1, 2)
        ref = v(k2) * zs.float() + v(k3)
        if u is not None:
            ref = ref + v(k1) * das.float() * mask
        if relu:
            ref = ref.clamp_min(0)
        assert float((got - ref).abs().max()) <= (2e-6 if split else 2e-3) * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("split", [False, True])
def test_device_weight_pack_matches_the_torch_pack(split):
    """nastar_pack_conv_weight_f16 (forward and input-gradient forms) against the torch-op pack of neural_astar/encoder_train.py"""
    from neural_astar import encoder_train as ET
    dev = _dev()
    L = ET._Lib(dev)
    g = torch.Generator().manual_seed(4)
    for (co, ci) in ((64, 32), (1, 256), (32, 2), (96, 40)):
        w = (torch.randn((co, ci, 3, 3), generator=g) * 0.03).to(dev)
        b = torch.randn(co, generator=g).to(dev)
        for tf in (False, True):
            wl = w.transpose(0, 1).flip(2, 3).contiguous() if tf else w
            ref_pack, ref_unscale = ET.pack_flat_weight(wl, split)
            wpack, scale, shift, scal = L.pack(w, tf, split, None if tf else b)
            if tf:  # the input-gradient pack re-using the forward pack's maximum is the same pack
                wp2 = L.pack(w, True, split, None, scal=L.pack(w, False, split, b)[3])[0]
                assert torch.equal(wp2, wpack)
            torch.cuda.synchronize()
            assert torch.equal(wpack.view(ref_pack.shape), ref_pack), (co, ci, tf)
            cout_l = ci if tf else co
            assert torch.equal(scale, ref_unscale.float().expand_as(scale))
            exp_shift = torch.zeros_like(shift)
            if not tf:
                exp_shift[:cout_l] = b
            assert torch.equal(shift, exp_shift)


@pytest.mark.parametrize("split", [False, True])
def test_maxpool2x2_backward_matches_torch(split):
    from neural_astar import _native
    lib = _native.load()
    dev = _dev()
    g = torch.Generator().manual_seed(8)
    B, H, W, C = 5, 12, 8, 40
    r = torch.relu(torch.randn((B, C, H, W), generator=g))   # post-ReLU activations (exact zeros tie: first maximum wins)
    dp = torch.randn((B, C, H // 2, W // 2), generator=g)
    r_, dp_ = _nhwc(r, split).to(dev), _nhwc(dp, split).to(dev)
    dr = torch.empty_like(r_)
    _native.check(lib.nastar_maxpool2x2_bwd_f16(r_.data_ptr(), dp_.data_ptr(), dr.data_ptr(), B, H, W, C, int(split),
                                                torch.cuda.current_stream(dev).cuda_stream), "pool bwd")
    o = dr.float().cpu()
    got = (o[..., :C] + (o[..., C:] if split else 0)).permute(0, 3, 1, 2)
    rr = _seen(r, split).requires_grad_(True)
    nn.functional.max_pool2d(rr, 2).backward(_seen(dp, split))
    assert torch.equal(got, rr.grad)


def _shipped_cnn_planner():
    """NeuralAstar(CNN) carrying the shipped mazes_032_moore_c8 checkpoint's weights (tests/golden/ckpt_mazes032_cnn.npz)"""
    from neural_astar.planner import NeuralAstar
    z = np.load(os.path.join(G.GOLDEN_DIR, "ckpt_mazes032_cnn.npz"))
    na = NeuralAstar(encoder_arch="CNN", Tmax=0.25)
    na.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    return na


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("precision,tol_grad,tol_cost", [("f16x3", 1e-4, 1e-5), ("f16", 1e-1, 5e-3)])
def test_cnn_encoder_training_step_matches_torch_autograd(precision, tol_grad, tol_cost):
    from neural_astar.utils import synthetic as syn
    dev = _dev()
    pr = syn.maze_maps(16, 32, seed=3)
    m, s, g = (torch.from_numpy(x) for x in pr)
    R = torch.randn((16, 1, 32, 32), generator=torch.Generator().manual_seed(1)) / (16 * 1024)  # an L1-mean-sized upstream gradient
    # truth: the torch module in float64 on the CPU, training mode
    ref = _shipped_cnn_planner().double().train()
    cost_ref = ref.encode(m.double(), s.double(), g.double())
    (cost_ref * R.double()).sum().backward()
    na = _shipped_cnn_planner().to(dev).train()
    na.encoder_backend = "hip_" + precision
    cost = na.encode(m.to(dev), s.to(dev), g.to(dev))
    assert cost.grad_fn is not None
    (cost * R.to(dev)).sum().backward()
    assert float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()) <= tol_cost
    worst = {}
    for (name, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()):
        assert p.grad is not None, name
        if name.endswith("bias") and p.dim() == 1 and name.split(".")[1] in ("0", "3", "6", "9", "12"):
            # conv biases in front of a BatchNorm: the true gradient is 0; float64 autograd returns rounding noise
            assert float(p.grad.abs().max()) == 0.0 and float(q.grad.abs().max()) <= 1e-12 * max(1.0, float(R.abs().max()) * 1e6)
            continue
        worst[name] = _rel(p.grad, q.grad)
    print("GRADERR", precision, {k: f"{v:.1e}" for k, v in worst.items()},
          "cost", float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()))
    assert max(worst.values()) <= tol_grad, worst
    # nn.BatchNorm2d's training-mode side effect: running statistics and the batch counter
    for (name, b), (_, c) in zip(na.encoder.named_buffers(), ref.encoder.named_buffers()):
        if b.dtype.is_floating_point:
            assert _rel(b, c) <= (1e-5 if precision == "f16x3" else 5e-3), name
        else:
            assert int(b) == int(c), name


@pytest.mark.parametrize("arch,depth,H,W,B", [("CNN", 4, 32, 32, 16), ("CNN", 2, 24, 40, 5), ("CNNDownSize", 2, 32, 64, 5), ("CNN", 4, 16, 64, 5)])
def test_eval_mode_with_gradients_runs_on_the_kernels_and_matches_torch_autograd(arch, depth, H, W, B):
    """module.eval() with gradients on (saliency maps, fine-tuning with frozen statistics): BatchNorm normalises with its RUNNING statistics and
    updates nothing, yet every parameter gets its gradient.  Round 6: on the MI355X kernels (the batch-statistics closed form in the limit of
    infinitely many pixels); against the torch module in float64, eval mode.  1e-4 per tensor when every ReLU decision of the reference is clear
    of the arithmetic's resolution; one mask bit decided by less than that moves a whole gradient ELEMENT (dbeta of a 256-channel block is a
    sum of ~sqrt(N) |dy|: one flipped pixel is ~1 / sqrt(B H W) of it), so 3e-2 is asserted then -- as in the training-mode tests above."""
    from neural_astar.planner import NeuralAstar
    dev = _dev()
    torch.manual_seed(depth * 7 + H)
    g = torch.Generator().manual_seed(H + W)
    ref = NeuralAstar(encoder_input="m+", encoder_arch=arch, encoder_depth=depth, const=3.0)
    with torch.no_grad():
        for m_ in ref.encoder.modules():
            if isinstance(m_, nn.BatchNorm2d):
                m_.weight.uniform_(0.5, 1.5); m_.bias.normal_(0, 0.2)
                m_.running_mean.normal_(0, 0.3); m_.running_var.uniform_(0.5, 2.0)
    na = copy.deepcopy(ref).to(dev).eval()
    ref = ref.double().eval()
    img = (torch.rand((B, 1, H, W), generator=g) > 0.25).float()
    s = torch.zeros((B, 1, H, W)); s[:, 0, 1, 1] = 1
    gl = torch.zeros((B, 1, H, W)); gl[:, 0, H - 2, W - 2] = 1
    ho, wo = (H >> depth, W >> depth) if arch == "CNNDownSize" else (H, W)
    R = torch.randn((B, 1, ho, wo), generator=g) / (B * ho * wo)
    before = {k: v.clone() for k, v in na.encoder.named_buffers()}
    margins, hooks = _decision_margins(ref.encoder)
    cost_ref = ref.encode(img.double(), s.double(), gl.double())
    for hk in hooks:
        hk.remove()
    (cost_ref * R.double()).sum().backward()
    cost = na.encode(img.to(dev), s.to(dev), gl.to(dev))
    assert na.last_encoder_route.endswith("-evalgrad/f16x3"), na.last_encoder_route
    (cost * R.to(dev)).sum().backward()
    assert float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()) <= 1e-5 * 3.0
    worst = {}
    for (name, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()):
        assert p.grad is not None, name
        if float(q.grad.abs().max()) == 0:
            continue
        worst[name] = _rel(p.grad, q.grad)
    clear = margins["relu"] >= 3e-6 and margins["pool"] >= 3e-6  # (eval-mode pre-activations are O(1..10): the resolution of split-fp16 products scales with them)
    print("EVALGRAD", arch, depth, H, W, margins, " ".join(f"{k}={v:.1e}" for k, v in worst.items()))
    assert len(worst) >= 3 * depth + 2 and max(worst.values()) <= (1e-4 if clear else 3e-2), (margins, worst)  # (eval mode: the conv biases in front of a BatchNorm DO get gradients)
    lower = [v for k, v in worst.items() if k.startswith("model.") and k.split(".")[1] in ("0", "1")]  # the first block sees a flipped element diluted through every block above it
    assert max(lower) <= 5e-3, worst
    for k, v in na.encoder.named_buffers():  # nothing was updated
        assert torch.equal(v, before[k]), k


def _decision_margins(encoder):
    """Forward hooks on a float64 reference encoder: the smallest relative gap between the two largest values of a max-pool window
    (both positive) and the smallest |ReLU input|.  ReLU masks and pooling arg-maxes are DISCRETE decisions: where the float64
    reference decides by less than the fp32-grade arithmetic's resolution (~2e-7), another correct implementation may decide the other
    way, which moves one whole gradient element (not a rounding error)."""
    res = {"pool": 1.0, "relu": 1.0}
    hooks = []

    def pool_hook(mod, inp, out):
        x = inp[0]
        B, C, H, W = x.shape
        win = x.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
        top = win.topk(2, dim=-1).values
        both = top[..., 1] > 0
        if bool(both.any()):
            res["pool"] = min(res["pool"], float(((top[..., 0] - top[..., 1]) / top[..., 0].clamp_min(1e-30))[both].min()))

    def relu_hook(mod, inp, out):
        res["relu"] = min(res["relu"], float(inp[0].abs().min()))

    for m in encoder.model:
        if isinstance(m, nn.MaxPool2d):
            hooks.append(m.register_forward_hook(pool_hook))
        if isinstance(m, nn.ReLU):
            hooks.append(m.register_forward_hook(relu_hook))
    return res, hooks


@pytest.mark.parametrize("arch,enc_in,depth,C,H,W,hw,const,seed", [
    ("CNNDownSize", "rgb+", 3, 3, 96, 96, 12, 10.0, 1000),   # WarCraft (train_warcraft.yaml); seed with clear pooling / ReLU decisions
    ("CNNDownSize", "rgb+", 3, 3, 96, 96, 12, 10.0, 0),      # ... and one where the float64 reference decides two pool windows by 3e-7
    ("CNNDownSize", "m+", 2, 1, 32, 64, None, None, 0),
    ("CNN", "m+", 3, 1, 20, 45, None, 2.0, 0),
    ("CNN", "m", 2, 1, 24, 24, None, None, 0),
    ("CNN", "m+", 4, 1, 64, 128, None, None, 0),             # round 6: the reference's own rectangle scenario (tests/astar_test.py:45-53), wider than
    ("CNN", "m+", 2, 1, 12, 200, None, 5.0, 1),
    ("CNN", "m+", 2, 1, 10, 127, None, 5.0, 1)])             # the flat tiles (2-D tiles in the convolutions, row segments in the weight gradients)
def test_other_encoder_stacks_train_on_the_hip_kernels(arch, enc_in, depth, C, H, W, hw, const, seed):
    """CNNDownSize (max-pool after every hidden block; the WarCraft configuration 96x96 RGB -> 12x12) and CNNs of other depths / map
    sizes: forward + every parameter gradient against the torch module in float64.  f16x3: 1e-4 relative per tensor whenever every
    discrete decision of the reference (pooling arg-max, ReLU mask) is clear of the arithmetic's resolution; if the reference itself
    decides by < 6e-7, single gradient elements may legitimately land elsewhere and only 3e-2 is asserted."""
    from neural_astar.planner import NeuralAstar
    dev = _dev()
    torch.manual_seed(depth * 10 + H + seed)
    B = 6
    g = torch.Generator().manual_seed(H + W + seed)
    img = torch.rand((B, C, H, W), generator=g) if C == 3 else (torch.rand((B, 1, H, W), generator=g) > 0.25).float()
    h, w = (hw, hw) if hw else ((H >> depth, W >> depth) if arch == "CNNDownSize" and enc_in == "rgb+" else (H, W))
    s = torch.zeros((B, 1, h, w)); s[:, 0, 1, 1] = 1
    gl = torch.zeros((B, 1, h, w)); gl[:, 0, h - 2, w - 2] = 1
    ref = NeuralAstar(encoder_input=enc_in, encoder_arch=arch, encoder_depth=depth, const=const)
    with torch.no_grad():
        for m_ in ref.encoder.modules():
            if isinstance(m_, nn.BatchNorm2d):
                m_.weight.uniform_(0.5, 1.5); m_.bias.normal_(0, 0.2)
    na = copy.deepcopy(ref).to(dev).train()
    ref = ref.double().train()
    ho, wo = (H >> depth, W >> depth) if arch == "CNNDownSize" else (H, W)
    R = torch.randn((B, 1, ho, wo), generator=g) / (B * ho * wo)
    margins, hooks = _decision_margins(ref.encoder)
    seen_in = {}
    for i, mod in enumerate(ref.encoder.model):  # what every ReLU / max-pool of the float64 reference saw, element by element
        if isinstance(mod, (nn.ReLU, nn.MaxPool2d)):
            hooks.append(mod.register_forward_hook(lambda m_, inp, out, i=i: seen_in.__setitem__(i, inp[0].detach().clone())))
    cost_ref = ref.encode(img.double(), s.double(), gl.double())
    for hk in hooks:
        hk.remove()
    (cost_ref * R.double()).sum().backward()
    na.encoder_backend = "hip_f16x3"
    dbg = {}
    na.encoder._nastar_debug = dbg
    cost = na.encode(img.to(dev), s.to(dev), gl.to(dev))
    assert cost.grad_fn is not None
    (cost * R.to(dev)).sum().backward()
    assert cost.shape == cost_ref.shape
    scale = float(const) if const else 1.0
    assert float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()) <= 1e-5 * scale
    worst = {}
    for (name, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()):
        assert p.grad is not None, name
        if name.endswith("bias") and "model." in name and float(p.grad.abs().max()) == 0:
            continue  # conv biases in front of a BatchNorm: exact zeros here, rounding noise in float64 autograd
        worst[name] = _rel(p.grad, q.grad)
    clear = margins["pool"] >= 6e-7 and margins["relu"] >= 6e-7
    print("GRADERR", arch, depth, H, W, seed, margins, " ".join(f"{k}={v:.1e}" for k, v in worst.items()))
    assert len(worst) >= 2 * depth + 2 and max(worst.values()) <= (1e-4 if clear else 3e-2), (margins, worst)
    if seed == 1000:
        assert clear, margins  # this seed was chosen for its clear decisions: the strict bound must have been the one applied
    # The float64 REFERENCE judges the discrete decisions, always: every ReLU mask bit / pooling arg-max that the
    # reference takes clearly (|input| >= 1e-4; runner-up >= 1e-4 relative below the winner) must be the HIP path's too, element by
    # element; only where the reference itself is ambiguous may the HIP path differ, and only those elements are then set to the HIP
    # path's choice in the float64 module that bounds the gradients at 1e-4 -- a mask bug on a clear decision cannot be copied.
    forced = copy.deepcopy(ref)
    for p_ in forced.parameters():
        p_.grad = None
    mods = list(forced.encoder.model)
    blk = n_amb = n_flip = 0
    for i, mod in enumerate(mods):
        if isinstance(mod, nn.ReLU):
            z, k2, k3, r, shape = dbg[f"fwd:{blk}"]
            hip = (k2.cpu().float().view(1, -1, 1, 1) * _unsplit(z, shape).float() + k3.cpu().float().view(1, -1, 1, 1)) > 0
            x_ref = seen_in[i]
            mine = x_ref > 0
            ok = x_ref.abs() >= TAU_RELU
            assert bool((hip == mine)[ok].all()), (blk, int((hip != mine)[ok].sum()), "clear ReLU decisions differ from float64")
            forced.encoder.model[i] = _ForcedReLU(torch.where(ok, mine, hip).double())
            n_amb += int((~ok).sum()); n_flip += int((hip != mine).sum())
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.MaxPool2d):
                hipa = _window(_unsplit(r, shape)).argmax(dim=-1)
                win = _window(seen_in[i + 1])
                minea = win.argmax(dim=-1)
                top = win.topk(2, dim=-1).values
                okp = ((top[..., 0] - top[..., 1]) >= TAU_POOL * top[..., 0].abs().clamp_min(1e-30)) | (top[..., 0] <= 0)
                assert bool((hipa == minea)[okp].all()), (blk, int((hipa != minea)[okp].sum()), "clear pooling decisions differ from float64")
                forced.encoder.model[i + 1] = _ForcedPool(nn.functional.one_hot(torch.where(okp, minea, hipa), 4).double())
                n_amb += int((~okp).sum()); n_flip += int((hipa != minea).sum())
            blk += 1
    (forced.encode(img.double(), s.double(), gl.double()) * R.double()).sum().backward()
    worst2 = {name: _rel(p.grad, q.grad) for (name, p), (_, q) in zip(na.encoder.named_parameters(), forced.encoder.named_parameters())
              if not (name.endswith("bias") and "model." in name and float(p.grad.abs().max()) == 0)}
    print("GRADERR ambiguous-only forcing:", f"{n_amb} ambiguous decisions by the reference's margins, {n_flip} decided the other way;",
          " ".join(f"{k}={v:.1e}" for k, v in worst2.items()))
    assert max(worst2.values()) <= 1e-4, worst2
    for (name, b), (_, c) in zip(na.encoder.named_buffers(), ref.encoder.named_buffers()):
        if b.dtype.is_floating_point:
            assert _rel(b, c) <= 1e-5, name


@pytest.mark.parametrize("split", [False, True])
def test_unet_decoder_plumbing_kernels(split):
    """nastar_upcat_f16 / nastar_upcat_bwd_f16 (nearest x2 upsampling + skip concatenation and its backward) and nastar_grad_add_f16
    (two gradients with different power-of-two scales) against torch"""
    from neural_astar import _native
    lib = _native.load()
    dev = _dev()
    st = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator().manual_seed(21)
    B, H, W, C1, C2 = 3, 8, 12, 64, 32
    x = torch.randn((B, C1, H // 2, W // 2), generator=g)
    sk = torch.randn((B, C2, H, W), generator=g)
    x_, sk_ = _nhwc(x, split).to(dev), _nhwc(sk, split).to(dev)
    M = 2 if split else 1
    cat = torch.empty((B, H, W, (C1 + C2) * M), dtype=torch.float16, device=dev)
    _native.check(lib.nastar_upcat_f16(x_.data_ptr(), sk_.data_ptr(), cat.data_ptr(), B, H, W, C1, C2, int(split), st), "upcat")
    ref = torch.cat((nn.functional.interpolate(_seen(x, split), scale_factor=2, mode="nearest"), _seen(sk, split)), dim=1)
    o = cat.float().cpu()
    C = C1 + C2
    got = (o[..., :C] + (o[..., C:] if split else 0)).permute(0, 3, 1, 2)
    assert torch.equal(got, ref)
    # backward
    d = torch.randn((B, C, H, W), generator=g)
    d_ = _nhwc(d, split).to(dev)
    dx = torch.empty((B, H // 2, W // 2, C1 * M), dtype=torch.float16, device=dev)
    dsk = torch.empty((B, H, W, C2 * M), dtype=torch.float16, device=dev)
    _native.check(lib.nastar_upcat_bwd_f16(d_.data_ptr(), dx.data_ptr(), dsk.data_ptr(), B, H, W, C1, C2, int(split), st), "upcat bwd")
    ds = _seen(d, split)
    ox, os_ = dx.float().cpu(), dsk.float().cpu()
    gx = (ox[..., :C1] + (ox[..., C1:] if split else 0)).permute(0, 3, 1, 2)
    gs = (os_[..., :C2] + (os_[..., C2:] if split else 0)).permute(0, 3, 1, 2)
    assert torch.equal(gs, ds[:, C1:])
    refx = nn.functional.avg_pool2d(ds[:, :C1].double(), 2) * 4
    assert float((gx.double() - refx).abs().max()) <= (2e-6 if split else 4e-3) * float(refx.abs().max())
    # scaled accumulate
    a = torch.randn((B, C2, H, W), generator=g) * 300
    b = torch.randn((B, C2, H, W), generator=g) * 20
    a_, b_ = _nhwc(a, split).to(dev), _nhwc(b, split).to(dev)
    Sa, Sb, So = torch.tensor([1024.0], device=dev), torch.tensor([64.0], device=dev), torch.zeros(1, device=dev)
    out = torch.empty_like(a_)
    _native.check(lib.nastar_grad_add_f16(a_.data_ptr(), Sa.data_ptr(), b_.data_ptr(), Sb.data_ptr(), out.data_ptr(), So.data_ptr(), B * H * W, C2,
                                          int(split), st), "grad add")
    assert float(So) == 64.0
    oo = out.float().cpu()
    got = (oo[..., :C2] + (oo[..., C2:] if split else 0)).permute(0, 3, 1, 2)
    ref = _seen(a, split).double() / 16 + _seen(b, split).double()
    assert float((got.double() - ref).abs().max()) <= (2e-6 if split else 2e-3) * float(ref.abs().max())


class _ForcedReLU(nn.Module):
    """ReLU with a prescribed mask: the float64 reference takes the HIP path's discrete decisions, so that the comparison sees only
    arithmetic (a ReLU mask or pooling arg-max decided the other way moves a whole gradient element: not a rounding error)."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask


class _ForcedPool(nn.MaxPool2d):  # still an nn.MaxPool2d: VggUnet splits its stages at the pools
    def __init__(self, onehot):
        super().__init__(kernel_size=2, stride=2)
        self.onehot = onehot  # [B,C,H/2,W/2,4] one-hot of the window element the HIP path took

    def forward(self, x):
        B, C, H, W = x.shape
        win = x.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
        return (win * self.onehot).sum(-1)


def _unsplit(buf, shape):
    B, h, w, C = shape
    o = buf.view(torch.float16).float().reshape(B, h, w, 2 * C).cpu().double()
    return (o[..., :C] + o[..., C:]).permute(0, 3, 1, 2)


TAU_RELU = 1e-4  # |BatchNorm output| below which the float64 reference's own ReLU decision is "ambiguous": 10x what fp32-grade arithmetic
TAU_POOL = 1e-4  # leaves on a normalised pre-activation (~1e-5); same, relative, for the two largest values of a pooling window


def _window(x):
    B, C, H, W = x.shape
    return x.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)


@pytest.mark.parametrize("seed,depth,size", [(4, 4, 32), (11, 4, 32), (23, 4, 32), (5, 4, 64), (6, 5, 64)])
def test_unet_trains_on_the_hip_kernels(seed, depth, size):
    """Unet(vgg16_bn) (this package's VggUnet definition; reference encoder.py:37-57) in training mode: 23 conv + batch-statistics BatchNorm +
    ReLU blocks, four max-pools, four upsample + concat decoder blocks and the head -- forward and every parameter gradient on the MI355X
    kernels against the torch module in float64, over three seeds.

    A 26-layer network takes ~2.5 million discrete decisions per batch here (ReLU masks, pooling arg-maxes); fp32-grade arithmetic leaves
    ~1e-5 on the normalised pre-activations, so a handful of them legitimately fall the other way than in float64, and each moves a whole
    gradient element (1/55 of a BatchNorm-bias gradient summed over 3072 pixels).  The judge is the float64 REFERENCE, not the code
    under test:
    (1) every decision the reference itself takes CLEARLY (|BatchNorm output| >= 1e-4; pooling runner-up >= 1e-4 relative below the
        winner) must be the HIP path's decision too -- compared element by element, no tolerance: a mask or arg-max bug cannot hide;
    (2) against the plain float64 module: cost map within 2e-5, every gradient within 3e-2 (the handful of ambiguous decisions), and
        the fraction of gradient tensors already within 2e-4 is printed ("unforced");
    (3) against the float64 module whose AMBIGUOUS decisions only (by the reference's own margins) are set to the HIP path's: every
        gradient within 2e-4.  Clear decisions stay the reference's own, so (3) cannot copy a bug into the judge."""
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import synthetic as syn
    from neural_astar import encoder_hip as E
    import test_unet_gpu as TU
    dev = _dev()
    B = 3
    pr = syn.random_obstacle_maps(B, size, size, 0.25, seed=seed)
    m, s, g = (torch.from_numpy(x) for x in pr)
    base = NeuralAstar(encoder_arch="Unet", encoder_depth=depth)  # (depth 5, round 6: its 16-channel last decoder block padded to 32 channels)
    base.encoder = TU._calibrated_unet(depth=depth, seed=3)
    for mod in base.encoder.modules():
        if isinstance(mod, nn.ReLU):
            mod.inplace = False
    na = copy.deepcopy(base).to(dev).train()
    R = torch.randn((B, 1, size, size), generator=torch.Generator().manual_seed(9 + seed)) / (B * size * size)
    dbg = {}
    na.encoder._nastar_debug = dbg
    na.encoder_backend = "hip_f16x3"
    cost = na.encode(m.to(dev), s.to(dev), g.to(dev))
    assert cost.grad_fn is not None
    (cost * R.to(dev)).sum().backward()
    torch.cuda.synchronize()

    def compare(ref):
        cost_ref = ref.encode(m.double(), s.double(), g.double())
        (cost_ref * R.double()).sum().backward()
        err_cost = float((cost.detach().cpu().double() - cost_ref.detach()).abs().max())
        worst = {}
        for (name, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()):
            assert p.grad is not None, name
            if name.endswith("bias") and float(p.grad.abs().max()) == 0 and float(q.grad.abs().max()) <= 1e-9 * float(R.abs().max()):
                continue  # conv biases in front of a BatchNorm
            worst[name] = _rel(p.grad, q.grad)
        return err_cost, worst

    def decision_sites(model):
        """[(container, index, kind, dbg key)] of every ReLU that follows a planned convolution and of every max-pool, in plan order"""
        plan = E.unet_layer_plan(model)
        conv_name = {id(st[4]): st[1] for st in plan if st[0] == "conv"}
        sites = []
        for seq in [mod for mod in model.modules() if isinstance(mod, nn.Sequential)]:
            ch = list(seq)
            for i, c in enumerate(ch):
                if isinstance(c, nn.Conv2d) and id(c) in conv_name and "fwd:" + conv_name[id(c)] in dbg:
                    for j in range(i + 1, min(i + 3, len(ch))):
                        if isinstance(ch[j], nn.ReLU):
                            sites.append((seq, j, "relu", "fwd:" + conv_name[id(c)]))
        pools = [st for st in plan if st[0] == "pool"]
        feats, pi = model.encoder.features, 0
        for i, mod in enumerate(feats):
            if isinstance(mod, nn.MaxPool2d) and pi < len(pools):
                sites.append((feats, i, "pool", "fwd:" + pools[pi][1]))
                pi += 1
        return sites

    # (2) the plain float64 module, with hooks that keep what every decision saw
    ref = copy.deepcopy(base).double().train()
    seen_in = {}
    hooks = []
    ref_sites = decision_sites(ref.encoder.model)
    n_relu = {4: 23, 5: 25}[depth]  # VGG16 stages of 2, 2, 3, 3, 3 convolutions (depth 4 stops in front of the fifth pool), center 2, decoder 2 per block
    assert sum(k == "relu" for _, _, k, _ in ref_sites) == n_relu and sum(k == "pool" for _, _, k, _ in ref_sites) == depth
    for n, (cont, idx, kind, key) in enumerate(ref_sites):
        hooks.append(cont[idx].register_forward_hook(lambda mod, inp, out, n=n: seen_in.__setitem__(n, inp[0].detach().clone())))
    err_cost, worst = compare(ref)
    for h in hooks:
        h.remove()
    unforced_ok = sum(v <= 2e-4 for v in worst.values())
    print("GRADERR unet plain seed", seed, "cost", err_cost, "n", len(worst), "max", max(worst.values()),
          f"unforced: {unforced_ok}/{len(worst)} gradient tensors within 2e-4")
    # (the plain comparison: every flipped decision moves one gradient element; 3e-2 holds on 32x32 maps, the larger batches of decisions at 64x64
    #  -- 4x the pixels -- are judged by (1) and (3) below alone)
    assert err_cost <= 2e-5 and len(worst) >= 24 + 2 * 23 and (size > 32 or max(worst.values()) <= 3e-2), sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    for (name, b), (_, c) in zip(na.encoder.named_buffers(), ref.encoder.named_buffers()):
        if b.dtype.is_floating_point:
            assert _rel(b, c) <= 2e-5, name
        else:
            assert int(b) == int(c), name
    # (1) + (3): decisions judged by the reference's own margins
    forced = copy.deepcopy(base).double().train()
    f_sites = decision_sites(forced.encoder.model)
    n_dec = n_amb = n_flip = 0
    for n, ((cont, idx, kind, key), (fcont, fidx, _, _)) in enumerate(zip(ref_sites, f_sites)):
        x_ref = seen_in[n]
        if kind == "relu":
            z, k2, k3, shape = dbg[key]
            hip = (k2.cpu().float().view(1, -1, 1, 1) * _unsplit(z, shape).float() + k3.cpu().float().view(1, -1, 1, 1)) > 0  # the kernels' fp32 test
            assert not bool(hip[:, x_ref.shape[1]:].any())  # (a 16-channel block padded to 32: the padding channels stay off)
            hip = hip[:, :x_ref.shape[1]]
            mine = x_ref > 0
            clear = x_ref.abs() >= TAU_RELU
            assert bool((hip == mine)[clear].all()), (key, int((hip != mine)[clear].sum()), "clear ReLU decisions differ from float64")
            fcont[fidx] = _ForcedReLU(torch.where(clear, mine, hip).double())
        else:
            src, shape = dbg[key]
            hip = _window(_unsplit(src, shape)).argmax(dim=-1)  # argmax: first maximum, the kernel's rule
            win = _window(x_ref)
            mine = win.argmax(dim=-1)
            top = win.topk(2, dim=-1).values
            clear = (top[..., 0] - top[..., 1]) >= TAU_POOL * top[..., 0].abs().clamp_min(1e-30)
            clear = clear | (top[..., 0] <= 0)  # an all-zero window (after ReLU): no gradient whichever element is taken
            assert bool((hip == mine)[clear].all()), (key, int((hip != mine)[clear].sum()), "clear pooling decisions differ from float64")
            fcont[fidx] = _ForcedPool(nn.functional.one_hot(torch.where(clear, mine, hip), 4).double())
        n_dec += clear.numel()
        n_amb += int((~clear).sum())
        n_flip += int((hip != mine).sum())
    err_cost2, worst2 = compare(forced)
    w3 = sorted(worst2.items(), key=lambda kv: -kv[1])[:4]
    print("GRADERR unet seed", seed, f"decisions {n_dec}, ambiguous by the reference's own margins {n_amb}, decided the other way {n_flip};",
          "ambiguous-only forcing: cost", err_cost2, "worst", " ".join(f"{k}={v:.1e}" for k, v in w3))
    assert n_amb <= 2e-3 * n_dec, (n_amb, n_dec)
    assert max(worst2.values()) <= 2e-4, w3


@pytest.mark.parametrize("n_maps,hw,const", [(3, 12, 10.0), (100, 32, None), (700, 32, 2.5)])
def test_last_block_batchnorm_sigmoid_matches_torch_autograd(n_maps, hw, const):
    """The closing 1-channel BatchNorm (batch statistics) + sigmoid * const block as two launches each way (nastar_bn1_*) against torch's
    float64 autograd of encoder.py's last block + :32-34: cost, dz, dgamma, dbeta, dconst and the running-statistics update; one, a few
    and the maximum number of partial rows; const as a parameter and as the plain 1.0."""
    from neural_astar import encoder_train as ET
    dev = _dev()
    g = torch.Generator().manual_seed(n_maps)
    z = (torch.randn((n_maps, 1, hw, hw), generator=g) * 1.3 + 0.2)
    up = torch.randn((n_maps, 1, hw, hw), generator=g) / z.numel()
    bn = nn.BatchNorm2d(1)
    with torch.no_grad():
        bn.weight.fill_(1.7); bn.bias.fill_(-0.3); bn.running_mean.fill_(0.4); bn.running_var.fill_(2.0)
    cpar = nn.Parameter(torch.ones(1) * const) if const is not None else None
    # truth: float64 torch modules
    import copy
    bn64 = copy.deepcopy(bn).double().train()
    z64 = z.double().requires_grad_(True)
    c64 = cpar.detach().double().requires_grad_(True) if cpar is not None else 1.0
    cost64 = torch.sigmoid(bn64(z64)) * c64
    (cost64 * up.double()).sum().backward()
    # device
    bnd = copy.deepcopy(bn).to(dev).train()
    zd = z.to(dev).requires_grad_(True)
    cd = nn.Parameter(cpar.detach().to(dev)) if cpar is not None else ET._const_tensor(1.0, dev)
    cost = ET._LastBlock.apply(zd, bnd.weight, bnd.bias, cd.reshape(1), bnd.eps, bnd.momentum, bnd.running_mean, bnd.running_var)
    (cost * up.to(dev)).sum().backward()
    assert float((cost.detach().cpu().double() - cost64.detach()).abs().max()) <= 2e-6 * (const or 1.0)
    assert _rel(zd.grad, z64.grad) <= 2e-5
    assert _rel(bnd.weight.grad, bn64.weight.grad) <= 2e-5 and _rel(bnd.bias.grad, bn64.bias.grad) <= 2e-5
    if cpar is not None:
        assert _rel(cd.grad, c64.grad) <= 2e-5
    assert _rel(bnd.running_mean, bn64.running_mean) <= 1e-6 and _rel(bnd.running_var, bn64.running_var) <= 1e-6


@pytest.mark.parametrize("split", [False, True])
def test_all_weight_packs_of_a_step_in_one_launch(split):
    """nastar_pack_conv_weights_multi_f16 == nastar_pack_conv_weight_f16 per weight (forward and input-gradient forms, shared maxima)"""
    from neural_astar import encoder_train as ET
    dev = _dev()
    L = ET._Lib(dev)
    g = torch.Generator().manual_seed(9)
    shapes = [(32, 2), (64, 32), (128, 64), (1, 128), (96, 40)]
    ws = [(torch.randn((co, ci, 3, 3), generator=g) * (0.02 + 0.3 * k)).to(dev) for k, (co, ci) in enumerate(shapes)]
    bs = [torch.randn(co, generator=g).to(dev) for co, _ in shapes]
    wmax = L.weight_maxima(ws) if split else None
    specs = [(ws[k], bs[k], False, k) for k in range(len(ws))] + [(ws[k], None, True, k) for k in range(1, len(ws))]
    for _ in range(2):  # the second call takes the cached table
        packs = L.pack_all(specs, split, wmax)
    assert packs is not None and len(packs) == len(specs)
    for (w, b, tf, k), (wpack, scale, shift, scal) in zip(specs, packs):
        rp, rs, rsh, rscal = L.pack(w, tf, split, b)
        torch.cuda.synchronize()
        assert torch.equal(wpack, rp) and torch.equal(scale, rs) and torch.equal(shift, rsh), (tuple(w.shape), tf)
        if split:
            assert torch.equal(scal, rscal)
    # a weight that is not plain fp32 declines (the caller packs one by one)
    assert L.pack_all([(ws[0].double(), None, False, 0)], split, None) is None


def test_fused_rmsprop_equals_torch_rmsprop():
    """utils/optim.FusedRMSprop (one launch per step) against torch.optim.RMSprop on the same gradients: parameters and state"""
    from neural_astar.utils.optim import FusedRMSprop
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(32, 2, 3, 3), (32,), (256, 128, 3, 3), (1,), (7, 5)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = FusedRMSprop(pa, 1e-3), torch.optim.RMSprop(pb, 1e-3)
    for step in range(5):
        for x, y in zip(pa, pb):
            gr = (torch.randn(x.shape, generator=g) * 10.0 ** (step - 3)).to(dev)
            x.grad, y.grad = gr.clone(), gr.clone()
        if step == 3:  # a parameter without a gradient is skipped by both
            pa[1].grad = pb[1].grad = None
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for x, y in zip(pa, pb):
        assert float((x - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))
        sa, sb = oa.state[x], ob.state[y]
        assert float(sa["step"]) == float(sb["step"])
        assert torch.allclose(sa["square_avg"], sb["square_avg"], rtol=2e-6, atol=1e-30)
    # checkpoints are interchangeable
    ob2 = torch.optim.RMSprop(pb, 1e-3)
    ob2.load_state_dict(oa.state_dict())
    assert torch.equal(ob2.state[pb[0]]["square_avg"], oa.state[pa[0]]["square_avg"])


def test_fused_rmsprop_closure_and_two_parameter_groups():
    """step(closure) with zero_grad + backward INSIDE the closure (Lightning's automatic optimisation) and two parameter groups with
    their own learning rates: the one-launch step uses this call's gradients and one cached device table per group."""
    from neural_astar.utils.optim import FusedRMSprop
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    shapes = [(64, 32, 3, 3), (64,), (9, 5)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:2], "lr": 1e-3}, {"params": ps[2:], "lr": 5e-3}]  # noqa: E731
    oa, ob = FusedRMSprop(groups(pa), 1e-3), torch.optim.RMSprop(groups(pb), 1e-3)
    tgt = [torch.randn(s, generator=g).to(dev) for s in shapes]

    def closure_for(params, opt):
        def closure():
            opt.zero_grad(set_to_none=True)
            loss = sum(((p - t) ** 2).sum() for p, t in zip(params, tgt))
            loss.backward()
            return loss
        return closure
    for _ in range(4):
        la, lb = oa.step(closure_for(pa, oa)), ob.step(closure_for(pb, ob))
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    torch.cuda.synchronize()
    assert len(oa._tables) == 2  # one cached table per group, neither rebuilt by the other
    for x, y in zip(pa, pb):
        assert float((x - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("B,H,W,C", [(1, 4, 4, 32), (6, 16, 8, 64), (300, 32, 32, 256)])
def test_two_launch_batchnorm_pass_equals_the_three_launch_form(split, B, H, W, C):
    """nastar_bn_stats_coef_{fwd,bwd}_f16 (partial rows, then ONE kernel that finishes the sums in the same fixed order and forms the
    coefficients) against nastar_chan_stats_f16_ws + nastar_bn_coef_{fwd,bwd}: identical outputs."""
    from neural_astar import _native, encoder_train as ET
    dev = _dev()
    L = ET._Lib(dev)
    lib = L.lib
    g = torch.Generator().manual_seed(31 + C)
    npix = B * H * W
    z = torch.randn((B, C, H, W), generator=g) * 1.5 + 0.2
    da = torch.randn((B, C, H, W), generator=g) * 3.0
    z_, da_ = _nhwc(z, split).to(dev), _nhwc(da, split).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.2).to(dev)
    eps, mom = 1e-5, 0.1
    sums = L.stats(None, z_, None, None, npix, C, split)
    k2r, k3r = L.f32(C), L.f32(C)
    meanr = torch.empty((C,), dtype=torch.float64, device=dev)
    invr = torch.empty((C,), dtype=torch.float64, device=dev)
    rm_r, rv_r = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    _native.check(lib.nastar_bn_coef_fwd(sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, npix, mom, rm_r.data_ptr(), rv_r.data_ptr(),
                                         k2r.data_ptr(), k3r.data_ptr(), meanr.data_ptr(), invr.data_ptr(), C, L.stream), "coef_fwd")
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd, k2, k3 = L.bn_fwd(z_, npix, C, split, gamma, beta, eps, mom, rm, rv)
    torch.cuda.synchronize()
    for a, b in ((mean, meanr), (invstd, invr), (k2, k2r), (k3, k3r), (rm, rm_r), (rv, rv_r)):
        assert torch.equal(a, b)
    amax = L.f32(1)
    sums_b = L.stats(da_, z_, k2r, k3r, npix, C, split, amax=amax)
    S_r = torch.full((1,), 4.0, device=dev)
    dgr, dbr, c1r, c2r, c3r = (L.f32(C) for _ in range(5))
    _native.check(lib.nastar_bn_coef_bwd(sums_b.data_ptr(), amax.data_ptr(), meanr.data_ptr(), invr.data_ptr(), gamma.data_ptr(), npix,
                                         S_r.data_ptr(), dgr.data_ptr(), dbr.data_ptr(), c1r.data_ptr(), c2r.data_ptr(), c3r.data_ptr(), C,
                                         L.stream), "coef_bwd")
    S_in, S_out = torch.full((1,), 4.0, device=dev), torch.zeros(1, device=dev)
    s_out = torch.empty((C, 2), dtype=torch.float64, device=dev)
    dg, db, c1, c2, c3 = L.bn_bwd(da_, z_, k2r, k3r, npix, C, split, meanr, invr, gamma, S_in, S_out, sums_out=s_out)
    torch.cuda.synchronize()
    assert float(S_in) == 4.0 and torch.equal(S_out, S_r) and torch.equal(s_out, sums_b)
    for a, b in ((dg, dgr), (db, dbr), (c1, c1r), (c2, c2r), (c3, c3r)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_closing_convolution_as_streams_equals_the_padded_matrix_form(precision):
    """csrc/nastar_encoder_co1.hip.h: the 1-channel closing convolution of the CNN encoder as streams -- forward projection + shifted
    sum (with the block in front's BatchNorm + ReLU applied while loading: its activation tensor is never written), streamed weight
    gradient, input gradient formed on the fly inside the BatchNorm-backward passes -- against the same step with that layer padded to
    32 channels on the MFMA (encoder_train.CO1_STREAMS = False): cost map and every gradient within the precision's own noise (the
    third printed column pair is each form's distance to float64 autograd), f16x3 cost map within 1e-5 of float64."""
    from neural_astar import encoder_train as ET
    from neural_astar.utils import synthetic as syn
    dev = _dev()
    pr = syn.maze_maps(24, 32, seed=5)
    m, s, g = (torch.from_numpy(x) for x in pr)
    R = torch.randn((24, 1, 32, 32), generator=torch.Generator().manual_seed(2)) / (24 * 1024)
    ref = _shipped_cnn_planner().double().train()
    cost_ref = ref.encode(m.double(), s.double(), g.double())
    (cost_ref * R.double()).sum().backward()
    out = {}
    keep = ET.CO1_STREAMS
    try:
        for mode in (True, False):
            ET.CO1_STREAMS = mode
            na = _shipped_cnn_planner().to(dev).train()
            na.encoder_backend = "hip_" + precision
            cost = na.encode(m.to(dev), s.to(dev), g.to(dev))
            (cost * R.to(dev)).sum().backward()
            out[mode] = (cost.detach(), {n: p.grad.clone() for n, p in na.encoder.named_parameters()})
    finally:
        ET.CO1_STREAMS = keep
    tol_c, tol_g = (2e-6, 5e-5) if precision == "f16x3" else (5e-3, 2e-1)
    assert float((out[True][0] - out[False][0]).abs().max()) <= tol_c
    worst = {}
    for (name, q) in ref.encoder.named_parameters():
        a, b = out[True][1][name], out[False][1][name]
        if float(q.grad.abs().max()) <= 1e-12:
            assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
            continue
        worst[name] = (_rel(a, b), _rel(a, q.grad), _rel(b, q.grad))
    print("CO1", precision, {k: tuple(f"{x:.1e}" for x in v) for k, v in worst.items()})
    assert max(v[0] for v in worst.values()) <= tol_g, worst
    if precision == "f16x3":  # (gradients against float64 are judged on the clear-margin batch of the test above: this batch has ReLU decisions inside fp32 noise)
        assert float((out[True][0].cpu().double() - cost_ref.detach()).abs().max()) <= 1e-5


@pytest.mark.parametrize("arch,depth,H,W", [("CNN", 4, 32, 32), ("CNN", 2, 20, 45), ("CNNDownSize", 2, 64, 64), ("Unet", 3, 32, 32)])
def test_training_mode_without_autograd_runs_on_the_kernels(arch, depth, H, W):
    """module.train() under torch.no_grad() -- a validation pass nobody switched to eval() (the reference's Lightning loop does switch): batch
    statistics, running statistics and num_batches_tracked updated, nothing recorded.  Round 6: the training kernels' forward (it used to fall
    back to torch.nn with a warning); cost map within 1e-5 of the same module in float64 on torch.nn, buffers within 1e-6."""
    import copy
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import synthetic as syn
    dev = torch.device("cuda:0")
    pr = syn.random_obstacle_maps(16, H, W, 0.2, seed=3)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    if arch == "CNNDownSize":
        s = torch.zeros(16, 1, H // 4, W // 4, device=dev)
        g = torch.zeros_like(s)
        s[:, 0, 1, 1] = 1
        g[:, 0, -2, -2] = 1
    torch.manual_seed(0)
    a = NeuralAstar(encoder_arch=arch, encoder_depth=depth, learn_obstacles=(arch == "CNNDownSize")).to(dev).train()
    b = copy.deepcopy(a).double()  # the reference in float64: torch's own kernels, no MIOpen solver choice in the comparison
    b.encoder_backend = "torch"
    with torch.no_grad():
        ca, cb = a.encode(m, s, g), b.encode(m.double(), s.double(), g.double())
    assert a.last_encoder_route.startswith(f"hip:{arch}-train/"), a.last_encoder_route
    assert float((ca.double() - cb).abs().max()) <= 1e-5
    for (n, x), (_, y) in zip(a.encoder.named_buffers(), b.encoder.named_buffers()):
        if "num_batches" in n:
            assert int(x) == int(y), n  # (1 for every layer the stack visits; a depth-3 U-Net leaves the deeper VGG layers at 0)
        elif "running" in n:
            assert float((x.double() - y).abs().max()) <= 1e-6, n


@pytest.mark.parametrize("seed,depth,size,training", [(4, 4, 32, False), (11, 3, 32, False)])
def test_unet_eval_mode_with_gradients_and_depth_5_run_on_the_kernels(seed, depth, size, training):
    """Two U-Net corners that fell back to torch.nn (with a warning) until late in round 6:
    * eval mode with gradients on (a planner somebody calls without torch.no_grad(), saliency maps, fine-tuning with frozen statistics): the training
      kernels with BatchNorm on its RUNNING statistics (coefficients formed on the host side, the backward through the unfused path with
      `npix -> 2^62`, conv biases in front of a BatchNorm get gamma invstd sum dy);
    * encoder_depth = 5, whose last decoder block has 16 channels: padded to 32 with zero channels (gamma = beta = 0 there), in eval-with-grad AND
      in training mode (64x64 maps: the deepest level is 2x2).
    Against the same module in float64: cost map within 2e-5, every parameter gradient within 3e-2 (a 26-layer network's handful of ReLU / pooling
    decisions that fp32-grade arithmetic takes the other way each move one gradient element: test_unet_trains_on_the_hip_kernels arbitrates those),
    at least half of the tensors within 2e-4; buffers untouched (eval) / equal to the reference's (training)."""
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import synthetic as syn
    import test_unet_gpu as TU
    dev = _dev()
    B = 3 if size < 64 else 8  # (a flipped decision moves ONE element of a gradient summed over B H W pixels)
    pr = syn.random_obstacle_maps(B, size, size, 0.25, seed=seed)
    m, s, g = (torch.from_numpy(x) for x in pr)
    base = NeuralAstar(encoder_arch="Unet", encoder_depth=depth)
    base.encoder = TU._calibrated_unet(depth=depth, seed=3)
    with torch.no_grad():
        for mod in base.encoder.modules():
            if isinstance(mod, nn.ReLU):
                mod.inplace = False
            if isinstance(mod, nn.BatchNorm2d):  # running statistics that are not the batch's
                mod.running_mean.mul_(0.9).add_(0.05)
                mod.running_var.mul_(1.2)
                mod.momentum = 0.1
    na = copy.deepcopy(base).to(dev).train(training)
    ref = copy.deepcopy(base).double().train(training)
    R = torch.randn((B, 1, size, size), generator=torch.Generator().manual_seed(9 + seed)) / (B * size * size)
    if size >= 64:
        # ~30 M ReLU / pooling decisions per batch here: a few dozen fall within fp32-grade rounding of zero and go the other way than in float64.
        # With an upstream gradient of random SIGN a per-channel sum like dbeta cancels down to ~sqrt(N) |dy|, and ONE flipped pixel is 1/sqrt(N) of it
        # (5e-3 .. 2e-2: measured, whatever the depth); with a one-signed upstream it is 1/N -- the comparison then sees the arithmetic, not the flips
        R = (R.abs() + 0.5 / (B * size * size))
    before = {k: v.clone() for k, v in na.encoder.named_buffers()}
    na.encoder_backend = "hip_f16x3"
    cost = na.encode(m.to(dev), s.to(dev), g.to(dev))
    assert na.last_encoder_route == ("hip:Unet-train/f16x3" if training else "hip:Unet-evalgrad/f16x3"), na.last_encoder_route
    (cost * R.to(dev)).sum().backward()
    cost_ref = ref.encode(m.double(), s.double(), g.double())
    (cost_ref * R.double()).sum().backward()
    assert float((cost.detach().cpu().double() - cost_ref.detach()).abs().max()) <= 2e-5
    worst = {}
    for (name, p), (_, q) in zip(na.encoder.named_parameters(), ref.encoder.named_parameters()):
        if q.grad is None:
            continue  # (VGG layers a shallower U-Net does not visit)
        assert p.grad is not None, name
        if float(q.grad.abs().max()) <= (1e-9 * float(R.abs().max()) if training else 0.0):
            continue  # (training mode: the conv biases in front of a BatchNorm have no gradient)
        worst[name] = _rel(p.grad, q.grad)
    tight = sum(v <= 2e-4 for v in worst.values())
    print("UNET", "TRAIN" if training else "EVALGRAD", "seed", seed, "depth", depth, "tensors", len(worst), "within 2e-4:", tight, "max", max(worst.values()))
    if os.environ.get("NASTAR_TEST_VERBOSE"):
        print("   ", " ".join(f"{k.replace('model.', '')}={v:.0e}" for k, v in worst.items()))
    assert len(worst) >= (40 if depth >= 4 else 30) and max(worst.values()) <= 3e-2, sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    assert training or tight >= len(worst) // 2  # (training mode: batch statistics couple every pixel to every flipped decision of its channel)
    if training:
        for (k, v), (_, c) in zip(na.encoder.named_buffers(), ref.encoder.named_buffers()):
            if v.dtype.is_floating_point:
                assert _rel(v, c) <= 2e-5, k
            else:
                assert int(v) == int(c), k
    else:
        assert any(k.endswith("bias") and "features" in k for k in worst)  # eval mode: the conv biases in front of a BatchNorm have gradients
        for k, v in na.encoder.named_buffers():
            assert torch.equal(v, before[k]), k
