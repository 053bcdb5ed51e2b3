"""Dev probe (GPU box): generic fp16 MFMA convolution + U-Net encoder -- per-case error breakdown, layer timings, whole-encoder timing
against the fp32 torch module.  Usage: python tools/probe_unet.py [B]   (writes gpurun_out/probe_unet.json)"""
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]

import test_unet_gpu as T  # noqa: E402
from neural_astar import encoder_hip as E  # noqa: E402
from neural_astar.planner import NeuralAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402


def breakdown(case):
    B, H, W, c1, c2, cout, relu, ups, split = case
    g = torch.Generator().manual_seed(1 + sum(int(v) for v in case))
    xa = torch.randn((B, c1, H // 2 if ups else H, W // 2 if ups else W), generator=g)
    xb = torch.randn((B, c2, H, W), generator=g) if c2 else None
    conv = nn.Conv2d(c1 + c2, cout, 3, padding=1)
    flags = (E.CONV_RELU if relu else 0) | (E.CONV_UPSAMPLE if ups else 0) | (E.CONV_SPLIT if split else 0)
    got = T._run_conv(xa, xb, conv, None, B, H, W, flags).double()
    xin = T._seen(xa, split)
    if ups:
        xin = nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    if c2:
        xin = torch.cat((xin, T._seen(xb, split)), dim=1)
    with torch.no_grad():
        y = nn.functional.conv2d(xin.double(), T._seen(conv.weight.detach(), split).double(), conv.bias.double(), padding=1)
    if relu:
        y = y.clamp_min(0)
    e = (got[:, :cout] - y).abs()
    bad = e > (1e-4 if split else 1e-2) * max(1.0, float(y.abs().max()))
    info = {"case": list(map(int, case)), "max_err": float(e.max()), "bad_frac": float(bad.float().mean())}
    if bad.any():
        info["bad_by_channel_block32"] = [float(bad[:, k:k + 32].float().mean()) for k in range(0, cout, 32)]
        info["bad_by_row"] = [float(bad[:, :, r].float().mean()) for r in range(H)][:16]
        info["bad_by_col"] = [float(bad[:, :, :, c].float().mean()) for c in range(W)][:16]
        info["bad_by_image"] = [float(bad[b].float().mean()) for b in range(B)][:16]
        flat = bad.permute(0, 2, 3, 1).reshape(-1, cout).any(dim=1).numpy()
        info["bad_by_tile256"] = [float(flat[k:k + 256].mean()) for k in range(0, len(flat), 256)][:16]
    return info


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    res = {"cases": [breakdown(c) for c in T.CASES]}
    for c in res["cases"]:
        print(c)
    planner = NeuralAstar(encoder_arch="Unet", encoder_depth=4)
    planner.encoder = T._calibrated_unet()
    planner = planner.to(dev).eval()
    pr = syn.maze_maps(B, 32, seed=1234) if B <= 4096 else syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        ref = planner.encode(m[:256], s[:256], g[:256])
        truth = T._truth64(planner.encoder, m[:256], s[:256], g[:256])
        res["torch_fp32_device_max_abs_diff_vs_float64_truth"] = float((ref.cpu() - truth).abs().max())
        for prec in ("f16", "f16x3"):
            planner.encoder_backend = "hip_" + prec
            got = planner.encode(m[:256], s[:256], g[:256])
            res[f"unet_{prec}_max_abs_diff_vs_torch_fp32_256_maps"] = float((got - ref).abs().max())
            res[f"unet_{prec}_max_abs_diff_vs_float64_truth"] = float((got.cpu() - truth).abs().max())
            ms = timed(lambda: planner.encode(m, s, g))
            fl = planner._hip_encoder.flops(32, 32) * B
            res[f"unet_{prec}_ms_per_{B}_maps"] = ms
            res[f"unet_{prec}_useful_tflops"] = fl / ms / 1e9
            print(prec, res[f"unet_{prec}_max_abs_diff_vs_torch_fp32_256_maps"], ms, "ms", fl / ms / 1e9, "TFLOP/s useful")
        planner.encoder_backend = "torch"
        nb = min(B, 1024)
        ms = timed(lambda: planner.encode(m[:nb], s[:nb], g[:nb]), reps=2)
        res["unet_torch_fp32_ms_per_maps"] = [ms, nb]
        print("torch fp32", ms, "ms per", nb)
        # per-layer timing of the f16 path (events around each launch)
        planner.encoder_backend = "hip_f16"
        planner.encode(m, s, g)
        enc = planner._hip_encoder
        lib = __import__("neural_astar._native", fromlist=["x"]).load()
        orig = lib.nastar_conv3x3_f16
        times = []

        def wrapped(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = orig(*a)
            e1.record()
            times.append((e0, e1, a[7:13]))
            return rc
        lib.nastar_conv3x3_f16 = wrapped
        planner.encode(m, s, g)
        torch.cuda.synchronize()
        lib.nastar_conv3x3_f16 = orig
        lay = []
        for e0, e1, shp in times:
            b, h, w, c1, c2, co = shp
            ms = e0.elapsed_time(e1)
            lay.append({"B,H,W,c1,c2,cout": list(map(int, shp)), "ms": ms, "tflops_padded": 2.0 * 9 * (c1 + c2) * co * b * h * w / ms / 1e9})
            print(lay[-1])
        res["unet_f16_layers"] = lay
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "probe_unet.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
