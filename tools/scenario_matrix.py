#!/usr/bin/env python3
"""A sweep of USER scenarios through the planner API (run on the GPU box): VanillaAstar / NeuralAstar (CNN; at 64x64 also Unet of depth 4 and 5 with g_ratio 0.2 --
the exact batch pipeline -- and CNNDownSize) x map sizes from 32x32 to 1024x1024 incl. rectangles, a prime width and sizes either side of every
kernel boundary x {eval forward, store_intermediate_results, one training step with torch.nn.L1Loss + RMSprop}.  One JSON line per scenario:
wall-clock ms per call, the encoder route the call took (hip:* = the MFMA kernels; "torch.nn" would be a fall-back) and any warning --
or the error.  Found in round 6 with it: training on maps above 65,519 cells raised (the replay backward's 16-bit history stamps)."""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neural-astar_amd")]
import torch  # noqa: E402

from neural_astar.planner import NeuralAstar, VanillaAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
bad = 0


def run(name, fn, n=3, route=None):
    global bad
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
        d = {"case": name, "ms": round(ms, 3), "warnings": sorted({str(x.message)[:120] for x in w})[:3]}
        if route is not None:
            d["encoder_route"] = route()
            bad += int(not d["encoder_route"].startswith("hip:"))
        print(json.dumps(d), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(json.dumps({"case": name, "error": f"{type(e).__name__}: {e}"[:300]}), flush=True)


for (H, W, B) in ((32, 32, 256), (64, 64, 128), (64, 128, 64), (79, 79, 64), (80, 80, 64), (96, 96, 64), (100, 100, 32), (127, 131, 8), (128, 128, 32),
                  (200, 150, 16), (256, 256, 8), (300, 300, 4), (512, 512, 2), (1024, 1024, 1)):
    pr = syn.random_obstacle_maps(B, H, W, 0.2, seed=3)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    va = VanillaAstar().to(dev).eval()

    def f_va():
        with torch.no_grad():
            va(m, s, g)
    run(f"VanillaAstar eval {H}x{W} B={B}", f_va)
    if H * W <= 100 * 100:
        def f_inter():
            with torch.no_grad():
                va(m, s, g, store_intermediate_results=True)
        run(f"VanillaAstar store_intermediate_results {H}x{W} B={B}", f_inter, n=1)
    for arch in (("CNN", "Unet", "Unet5", "CNNDownSize") if (H, W) == (64, 64) else ("CNN",)):
        torch.manual_seed(0)
        depth = {"CNNDownSize": 2, "Unet5": 5}.get(arch, 4)  # (Unet5: encoder_depth 5, a 16-channel last decoder block)
        arch = "Unet" if arch == "Unet5" else arch
        na = NeuralAstar(encoder_arch=arch, encoder_depth=depth, Tmax=0.25, g_ratio=0.2 if arch == "Unet" else 0.5).to(dev)
        ss, gg = s, g
        if arch == "CNNDownSize":  # the WarCraft arrangement: the search runs on the pooled grid, obstacles are learnt
            ss = torch.zeros(B, 1, H // 4, W // 4, device=dev)
            gg = torch.zeros_like(ss)
            ss[:, 0, 1, 1] = 1
            gg[:, 0, -2, -2] = 1
            na.learn_obstacles = True
        na.eval()

        def f_na():
            with torch.no_grad():
                na(m, ss, gg)
        run(f"NeuralAstar({arch}, depth {depth}) eval {H}x{W} B={B}", f_na, route=lambda: na.last_encoder_route)
        na.train()
        opt = torch.optim.RMSprop(na.parameters(), lr=1e-3)
        traj = (torch.rand_like(ss) < 0.1).float()

        def f_tr():
            opt.zero_grad()
            out = na(m, ss, gg)
            torch.nn.L1Loss()(out.histories, traj).backward()
            opt.step()
        run(f"NeuralAstar({arch}, depth {depth}) training step {H}x{W} B={B}", f_tr, route=lambda: na.last_encoder_route)
print(json.dumps({"scenarios_with_an_error_or_a_fall_back_to_torch_nn": bad}))
sys.exit(1 if bad else 0)
