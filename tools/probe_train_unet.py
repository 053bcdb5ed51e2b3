"""Dev probe (GPU box): U-Net encoder training step (forward + backward), MI355X kernels vs torch.nn.  Usage: probe_train_unet.py [B] [backends]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_unet_gpu as TU  # noqa: E402
from neural_astar.planner import NeuralAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
backends = sys.argv[2].split(",") if len(sys.argv) > 2 else ["hip_f16x3", "hip_f16", "torch"]
dev = torch.device("cuda:0")
pr = syn.maze_maps(B, 32, seed=3)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
R = torch.randn((B, 1, 32, 32), device=dev) / (B * 1024)
res = {}
for backend in backends:
    na = NeuralAstar(encoder_arch="Unet", encoder_depth=4)
    na.encoder = TU._calibrated_unet(seed=3)
    na = na.to(dev).train()
    na.encoder_backend = backend

    def one():
        for p in na.parameters():
            p.grad = None
        (na.encode(m, s, g) * R).sum().backward()
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5 if backend != "torch" else 3
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    res[f"B{B}_{backend}_fwd_bwd_ms"] = (time.perf_counter() - t0) / reps * 1e3
    print(backend, res[f"B{B}_{backend}_fwd_bwd_ms"], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"probe_train_unet_B{B}.json"), "w"), indent=1)
