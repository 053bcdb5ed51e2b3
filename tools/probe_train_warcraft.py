"""Dev probe (GPU box): WarCraft encoder (CNNDownSize, rgb+, depth 3, 96x96 -> 12x12; BASELINE config 5) training step, encoder
forward + backward only: MI355X training kernels vs torch.nn.  Usage: python tools/probe_train_warcraft.py [B] [backends]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
from neural_astar.planner import NeuralAstar  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
backends = sys.argv[2].split(",") if len(sys.argv) > 2 else ["hip_f16x3", "hip_f16", "torch"]
dev = torch.device("cuda:0")
img = torch.rand((B, 3, 96, 96), device=dev)
s = torch.zeros((B, 1, 12, 12), device=dev); s[:, 0, 0, 0] = 1
g = torch.zeros((B, 1, 12, 12), device=dev); g[:, 0, 11, 11] = 1
R = torch.randn((B, 1, 12, 12), device=dev) / (B * 144)
res = {}
for backend in backends:
    torch.manual_seed(0)
    na = NeuralAstar(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0).to(dev).train()
    na.encoder_backend = backend

    def one():
        for p in na.parameters():
            p.grad = None
        (na.encode(img, s, g) * R).sum().backward()
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    res[f"B{B}_{backend}_fwd_bwd_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    print(backend, res[f"B{B}_{backend}_fwd_bwd_ms"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"probe_train_warcraft_B{B}.json"), "w"), indent=1)
