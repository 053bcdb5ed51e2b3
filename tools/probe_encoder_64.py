import os, sys
sys.path[:0] = ["/root/repo/neural-astar_amd", "/root/repo"]
import torch
from neural_astar.utils import synthetic as syn
from neural_astar.planner import NeuralAstar
dev = torch.device("cuda:0")
for (B, H) in ((1024, 64),):
    pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=1)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    na = NeuralAstar(encoder_arch="CNN").to(dev).eval(); na.encoder_backend = "hip_bf16"
    with torch.no_grad():
        for _ in range(3): na.encode(m, s, g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): na.encode(m, s, g)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flop = 2 * B * H * H * 9 * (2 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256)
    print(f"B={B} {H}x{H}: {ms:.3f} ms  {flop/ms/1e9:.0f} TFLOP/s useful")
