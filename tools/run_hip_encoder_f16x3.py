"""Dev probe: time the fp16 HIP encoders: run_hip_encoder_f16x3.py [B] [hip_f16x3|hip_f16|hip_bf16]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import torch
from neural_astar.utils import synthetic as syn
from neural_astar.planner import NeuralAstar
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
backend = sys.argv[2] if len(sys.argv) > 2 else "hip_f16x3"
na.encoder_backend = backend
with torch.no_grad():
    for _ in range(3): c = na.encode(m, s, g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    e0.record()
    for _ in range(5): c = na.encode(m, s, g)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
na.encoder_backend = "torch"  # (16 maps only: MIOpen searches its kernels on the first call of a shape)
with torch.no_grad():
    r = na.encode(m[:16], s[:16], g[:16])
print(f"{backend} encoder B={B}: {ms:.3f} ms; max |err| vs torch fp32 on 16 maps: {float((c[:16] - r).abs().max()):.2e}")
