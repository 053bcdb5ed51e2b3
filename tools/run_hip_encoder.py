"""Dev probe target: only the HIP encoder (no torch convs, so a rocprofv3 trace shows just our kernels)."""
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
na.encoder_backend = "hip_bf16"
with torch.no_grad():
    for _ in range(6):
        c = na.encode(m, s, g)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    e0.record()
    for _ in range(10): c = na.encode(m, s, g)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
flop = 2 * B * 1024 * 9 * (2 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256 * 1)
print(f"HIP encoder B={B}: {ms:.3f} ms  ({flop / ms / 1e9:.0f} TFLOP/s useful)")
