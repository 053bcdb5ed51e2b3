"""Dev probe: NeuralAstar.forward time split -- torch/MIOpen encoder vs the bf16-MFMA HIP encoder vs the HIP search."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import torch
from neural_astar.utils import synthetic as syn
from neural_astar.planner import NeuralAstar
dev = torch.device("cuda:0")
pr = syn.maze_maps(4096, 32, seed=1234)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
na = NeuralAstar(encoder_arch="CNN").to(dev).eval()
na.astar.check_solvable = False
with torch.no_grad():
    enc = t(lambda: na.encode(m, s, g))
    full = t(lambda: na(m, s, g))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        enc_bf16 = t(lambda: na.encode(m, s, g))
    na.encoder_backend = "hip_bf16"
    enc_hip = t(lambda: na.encode(m, s, g), reps=10)
    full_hip = t(lambda: na(m, s, g), reps=10)
flop = 2 * 4096 * 1024 * 9 * (2 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256 * 1)
print(f"B=4096 32x32 CNN encoder: torch fp32 {enc:.2f} ms, torch bf16 autocast {enc_bf16:.2f} ms, HIP bf16 MFMA {enc_hip:.2f} ms "
      f"({flop / enc_hip / 1e9:.0f} TFLOP/s useful); NeuralAstar.forward: torch encoder {full:.2f} ms, HIP encoder {full_hip:.2f} ms "
      f"-> {4096 / full_hip / 1e3:.2f} M maps/s end to end")
