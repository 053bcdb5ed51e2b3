"""Dev probe: where do the cycles of the fused 128->256(+1) conv kernel go?  NASTAR_ENCODER_FLAGS=256 makes every wave write
{total, barrier wait, slice loops, epilogue} s_memtime totals into the unused 256-channel output slab."""
import os, sys
os.environ["NASTAR_ENCODER_FLAGS"] = "256"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np, torch
from neural_astar.utils import synthetic as syn
from neural_astar.planner import NeuralAstar
dev = torch.device("cuda:0")
B = 1024
pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
na = NeuralAstar(encoder_arch="CNN").to(dev).eval(); na.encoder_backend = "hip_bf16"
with torch.no_grad():
    for _ in range(3): na.encode(m, s, g)
torch.cuda.synchronize()
enc = na._hip_encoder if hasattr(na, "_hip_encoder") else None
if enc is None:
    enc = [v for v in vars(na).values() if hasattr(v, "_ws")][0]
ws = enc._ws
off = B * 1024 * (16 + 128) * 2
d = ws[off: off + 256 * 8 * 4 * 8].view(torch.int64).cpu().numpy().reshape(256, 8, 4)
tot, bar, main, epi = (d[..., i].astype(np.float64) for i in range(4))
print("per-wave cycles (mean over 256 WGs x 8 waves): total %.0f  barrier %.0f (%.1f%%)  slices %.0f (%.1f%%)  epilogue %.0f (%.1f%%)" % (
    tot.mean(), bar.mean(), 100 * bar.mean() / tot.mean(), main.mean(), 100 * main.mean() / tot.mean(), epi.mean(), 100 * epi.mean() / tot.mean()))
print("by wave: barrier", bar.mean(0).round(0), "slices", main.mean(0).round(0))
print("MFMA floor per wave: 4 images x 4 groups x 8 slices x 72 MFMA x 32 cyc x 2 waves/SIMD =", 4 * 4 * 8 * 72 * 32 * 2)
