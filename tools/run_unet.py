"""rocprofv3 target: a few U-Net encoder forwards through the generic MFMA convolution.  Usage: python tools/run_unet.py B precision [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_unet_gpu as T  # noqa: E402
from neural_astar.planner import NeuralAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

B, prec = int(sys.argv[1]), sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
planner = NeuralAstar(encoder_arch="Unet", encoder_depth=4)
planner.encoder = T._calibrated_unet()
planner = planner.to(dev).eval()
planner.encoder_backend = "hip_" + prec
pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
with torch.no_grad():
    for _ in range(reps):
        planner.encode(m, s, g)
torch.cuda.synchronize()
print("ok")
