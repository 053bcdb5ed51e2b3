"""rocprofv3 target: a few encoder training steps (forward + backward through the HIP kernels) at a given batch and precision.
Usage: python tools/train_step_profile.py B precision(f16|f16x3) [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_encoder_train_gpu as T  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402

B, prec = int(sys.argv[1]), sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=3)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
R = torch.randn((B, 1, 32, 32), device=dev) / (B * 1024)
na = T._shipped_cnn_planner().to(dev).train()
na.encoder_backend = "hip_" + prec
for _ in range(steps):
    for p in na.parameters():
        p.grad = None
    (na.encode(m, s, g) * R).sum().backward()
torch.cuda.synchronize()
print("ok")
