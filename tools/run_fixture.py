"""Dev probe target for rocprofv3 PMC runs: a few launches of one configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np, torch
from neural_astar.utils import synthetic as syn
from neural_astar import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fixture64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
if kind == "fixture64": pr = syn.fixture_block(B, 64, 64)
elif kind == "maze32": pr = syn.maze_maps(B, 32, seed=1234)
else: pr = syn.random_obstacle_maps(B, 32, 32, 0.25, seed=1234)
m, s, g = (torch.from_numpy(x[:, 0]).to(dev) for x in pr)
for _ in range(3):
    out = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, m.shape[-1] ** 2, False)
torch.cuda.synchronize()
print(kind, B, "iters sum", int(out[2].sum()), "max", int(out[2].max()))
