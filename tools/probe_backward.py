import sys
sys.path[:0] = ["/root/repo/neural-astar_amd", "/root/repo"]
import torch
from neural_astar import ops
from neural_astar.utils import synthetic as syn
dev = torch.device("cuda:0")
pr = syn.maze_maps(4096, 32, seed=1234)
m, s, g = (torch.from_numpy(x[:, 0]).to(dev).contiguous() for x in pr)
cost = torch.from_numpy(syn.random_costs(4096, 32, 32, seed=3)[:, 0]).to(dev)
hist, _, iters, _, _ = torch.ops.nastar.astar_forward(cost, s, g, m, 0.5, 256, False)
gh = torch.randn_like(hist); tb = (iters.amax() - 1).to(torch.int32).reshape(1)
for _ in range(3): torch.ops.nastar.astar_backward(gh, cost, s, g, m, 0.5, 256, iters, tb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): torch.ops.nastar.astar_backward(gh, cost, s, g, m, 0.5, 256, iters, tb)
e1.record(); torch.cuda.synchronize()
print("backward ms per 4096 maps: %.4f" % (e0.elapsed_time(e1) / 20))
