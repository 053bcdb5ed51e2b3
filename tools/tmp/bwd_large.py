import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neural-astar_amd")]
import torch
from neural_astar import ops
from neural_astar.planner.differentiable_astar import DifferentiableAstar
from neural_astar.utils import synthetic as syn
dev = torch.device("cuda:0")
for (H, B, Tmax) in ((128, 256, 1.0), (256, 64, 1.0), (512, 64, 1.0), (1024, 16, 1.0)):
    pr = syn.random_obstacle_maps(B, H, H, 0.2, seed=7)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    cost = (torch.from_numpy(syn.random_costs(B, H, H, seed=3)).to(dev) * 0.2 + m * 0.8).requires_grad_(True)  # mostly the map: long searches
    da = DifferentiableAstar(0.5, Tmax).to(dev).train(Tmax < 1.0)
    def fwd():
        return da(cost, s, g, m)
    out = fwd(); out.histories.sum().backward(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    n = 5
    for _ in range(n):
        cost.grad = None
        e[0].record(); out = fwd(); e[1].record(); out.histories.sum().backward(); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    it = da.last_iters
    print(json.dumps({"H": H, "B": B, "Tmax": Tmax, "in_lds": ops.in_lds(H, H), "fwd_ms": tf / n, "bwd_ms": tb / n, "max_iters": int(it.max()), "sum_iters": int(it.sum()),
                      "bwd_ns_per_step_of_longest": tb / n * 1e6 / int(it.max())}), flush=True)
