"""Dev probe: 64x64 forward throughput, backward throughput, planner-level (Python boundary) overhead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np, torch
from neural_astar.utils import synthetic as syn
from neural_astar import ops
from neural_astar.planner import VanillaAstar
from neural_astar.planner.differentiable_astar import DifferentiableAstar
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
def load(pr): return tuple(torch.from_numpy(x).to(dev) for x in pr)
# 64x64 forward
pr = syn.random_obstacle_maps(4096, 64, 64, 0.2, seed=1234)
m, s, g = load(pr)
f = lambda: torch.ops.nastar.astar_forward(m[:, 0], s[:, 0], g[:, 0], m[:, 0], 0.5, 4096, False)
out = f(); it = out[2].float()
us, med = timeit(f)
print(f"fwd 64x64 rand p=0.2 B=4096: {us:.1f} us (median {med:.1f}) -> {4096/us:.2f} Mmaps/s, iters mean {it.mean():.1f} max {int(it.max())}, HBM frac {4096*28*4096/us*1e6/8e12:.3f}")
# backward 32x32 (eval-mode search, random upstream gradient), via the autograd op
for kind in ("maze32", "rand32"):
    pr = syn.maze_maps(4096, 32, seed=1234) if kind == "maze32" else syn.random_obstacle_maps(4096, 32, 32, 0.25, seed=1234)
    m, s, g = load(pr)
    cost = torch.from_numpy(syn.random_costs(4096, 32, 32, seed=3)).to(dev)
    for T, training in ((1.0, False), (0.25, True)):
        da = DifferentiableAstar(0.5, T).to(dev); da.train(training)
        mi = ops.max_iters_for(32, T, training)
        ff = lambda: torch.ops.nastar.astar_forward(cost[:, 0], s[:, 0], g[:, 0], m[:, 0], 0.5, mi, False)
        hist, paths, iters, status, _ = ff()
        gh = torch.randn_like(hist); tb = (iters.amax() - 1).to(torch.int32).reshape(1)
        bf = lambda: torch.ops.nastar.astar_backward(gh, cost[:, 0], s[:, 0], g[:, 0], m[:, 0], 0.5, mi, iters, tb)
        uf, _ = timeit(ff); ub, _ = timeit(bf)
        print(f"{kind} ucost Tmax={T} train={training}: fwd {uf:.1f} us, bwd {ub:.1f} us, iters mean {iters.float().mean():.1f} max {int(iters.max())}")
# planner-level call (Python boundary incl. status check sync)
pr = syn.maze_maps(4096, 32, seed=1234); m, s, g = load(pr)
va = VanillaAstar().to(dev).eval()
for chk in (True, False):
    va.astar.check_solvable = chk
    for _ in range(5): o = va(m, s, g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): o = va(m, s, g)
    torch.cuda.synchronize(); print(f"VanillaAstar.forward wall per call (check_solvable={chk}): {(time.perf_counter()-t0)/50*1e6:.1f} us (kernel ~285)")
mm, ss, gg = m[:, 0].contiguous(), s[:, 0].contiguous(), g[:, 0].contiguous()
for _ in range(5): torch.ops.nastar.astar_forward(mm, ss, gg, mm, 0.5, 1024, False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): torch.ops.nastar.astar_forward(mm, ss, gg, mm, 0.5, 1024, False)
torch.cuda.synchronize(); print(f"torch.ops.nastar.astar_forward wall per call: {(time.perf_counter()-t0)/50*1e6:.1f} us")
