"""Dev probe: backward kernels on the 4096-map maze batch (U(0,1) costs), training budget Tmax = 0.25 and eval budget."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import torch
from neural_astar import ops
from neural_astar.utils import synthetic as syn
dev = torch.device("cuda:0")
pr = syn.maze_maps(4096, 32, seed=1234)
m, s, g = (torch.from_numpy(x[:, 0]).to(dev).contiguous() for x in pr)
cost = torch.from_numpy(syn.random_costs(4096, 32, 32, seed=3)[:, 0]).to(dev)

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for B in (100, 4096):
    for mi in (256, 1024):
        cb, sb, gb, mb = cost[:B], s[:B], g[:B], m[:B]
        hist, _, iters, _, log = torch.ops.nastar.astar_forward(cb, sb, gb, mb, 0.5, mi, True)
        gh = torch.randn_like(hist); tb = (iters.amax() - 1).to(torch.int32).reshape(1)
        a = torch.ops.nastar.astar_backward(gh, cb, sb, gb, mb, 0.5, mi, iters, tb)
        r = torch.ops.nastar.astar_backward_replay(gh, cb, sb, gb, mb, log, 0.5, mi, iters, tb)
        err = float((a - r).abs().max()); sc = float(a.abs().max())
        f0 = t(lambda: torch.ops.nastar.astar_forward(cb, sb, gb, mb, 0.5, mi, False))
        f1 = t(lambda: torch.ops.nastar.astar_forward(cb, sb, gb, mb, 0.5, mi, True))
        b0 = t(lambda: torch.ops.nastar.astar_backward(gh, cb, sb, gb, mb, 0.5, mi, iters, tb))
        b1 = t(lambda: torch.ops.nastar.astar_backward_replay(gh, cb, sb, gb, mb, log, 0.5, mi, iters, tb))
        print(f"B={B} max_iters={mi} mean iters {float(iters.float().mean()):.0f}: forward {f0*1e3:.0f} us (with log {f1*1e3:.0f}), "
              f"backward reselect {b0*1e3:.0f} us, replay {b1*1e3:.0f} us; |reselect-replay| {err:.2e} (max|grad| {sc:.2e})")
