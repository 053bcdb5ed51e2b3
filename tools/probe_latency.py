"""Dev probe: per-iteration latency / throughput of the forward kernel at several batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import numpy as np, torch
from neural_astar.utils import synthetic as syn
from neural_astar import ops

dev = torch.device("cuda:0")
def run(pr, reps=20):
    m, s, g = (torch.from_numpy(x[:, 0]).to(dev) for x in pr)
    for _ in range(3):
        out = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, m.shape[-1] ** 2, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); out = torch.ops.nastar.astar_forward(m, s, g, m, 0.5, m.shape[-1] ** 2, False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    it = out[2].cpu().numpy()
    return min(ts) * 1e3, it

fx = syn.fixture_block(1, 64, 64)
us, it = run(fx); print(f"fixture64 B=1: {us:.1f} us, iters {it[0]} -> {us*1e3/it[0]:.1f} ns/iter")
fx32 = syn.fixture_block(1, 32, 32)
us, it = run(fx32); print(f"fixture32 B=1: {us:.1f} us, iters {it[0]} -> {us*1e3/it[0]:.1f} ns/iter")
mz = syn.maze_maps(4096, 32, seed=1234)
order = None
for B in (1, 64, 256, 512, 1024, 2048, 2304, 4096):
    pr = syn.Problems(*(x[:B] for x in mz))
    us, it = run(pr)
    print(f"maze32 B={B}: {us:.1f} us  iters sum {it.sum()} max {it.max()}  -> {us*1e3/it.max():.1f} ns per max-iter, {it.sum()/us/1e3:.3f} Gexp/s")
# all maps identical length: throughput without tail
one = syn.Problems(*(np.repeat(x[:1], 4096, 0) for x in mz))
for B in (256, 1024, 2304, 4096):
    pr = syn.Problems(*(x[:B] for x in one))
    us, it = run(pr)
    print(f"same-map B={B}: {us:.1f} us iters each {it[0]} -> {us*1e3/it[0]:.1f} ns/iter-round, {it.sum()/us/1e3:.3f} Gexp/s")
