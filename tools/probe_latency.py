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
# ONE long search on a busy chip at its working clocks: the longest maze of the bench batch in workgroup 0, every other map of the 4096
# solved in 2 steps (start next to goal) -- launch time = fixed part + max-iter x (step of a wavefront that has its SIMD to itself).
# (The B = 1 line above runs on an otherwise idle GPU whose clocks never leave the idle state: ~280 ns per step there.)
its_all = run(mz)[1]
rank = np.argsort(-its_all)
triv = syn.Problems(*(np.repeat(x[:1], 4096, 0).copy() for x in mz))
triv.map_designs[:] = 1.0
triv.start_maps[:] = 0.0
triv.goal_maps[:] = 0.0
triv.start_maps[:, 0, 5, 5] = 1.0
triv.goal_maps[:, 0, 5, 6] = 1.0
us_t, _ = run(triv, reps=200)
print(f"4096 two-step maps: {us_t:.1f} us")
pts = []
for q in (0, 200, 1000, 2000, 3000):  # searches of decreasing length, each alone (workgroup 0) among the trivial ones
    i = int(rank[q])
    triv.map_designs[0], triv.start_maps[0], triv.goal_maps[0] = mz.map_designs[i], mz.start_maps[i], mz.goal_maps[i]
    us, it = run(triv, reps=200)
    pts.append((int(it.max()), us))
    print(f"one maze of {it.max()} steps at workgroup 0 among 4095 two-step maps: {us:.1f} us")
x = np.array([p[0] for p in pts], float)
y = np.array([p[1] for p in pts], float)
sl, ic = np.polyfit(x, y, 1)
print(f"least-squares line through those points: {sl * 1e3:.1f} ns per step (a wavefront with its SIMD to itself, chip at working clocks) + {ic:.1f} us fixed")
