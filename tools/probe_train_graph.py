"""Does the whole training step (planner forward, fused L1 loss, backward through search + encoder, RMSprop) replay from a hipGraph,
and what does that buy at the reference's batch of 100?  Eager DataParallelTrainer.train_step vs a captured step (static input
buffers, capturable RMSprop), same weights / same batches: ms per step and the loss after N steps of each.
Usage (GPU box): python tools/probe_train_graph.py [maze|warcraft|unet] [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT]
import bench  # noqa: E402
import bench_extras  # noqa: E402
from neural_astar.planner import NeuralAstar  # noqa: E402
from neural_astar.utils import distributed as D  # noqa: E402
from neural_astar.utils.training import fused_l1_step  # noqa: E402
from neural_astar import ops  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "maze"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
if cfg_name == "unet":
    kw = dict(encoder_input="m+", encoder_arch="Unet", encoder_depth=4, Tmax=0.25)
    data_cfg = "maze"
else:
    kw = bench_extras.TRAIN_CONFIGS[cfg_name]["kw"]
    data_cfg = cfg_name
batches = [bench_extras.train_batch(data_cfg, B, 1234 + 1000 * k, dev) for k in range(4)]


def make():
    torch.manual_seed(1234)
    p = NeuralAstar(**kw).to(dev)
    p.encoder_backend = "hip_f16x3"
    return p


def timed(fn, n=40, warm=10):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        last = fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, last


# ---- eager ----
pe = make()
te = D.DataParallelTrainer(pe, lr=1e-3, coupling="local")
ms_e, loss_e = timed(lambda i: te.train_step(*batches[i % 4]))
print(f"{cfg_name} B={B} eager : {ms_e:.3f} ms/step, loss {float(loss_e):.6f}", flush=True)

# ---- captured ----
pg = make()
pg.train()
opt = torch.optim.RMSprop(pg.parameters(), 1e-3, capturable=True)
static = [x.clone() for x in batches[0]]
prev = ops.BatchCoupling.mode
ops.BatchCoupling.mode = "batch"


def step_body():
    opt.zero_grad(set_to_none=False)
    loss, _ = fused_l1_step(pg, *static)
    loss.backward()
    opt.step()
    return loss.detach()


side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for k in range(3):
        for dst, src in zip(static, batches[k % 4]):
            dst.copy_(src)
        step_body()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    gloss = step_body()
torch.cuda.synchronize()


def replay(i):
    for dst, src in zip(static, batches[i % 4]):
        dst.copy_(src, non_blocking=True)
    graph.replay()
    return gloss


ms_g, loss_g = timed(replay)
ops.BatchCoupling.mode = prev
print(f"{cfg_name} B={B} graph : {ms_g:.3f} ms/step, loss {float(loss_g):.6f}", flush=True)
# same number of optimiser steps on the same batch sequence? eager did 50, the captured model 3 + 1 + 50: compare trajectories instead
pa, pb = make(), make()
ta = D.DataParallelTrainer(pa, lr=1e-3, coupling="local")
la = [float(ta.train_step(*batches[i % 4])) for i in range(6)]
print("eager losses   ", " ".join(f"{x:.6f}" for x in la))
