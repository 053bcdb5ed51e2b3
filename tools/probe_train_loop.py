"""Dev probe (GPU box): ONE full training step of the reference's maze configuration (scripts/config/train.yaml: CNN depth 4, m+, Tmax
0.25, batch 100, RMSprop lr 1e-3) -- planner forward, L1 loss on histories, backward, optimiser step -- three ways on the same MI355X:
  reference   the reference's OWN DifferentiableAstar (staged oracle/_ref module: ~45 ATen ops per search iteration, autograd tape)
              behind this package's torch.nn CNN encoder = what utils/training.py:55-61 runs through PyTorch-ROCm
  hip_search  this package's fused search step (HIP forward + replay backward) with the torch.nn encoder
  hip_all     ... and the encoder's forward + backward on the MI355X training kernels (encoder_backend = hip_f16x3 / hip_f16)
Usage: python tools/probe_train_loop.py [B]   (writes gpurun_out/probe_train_loop.json)"""
import importlib.util
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]
from neural_astar.planner import NeuralAstar, VanillaAstar  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402
from neural_astar.utils.training import fused_l1_step  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
WARCRAFT = len(sys.argv) > 2 and sys.argv[2] == "warcraft"   # BASELINE config 5: CNNDownSize, 96x96 RGB -> 12x12, learn_obstacles
dev = torch.device("cuda:0")
if WARCRAFT:
    m = torch.rand((B, 3, 96, 96), device=dev)
    s = torch.zeros((B, 1, 12, 12), device=dev)
    g = torch.zeros((B, 1, 12, 12), device=dev)
    s[:, 0, 0, 0] = 1
    g[:, 0, 11, 11] = 1
    traj = torch.zeros((B, 1, 12, 12), device=dev)
    traj[:, 0, torch.arange(12), torch.arange(12)] = 1
else:
    pr = syn.maze_maps(B, 32, seed=7)
    m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
    with torch.no_grad():
        traj = VanillaAstar().to(dev).eval()(m, s, g).paths.float()
res = {"batch": B, "config": "warcraft (train_warcraft.yaml)" if WARCRAFT else "maze (train.yaml)"}


def timed(step, reps):
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name in ("hip_all_f16x3", "hip_all_f16", "hip_search", "reference"):
    torch.manual_seed(0)
    if WARCRAFT:
        na = NeuralAstar(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0, Tmax=0.25,
                         learn_obstacles=True).to(dev).train()
    else:
        na = NeuralAstar(encoder_arch="CNN", Tmax=0.25).to(dev).train()
    opt = torch.optim.RMSprop(na.parameters(), 1e-3)
    if name == "reference":
        spec = importlib.util.spec_from_file_location("ref_da", os.path.join(ROOT, "oracle", "_ref", "differentiable_astar.py"))
        if not os.path.exists(spec.origin):
            continue
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        astar = ref.DifferentiableAstar(g_ratio=0.5, Tmax=0.25).to(dev).train()

        def step():
            opt.zero_grad(set_to_none=True)
            cost = na.encode(m, s, g)
            out = astar(cost, s, g, torch.ones_like(s) if WARCRAFT else m)
            nn.L1Loss()(out.histories, traj).backward()
            opt.step()
        reps = 2
    else:
        na.encoder_backend = {"hip_all_f16x3": "hip_f16x3", "hip_all_f16": "hip_f16", "hip_search": "torch"}[name]

        def step():
            opt.zero_grad(set_to_none=True)
            loss, _ = fused_l1_step(na, m, s, g, traj)
            loss.backward()
            opt.step()
        reps = 10
    res[name + "_ms_per_step"] = timed(step, reps)
    print(name, res[name + "_ms_per_step"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe_train_loop_warcraft.json" if WARCRAFT else "probe_train_loop.json"), "w"), indent=1)
