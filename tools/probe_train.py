"""Dev probe (GPU box): encoder training step on the HIP kernels vs torch autograd -- per-tensor gradient errors and step timings.
Usage: python tools/probe_train.py [B]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_encoder_train_gpu as T  # noqa: E402
from neural_astar.utils import synthetic as syn  # noqa: E402


def step_ms(na, m, s, g, R, reps=5):
    def one():
        for p in na.parameters():
            p.grad = None
        (na.encode(m, s, g) * R).sum().backward()
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    res = {}
    backends = sys.argv[2].split(",") if len(sys.argv) > 2 else ["hip_f16x3", "hip_f16", "torch"]
    for B in ([int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [100, 1024, 4096]):
        pr = syn.maze_maps(B, 32, seed=3)
        m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
        R = torch.randn((B, 1, 32, 32), device=dev) / (B * 1024)
        for backend in backends:
            na = T._shipped_cnn_planner().to(dev).train()
            na.encoder_backend = backend
            try:
                res[f"B{B}_{backend}_fwd_bwd_ms"] = step_ms(na, m, s, g, R, reps=3 if backend == "torch" else 5)
            except Exception as e:  # noqa: BLE001
                res[f"B{B}_{backend}_fwd_bwd_ms"] = repr(e)
            print(B, backend, res[f"B{B}_{backend}_fwd_bwd_ms"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe_train.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
