#!/bin/bash
# dev probe: shader clock / power while the HIP encoder runs in a loop (is the matrix pipe power-throttled?)
R=${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY &
import sys, time, os
sys.path[:0] = ["$R/neural-astar_amd", "$R"]
import torch
from neural_astar.utils import synthetic as syn
from neural_astar.planner import NeuralAstar
dev = torch.device("cuda:0")
pr = syn.random_obstacle_maps(4096, 32, 32, 0.25, seed=1)
m, s, g = (torch.from_numpy(x).to(dev) for x in pr)
na = NeuralAstar(encoder_arch="CNN").to(dev).eval(); na.encoder_backend = "hip_bf16"
with torch.no_grad():
    t0 = time.time()
    while time.time() - t0 < 8:
        for _ in range(50): na.encode(m, s, g)
        torch.cuda.synchronize()
PY
PID=$!
sleep 4
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | head -8; sleep 0.7; done
wait $PID
