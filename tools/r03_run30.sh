#!/bin/bash
# round 3, GPU call: does s_setprio 3 around the search loop (flags 64 = without) cost the pipelined mode (3-4 batches in flight)?
mkdir -p gpurun_out/r03
for rep in 1 2; do for f in 0 64; do
NASTAR_FORWARD_FLAGS=$f python - <<P
import sys, os, torch
sys.path[:0] = ["neural-astar_amd", "."]
import bench
dev = torch.device("cuda:0")
pr = bench.make_problem("maze32", 4096, seed=1234)
run = bench.Runner(pr, dev); bench.prewarm(run, dev, 0.5)
out = {k: round(bench.multi_stream_throughput(pr, 300, dev, k) / 1e6, 2) for k in (1, 2, 3, 4)}
print("flags", os.environ["NASTAR_FORWARD_FLAGS"], "rep $rep", out, flush=True)
P
done; done 2>&1 | grep flags | tee gpurun_out/r03/prio_streams.txt
