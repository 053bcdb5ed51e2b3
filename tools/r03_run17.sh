#!/bin/bash
# round 3, GPU call: every world > 1 branch of bench.py on the 1-GPU box: 2 ranks sharing cuda:0 over gloo (timings meaningless, code paths real)
mkdir -p gpurun_out/r03
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577"
$L bench.py --gpus 2 --steps 20 --warmup 3 --dist-backend gloo --share-gpu > gpurun_out/r03/world2_forward.json 2> gpurun_out/r03/world2_forward.err; echo "forward rc=$?"
$L bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --share-gpu --workload rand64 --global-batch 2048 --shard interleaved > gpurun_out/r03/world2_rand64.json 2> gpurun_out/r03/world2_rand64.err; echo "rand64 rc=$?"
$L bench.py --gpus 2 --mode train --config warcraft --steps 10 --warmup 2 --dist-backend gloo --share-gpu > gpurun_out/r03/world2_train_warcraft.json 2> gpurun_out/r03/world2_train_warcraft.err; echo "train rc=$?"
$L bench.py --gpus 2 --mode train --config maze --steps 10 --warmup 2 --dist-backend gloo --share-gpu > gpurun_out/r03/world2_train_maze.json 2> gpurun_out/r03/world2_train_maze.err; echo "train rc=$?"
for f in forward rand64 train_warcraft train_maze; do echo "== $f"; cut -c1-330 gpurun_out/r03/world2_$f.json; python - <<P
import json
try:
    j=json.load(open("gpurun_out/r03/world2_$f.json")); print(j["n_gpus"], j["config"].get("parallelism"), j["config"].get("collate","")[:120])
except Exception as e: print("ERR", e)
P
tail -3 gpurun_out/r03/world2_$f.err | cut -c1-300; done
