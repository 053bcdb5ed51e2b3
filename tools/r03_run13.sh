#!/bin/bash
# round 3, GPU call: the closing BatchNorm + sigmoid block as two launches each way: unit test, composed-step goldens, sync-BN equivalence, timings
mkdir -p gpurun_out/r03
python tools/probe_train_unet.py 100 hip_f16x3,torch 2>&1 | grep -v Warn | tail -2
for c in maze warcraft; do
  python bench.py --mode train --config $c --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03/train4_$c.json 2>/dev/null
  python bench.py --mode train --config $c --steps 20 --warmup 3 --no-cpu-baseline --force-collate > gpurun_out/r03/train4_${c}_rccl1.json 2>/dev/null
done
python - <<'P'
import json
for c in ("maze","warcraft","maze_rccl1","warcraft_rccl1"):
    j=json.load(open(f"gpurun_out/r03/train4_{c}.json")); print(c, round(j["ms_per_step"],3), round(j["value"]))
P
