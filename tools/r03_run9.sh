#!/bin/bash
# round 3, GPU call: multi-tensor weight maxima: encoder-training parity, step timings (CNN, WarCraft, U-Net)
mkdir -p gpurun_out/r03
python -m pytest tests/test_encoder_train_gpu.py tests/test_trainstep_golden_gpu.py tests/test_distributed_training.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r03/t9.log
tail -2 gpurun_out/r03/t9.log
python tools/probe_train_unet.py 2>&1 | grep -v Warn | tail -4
python tools/probe_train.py 100 hip_f16x3,hip_f16 2>&1 | grep -v Warn | tail -2
for c in maze warcraft; do
  python bench.py --mode train --config $c --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r03/train3_$c.json 2>/dev/null
done
python - <<'P'
import json
for c in ("maze","warcraft"):
    j=json.load(open(f"gpurun_out/r03/train3_{c}.json")); print(c, round(j["ms_per_step"],3), round(j["value"]))
P
