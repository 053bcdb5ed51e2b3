#!/bin/bash
# round 3, GPU call: one-launch forms of the training step's small work (BatchNorm statistics + coefficients, all weight packs, RMSprop):
# kernel tests, the training-path parity tests, then the step census and the train bench
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_encoder_train_gpu.py tests/test_trainstep_golden_gpu.py tests/test_unet_gpu.py tests/test_distributed_training.py -m gpu -q -x 2>&1 | tail -5
for c in maze warcraft unet; do timeout 250 python tools/probe_train_graph.py $c 100 2>&1 | grep "eager :"; done
for c in maze warcraft; do
  python bench.py --mode train --config $c --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('bench train $c', round(j['ms_per_step'],3), 'ms/step', round(j['value']))"
done
python tools/probe_train.py 100,4096 hip_f16x3 2>&1 | grep -v Warn | tail -4
