#!/bin/bash
# round 3, GPU call: training-step launch consolidation that survived (all weight packs in one launch, fused RMSprop, multi-tensor
# counter increments / bias-gradient zeroing, no scale clones): parity tests + step timings + census
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_encoder_train_gpu.py tests/test_trainstep_golden_gpu.py tests/test_unet_gpu.py tests/test_distributed_training.py -m gpu -q -x 2>&1 | tail -3
for c in maze warcraft unet; do timeout 250 python tools/probe_train_graph.py $c 100 2>&1 | grep "eager :"; done
python tools/probe_train.py 100,4096 hip_f16x3 2>&1 | grep -v Warn | tail -2
bash tools/r03_run23.sh > gpurun_out/r03/census_after.txt 2>&1; grep -A3 "^==" gpurun_out/r03/census_after.txt | grep "==\|wall\|launches"
