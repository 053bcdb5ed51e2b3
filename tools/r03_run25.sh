#!/bin/bash
# round 3, GPU call: one-launch BatchNorm statistics with agent-scope atomics instead of fences: kernel tests + step timings
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_encoder_train_gpu.py -m gpu -q -x -k "one_launch or all_weight_packs or fused_rmsprop or training_step_matches or other_encoder or unet_trains" 2>&1 | tail -3
for c in maze warcraft unet; do timeout 250 python tools/probe_train_graph.py $c 100 2>&1 | grep "eager :"; done
python tools/probe_train.py 100,4096 hip_f16x3 2>&1 | grep -v Warn | tail -2
