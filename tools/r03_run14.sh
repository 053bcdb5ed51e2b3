#!/bin/bash
# round 3, GPU call: 128-channel workgroups for the generic convolution (NASTAR_ENCODER_FLAGS=8192): parity + U-Net / training timings vs default
mkdir -p gpurun_out/r03
NASTAR_ENCODER_FLAGS=8192 python -m pytest tests/test_unet_gpu.py tests/test_encoder_train_gpu.py -m gpu -q -x 2>&1 | tail -2
for f in 0 8192; do
  echo "== flags $f"
  NASTAR_ENCODER_FLAGS=$f python tools/probe_unet.py 4096 2>&1 | grep -v Warn | tail -6
  cp gpurun_out/probe_unet.json gpurun_out/r03/probe_unet_f$f.json
  NASTAR_ENCODER_FLAGS=$f python tools/probe_train.py 4096 hip_f16x3,hip_f16 2>&1 | grep -v Warn | tail -2
done
