#!/bin/bash
# round 3, GPU call: generic convolution with the LDS-transposed epilogue: parity (U-Net, training, flat-conv cases) + per-layer / whole-encoder timings
mkdir -p gpurun_out/r03
python -m pytest tests/test_unet_gpu.py tests/test_encoder_train_gpu.py tests/test_trainstep_golden_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/probe_unet.py 4096 2>&1 | grep -v Warn | tail -3
cp gpurun_out/probe_unet.json gpurun_out/r03/probe_unet_epilogue.json
python tools/probe_train.py 100,4096 hip_f16x3,hip_f16 2>&1 | grep -v Warn | tail -4
python tools/probe_train_unet.py 100 hip_f16x3,hip_f16,torch 2>&1 | grep -v Warn | tail -3
