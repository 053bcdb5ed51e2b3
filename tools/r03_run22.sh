#!/bin/bash
# round 3, GPU call: whole training step captured into a hipGraph vs eager, batch 100
mkdir -p gpurun_out/r03
for c in maze warcraft unet; do timeout 250 python tools/probe_train_graph.py $c 100 2>&1 | grep -v "amdgpu.ids" | tail -6; done | tee gpurun_out/r03/train_graph.txt
