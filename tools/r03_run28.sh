#!/bin/bash
# round 3, GPU call: tiled multi-tensor weight pack (LDS transpose): pack tests, training parity, U-Net / CNN step timing, bench contract tests
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_encoder_train_gpu.py tests/test_trainstep_golden_gpu.py tests/test_unet_gpu.py tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -3
for c in maze unet; do timeout 250 python tools/probe_train_graph.py $c 100 2>&1 | grep "eager :"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/census_unet; timeout 250 rocprofv3 --kernel-trace -d /tmp/census_unet -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/train_census.py run unet 100 > /tmp/census_unet.log 2>&1
python $GRAFT_REPO_ROOT/tools/train_census.py parse $(find /tmp/census_unet -name "*kernel_trace.csv" | head -1) | grep "launches\|wall\|pack_weight\|rmsprop\|absmax"
