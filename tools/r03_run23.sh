#!/bin/bash
# round 3, GPU call: timeline census of one training step (maze, warcraft, unet at batch 100)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
for c in maze warcraft unet; do
  rm -rf /tmp/census_$c
  timeout 250 rocprofv3 --kernel-trace -d /tmp/census_$c -o t --output-format csv -- python $R/tools/train_census.py run $c 100 > /tmp/census_$c.log 2>&1
  F=$(find /tmp/census_$c -name "*kernel_trace.csv" | head -1)
  echo "== $c"; python $R/tools/train_census.py parse "$F" | tee $R/gpurun_out/r03/census_$c.txt | head -45
done
