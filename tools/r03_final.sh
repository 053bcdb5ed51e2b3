#!/bin/bash
# round 3, final GPU session: whole GPU suite + smoke, the profile round, the driver-style bench and both training benches
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -6 > gpurun_out/r03/t_final.log
tail -3 gpurun_out/r03/t_final.log
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/profile_round.sh r03 > gpurun_out/r03/profile_round.log 2>&1
bash tools/r03_bench.sh 2>&1 | tail -40
