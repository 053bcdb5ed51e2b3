#!/bin/bash
# round 3, GPU call: the 64x64 "dive": parity (goldens, oracle, stream equality incl. logs) and A/B timing vs NASTAR_FLAG_NO_DIVE (32)
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or oracle or instruction_streams or full_size or backward" 2>&1 | tail -4 > gpurun_out/r03/t8.log
tail -3 gpurun_out/r03/t8.log
for f in 0 32; do
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 5 --workload rand64 > gpurun_out/r03/dive_rand64_f$f.json 2>> gpurun_out/r03/dive.err
done
NASTAR_FORWARD_FLAGS=0 python tools/probe_latency.py 2>&1 | grep fixture
python - <<'P'
import json
for f in (0,32):
    j=json.load(open(f"gpurun_out/r03/dive_rand64_f{f}.json")); print("rand64 flags",f,round(j["value"]/1e6,2),"M maps/s", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us", round(j["roofline"]["frac"],4))
P
