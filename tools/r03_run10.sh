#!/bin/bash
# round 3, GPU call: ONE LDS atomic per step (chunk re-insertion + relaxed neighbours under a merged EXEC): parity + timing
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or oracle or instruction_streams or full_size or mazes_and_train" 2>&1 | tail -3 > gpurun_out/r03/t10.log
tail -2 gpurun_out/r03/t10.log
for w in maze32 rand32 rand64; do
  python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --workload $w > gpurun_out/r03/c13_${w}.json 2>> gpurun_out/r03/c13.err
done
NASTAR_FORWARD_FLAGS=16 python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > gpurun_out/r03/c13_maze32_v2.json 2>> gpurun_out/r03/c13.err
python tools/probe_latency.py 2>&1 | grep "fixture\|B=4096\|B=1:"
python - <<'P'
import json
for w in ("maze32","rand32","rand64","maze32_v2"):
    j=json.load(open(f"gpurun_out/r03/c13_{w}.json")); print(w,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1), round(j["roofline"]["launch_ms_median"]*1e3,1),"us median", round(j["roofline"]["frac"],4))
P
