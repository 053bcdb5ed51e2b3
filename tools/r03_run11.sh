#!/bin/bash
# round 3, GPU call: the dive at 32x32 (NASTAR_FLAG_DIVE_SMALL = 64) vs the default loop: maze32, rand32, Tmax 0.25
mkdir -p gpurun_out/r03
NASTAR_FORWARD_FLAGS=64 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or oracle or full_size or mazes_and_train" 2>&1 | tail -2
for f in 0 64; do for w in maze32 rand32; do
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --workload $w > gpurun_out/r03/d32_${w}_f$f.json 2>> gpurun_out/r03/d32.err
done; done
python - <<'P'
import json
for w in ("maze32","rand32"):
    for f in (0,64):
        j=json.load(open(f"gpurun_out/r03/d32_{w}_f{f}.json")); print(w,f,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1), round(j["roofline"]["launch_ms_median"]*1e3,1),"us median")
P
