#!/bin/bash
# round 3, GPU call: the round-3b step (<= 32x32): parity + A/B against NASTAR_FLAG_NO_DIVE (32) = the plain round-3 loop
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_reference_suite_gpu.py -m gpu -q -x -k "golden or oracle or instruction_streams or full_size or mazes_and_train or backward or unsolvable or suite or boundary" 2>&1 | tail -3
for f in 0 32; do for w in maze32 rand32; do
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --workload $w > gpurun_out/r03/er_${w}_f$f.json 2>> gpurun_out/r03/er.err
done; done
python - <<'P'
import json
for w in ("maze32","rand32"):
    for f in (0,32):
        j=json.load(open(f"gpurun_out/r03/er_{w}_f{f}.json")); print(w,f,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1), round(j["roofline"]["launch_ms_median"]*1e3,1),"us median", round(j["roofline"]["frac"],4))
P
for f in 0 32; do echo flags $f; NASTAR_FORWARD_FLAGS=$f timeout 200 python tools/probe_latency.py 2>&1 | grep "fixture\|B=1:\|B=256\|B=4096"; done
