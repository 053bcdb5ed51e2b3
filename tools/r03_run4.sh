#!/bin/bash
# round 3, GPU call: the whole GPU suite + smoke on the current build, then the three forward workloads
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -15 > gpurun_out/r03/t4.log
tail -8 gpurun_out/r03/t4.log
python __graft_entry__.py smoke 2>&1 | tail -2
for w in maze32 rand32 rand64; do
  python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --workload $w > gpurun_out/r03/c5_${w}.json 2>> gpurun_out/r03/c5.err
done
python - <<'P'
import json
for w in ("maze32","rand32","rand64"):
    try:
        j=json.load(open(f"gpurun_out/r03/c5_{w}.json")); print(w,round(j["value"]/1e6,2),"M maps/s", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us", round(j["roofline"]["frac"],4))
    except Exception as e: print(w,"ERR",e)
P
