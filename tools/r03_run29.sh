#!/bin/bash
# round 3, GPU call: s_setprio 3 around the search loop (flags 64 = without), A/B on the three 4096-map workloads
mkdir -p gpurun_out/r03
for rep in 1 2; do for f in 0 64; do for w in maze32 rand32 rand64; do
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --workload $w > gpurun_out/r03/prio_${w}_f${f}_$rep.json 2>> gpurun_out/r03/prio.err
done; done; done
python - <<'P'
import json
for w in ("maze32","rand32","rand64"):
    for f in (0,64):
        for rep in (1,2):
            j=json.load(open(f"gpurun_out/r03/prio_{w}_f{f}_{rep}.json")); print(w,"flags",f,"rep",rep,round(j["value"]/1e6,2),"M maps/s", round(j["ms_per_step"]*1e3,1), round(j["roofline"]["launch_ms_avg"]*1e3,1),"us avg", round(j["roofline"]["launch_ms_median"]*1e3,1),"median")
P
