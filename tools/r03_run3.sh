#!/bin/bash
# round 3, GPU call: narrow-EXEC A/B, aliased cost/passable load, deferred solvability check + hipGraph capture, fixed goldens / sync-BN tests
mkdir -p gpurun_out/r03
python -m pytest tests/test_trainstep_golden_gpu.py tests/test_distributed_training.py -m gpu -q 2>&1 | grep -v Warning | tail -12 > gpurun_out/r03/t3a.log
tail -6 gpurun_out/r03/t3a.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "unsolvable or hipgraph or instruction_streams or golden or full_size or fused or planner_modules" 2>&1 | tail -12 > gpurun_out/r03/t3b.log
tail -6 gpurun_out/r03/t3b.log
for f in 0 32; do
  for w in maze32 rand32 rand64; do
    NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --workload $w > gpurun_out/r03/ab2_${w}_f$f.json 2>> gpurun_out/r03/ab2.err
  done
done
python - <<'P'
import json
for w in ("maze32","rand32","rand64"):
    for f in (0,32):
        try:
            j=json.load(open(f"gpurun_out/r03/ab2_{w}_f{f}.json")); print(w,f,round(j["value"]/1e6,2),"M maps/s", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us", round(j["roofline"]["frac"],4))
        except Exception as e: print(w,f,"ERR",e)
P
