#!/bin/bash
# round 3, GPU call: parity of the re-scheduled select phase, then the round's profile evidence and the driver-style default bench
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "instruction_streams or golden or full_size or mazes_and_train or unsolvable or intermediate" 2>&1 | tail -4 > gpurun_out/r03/t5.log
tail -3 gpurun_out/r03/t5.log
bash tools/profile_round.sh r03 > gpurun_out/r03/profile_round.log 2>&1
tail -5 gpurun_out/r03/profile_round.log
timeout 600 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_default.json"))
for k in ("value","ms_per_step","roofline","cpu_baseline","through_module","expansions_per_s"): print(k, json.dumps(d.get(k))[:900])
for s_ in d.get("secondary",[]): print(json.dumps(s_)[:300])
for k,v in d.get("extra",{}).items(): print(k, json.dumps(v)[:300])
P
