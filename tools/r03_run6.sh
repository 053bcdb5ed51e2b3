#!/bin/bash
# round 3, GPU call: driver-style default bench + fused-step tests after the autograd-node change
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py tests/test_trainstep_golden_gpu.py tests/test_data_path.py -m gpu -q -x -k "fused or trainstep or training or backward or planner_module" 2>&1 | tail -4 > gpurun_out/r03/t6.log
tail -3 gpurun_out/r03/t6.log
timeout 900 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r03/bench_default.json"))
for k in ("value","ms_per_step","roofline","cpu_baseline","through_module","expansions_per_s"): print(k, json.dumps(d.get(k))[:1200])
for s_ in d.get("secondary",[]): print(json.dumps(s_)[:300])
for k,v in d.get("extra",{}).items(): print(k, json.dumps(v)[:400])
P
tail -3 gpurun_out/r03/bench_default.err
