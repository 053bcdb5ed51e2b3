#!/bin/bash
# round 3, GPU call: asm v3 parity (golden + oracle + stream equality) and A/B timing v3 (flags 0) vs v2 (flags 16)
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or oracle or instruction_streams or mazes_and_train or unsolvable or full_size" 2>&1 | tail -15 > gpurun_out/r03/t2.log
tail -8 gpurun_out/r03/t2.log
for f in 0 16; do
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > gpurun_out/r03/ab_maze32_f$f.json 2> gpurun_out/r03/ab_maze32_f$f.err
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --workload rand32 > gpurun_out/r03/ab_rand32_f$f.json 2>> gpurun_out/r03/ab_maze32_f$f.err
  NASTAR_FORWARD_FLAGS=$f python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 5 --workload rand64 > gpurun_out/r03/ab_rand64_f$f.json 2>> gpurun_out/r03/ab_maze32_f$f.err
  NASTAR_FORWARD_FLAGS=$f python tools/probe_latency.py 2>&1 > gpurun_out/r03/lat_f$f.txt
done
python - <<'P'
import json
for w in ("maze32","rand32","rand64"):
    for f in (0,16):
        try:
            j=json.load(open(f"gpurun_out/r03/ab_{w}_f{f}.json")); print(w,f,round(j["value"]/1e6,2),"M maps/s", round(j["roofline"]["launch_ms_avg"]*1e3,1),"us", round(j["roofline"]["frac"],4))
        except Exception as e: print(w,f,"ERR",e)
P
cat gpurun_out/r03/lat_f0.txt gpurun_out/r03/lat_f16.txt
