#!/bin/bash
# round 3, GPU call: collated step (1-rank RCCL group) with the search on a high-priority stream, A/B against the default stream
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "high_priority or hipgraph" 2>&1 | tail -2
for w in maze32 rand64; do
  for v in hp nohp; do
    f=""; [ $v = nohp ] && f="--no-priority-stream"
    python bench.py --gpus 1 --force-collate $f --workload $w --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > gpurun_out/r03/force_collate_${w}_$v.json 2> gpurun_out/r03/force_collate_${w}_$v.err
    python - <<P
import json
j=json.load(open("gpurun_out/r03/force_collate_${w}_$v.json")); print("$w $v", round(j["value"]/1e6,2), "M maps/s", round(j["ms_per_step"]*1e3,1), "us/step", j["config"]["collate"][-60:])
P
  done
done
python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('plain', round(j['value']/1e6,2), round(j['ms_per_step']*1e3,1))"
